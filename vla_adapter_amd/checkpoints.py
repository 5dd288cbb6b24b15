"""Checkpoint interop with the reference's file / key layout.

* VLM weights: the HF-style state dict of ``OpenVLAForActionPrediction`` (modeling_prismatic.py) -
  ``vision_backbone.featurizer.*`` (DINOv2 in the fused setup, the only backbone otherwise),
  ``vision_backbone.fused_featurizer.*`` (SigLIP), ``projector.fc{1,2,3}.*``, ``language_model.model.*``,
  ``action_queries.weight`` - or a native Prismatic checkpoint, whose keys the reference renames with the map at
  vla-scripts/finetune.py:792-800 before loading.  ``split_reference_state_dict`` turns either into the weight dict
  ``engine.VLAEngine`` takes; ``merge_reference_state_dict`` is its inverse (what ``vla.state_dict()`` would hold).
* Trainable parts: ``action_head--{step}_checkpoint.pt`` / ``proprio_projector--{step}_checkpoint.pt`` with the reference's
  key names (finetune.py:527-572) are written by ``finetune.save_training_checkpoint`` and read back through
  ``Head.load_state_dicts``; DDP's ``module.`` prefix is stripped on load (finetune.py:132-154).
Only loaders that execute nothing from the file are used (torch.load(weights_only=True) / safetensors).
* LoRA adapters: ``lora_adapter/adapter_model.safetensors`` under peft's key names is written by
  ``finetune.save_training_checkpoint`` and read back by ``load_lora_adapter`` (resume) / ``merge_lora_into_state_dict`` (the
  offline merge of vla-scripts/merge_lora_weights_and_save.py:44-103: W + (alpha / r) B A into a base state dict).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .engine import VLACfg

# vla-scripts/finetune.py:792-800 (native Prismatic checkpoint -> HF module names)
PRISMATIC_TO_HF = [("vision_backbone.dino_featurizer", "vision_backbone.featurizer"),
                   ("vision_backbone.siglip_featurizer", "vision_backbone.fused_featurizer"),
                   ("llm_backbone.llm", "language_model"),
                   ("projector.projector.0", "projector.fc1"), ("projector.projector.2", "projector.fc2"),
                   ("projector.projector.4", "projector.fc3"), ("gamma", "scale_factor")]


def rename_prismatic_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """rename_state_dict_keys of finetune.py:802-810: every matching substring is replaced, in map order."""
    out = {}
    for k, v in sd.items():
        nk = k
        for old, new in PRISMATIC_TO_HF:
            if old in nk:
                nk = nk.replace(old, new)
        out[nk] = v
    return out


def strip_ddp_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """remove_ddp_in_checkpoint (finetune.py:132-154)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_file(path: str) -> Dict[str, torch.Tensor]:
    """.safetensors or a torch checkpoint (tensors only; weights_only=True refuses anything that would unpickle code)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file as _lf
        return _lf(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    return obj.get("model", obj) if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict) else obj


def split_reference_state_dict(sd: Dict[str, torch.Tensor], cfg: VLACfg, head: Optional[Dict[str, torch.Tensor]] = None,
                               proprio: Optional[Dict[str, torch.Tensor]] = None) -> Dict:
    """HF-style (or native Prismatic) VLM state dict -> {"vit": [sd...], "proj": sd, "llm": sd, "action_queries": t,
    "head": sd, "proprio": sd} with the per-module key names engine.ViT / engine.LLM read."""
    sd = strip_ddp_prefix(rename_prismatic_keys(sd))
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    feats = [sub("vision_backbone.featurizer.")]
    if cfg.fused:
        feats.append(sub("vision_backbone.fused_featurizer."))
    assert all(feats), "state dict holds no vision_backbone.(fused_)featurizer.* weights"
    for i, f in enumerate(feats):                 # LayerScale may still be called gamma in an un-renamed timm checkpoint
        feats[i] = {k.replace(".gamma", ".scale_factor"): v for k, v in f.items()}
    llm = sub("language_model.model.")
    assert "embed_tokens.weight" in llm and "norm.weight" in llm, "state dict holds no language_model.model.* weights"
    W = dict(vit=feats, proj=sub("projector."), llm=llm)
    if "action_queries.weight" in sd:
        W["action_queries"] = sd["action_queries.weight"]
    if head is not None:
        W["head"] = strip_ddp_prefix(head)
    if proprio is not None:
        W["proprio"] = strip_ddp_prefix(proprio)
    return W


def merge_reference_state_dict(W: Dict, cfg: VLACfg) -> Dict[str, torch.Tensor]:
    """Inverse of split_reference_state_dict for the VLM part (+ action_queries)."""
    out = {}
    names = ["vision_backbone.featurizer."] + (["vision_backbone.fused_featurizer."] if cfg.fused else [])
    for pre, f in zip(names, W["vit"]):
        out.update({pre + k: v for k, v in f.items()})
    out.update({"projector." + k: v for k, v in W["proj"].items()})
    out.update({"language_model.model." + k: v for k, v in W["llm"].items()})
    if W.get("action_queries") is not None:
        out["action_queries.weight"] = W["action_queries"]
    return out


def load_run_dir(run_dir: str, step, device="cpu", with_action_queries: bool = False):
    """(head state dict, proprio state dict[, action queries or None]) of a reference-layout run directory /
    ``--{step}_chkpt`` directory.  ``action_queries--*.pt`` is this build's addition (the reference keeps the queries in its
    LoRA / VLM checkpoint); a directory written by the reference has none, and one without a proprio projector file
    (use_proprio was off) yields an empty dict for it."""
    suffix = "latest_checkpoint.pt" if step in (None, "latest") else f"{step}_checkpoint.pt"
    path = lambda n: os.path.join(run_dir, f"{n}--{suffix}")
    ld = lambda n: strip_ddp_prefix(torch.load(path(n), map_location=device, weights_only=True))
    head = ld("action_head")
    proprio = ld("proprio_projector") if os.path.exists(path("proprio_projector")) else {}
    if not with_action_queries:
        return head, proprio
    aq = ld("action_queries")["weight"] if os.path.exists(path("action_queries")) else None
    return head, proprio, aq


def engine_vlm_state_dict(eng) -> Dict[str, torch.Tensor]:
    """The VLM weights the engine holds (fused / padded layouts) under the reference's HF key names: q|k|v split, gate / up
    de-interleaved, the ViT MLP's zero padding and the patch embedding's K padding removed.  Used for full-fine-tune and
    LoRA-merged checkpoints (vla-scripts/finetune.py:556-601 saves the whole VLM with save_pretrained).  The last ViT block
    (never computed here, dead in the reference too) is not part of the engine and therefore absent."""
    cfg, out = eng.cfg, {}
    c = cfg.llm
    H, KV, dh, I, D = c.heads, c.kv_heads, c.dh, c.inter, c.d
    cl = lambda t: t.detach().clone()
    for i, L in enumerate(eng.llm.layers):
        p = f"language_model.model.layers.{i}."
        w, b = L["wqkv"], L["bqkv"]
        for n, lo, hi in (("q_proj", 0, H * dh), ("k_proj", H * dh, (H + KV) * dh), ("v_proj", (H + KV) * dh, (H + 2 * KV) * dh)):
            out[p + f"self_attn.{n}.weight"], out[p + f"self_attn.{n}.bias"] = cl(w[lo:hi]), cl(b[lo:hi])
        out[p + "self_attn.o_proj.weight"] = cl(L["wo"])
        gu = L["wgu"].view(I // 16, 2, 16, D)
        out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = gu[:, 0].reshape(I, D).clone(), gu[:, 1].reshape(I, D).clone()
        out[p + "mlp.down_proj.weight"] = cl(L["wd"])
        out[p + "input_layernorm.weight"], out[p + "post_attention_layernorm.weight"] = cl(L["n1"]), cl(L["n2"])
    out["language_model.model.norm.weight"], out["language_model.model.embed_tokens.weight"] = cl(eng.llm.norm), cl(eng.llm.embed)
    names = ["vision_backbone.featurizer."] + (["vision_backbone.fused_featurizer."] if cfg.fused else [])
    for pre, v in zip(names, eng.vits):
        vc = v.cfg
        P_ = vc.patch
        if vc.n_prefix:                              # DINOv2: cls + register tokens (timm names)
            out[pre + "cls_token"] = v.prefix[:1].reshape(1, 1, vc.d).clone()
            if vc.n_prefix > 1:
                out[pre + "reg_token"] = v.prefix[1:].reshape(1, vc.n_prefix - 1, vc.d).clone()
        out[pre + "patch_embed.proj.weight"] = v.wpe[:, :3 * P_ * P_].reshape(vc.d, 3, P_, P_).clone()
        out[pre + "patch_embed.proj.bias"], out[pre + "pos_embed"] = cl(v.bpe), v.pos.reshape(1, -1, vc.d).clone()
        for i, b in enumerate(v.blocks):
            q = f"{pre}blocks.{i}."
            out[q + "norm1.weight"], out[q + "norm1.bias"], out[q + "norm2.weight"], out[q + "norm2.bias"] = cl(b["n1w"]), cl(b["n1b"]), cl(b["n2w"]), cl(b["n2b"])
            out[q + "attn.qkv.weight"], out[q + "attn.qkv.bias"] = cl(b["wqkv"]), cl(b["bqkv"])
            out[q + "attn.proj.weight"], out[q + "attn.proj.bias"] = cl(b["wproj"]), cl(b["bproj"])
            out[q + "mlp.fc1.weight"], out[q + "mlp.fc1.bias"] = b["w1"][:vc.mlp].clone(), b["b1"][:vc.mlp].clone()
            out[q + "mlp.fc2.weight"], out[q + "mlp.fc2.bias"] = b["w2"][:, :vc.mlp].clone(), cl(b["b2"])
            if vc.layerscale:                        # the parameters themselves (engine.ViT keeps them beside the folded copies)
                out[q + "ls1.scale_factor"], out[q + "ls2.scale_factor"] = cl(b["ls1"]), cl(b["ls2"])
    for k, t in eng.proj.items():
        out["projector." + k] = cl(t)
    out["action_queries.weight"] = cl(eng.head.P.view("action_queries"))
    return out


# heads are not recoverable from tensor shapes: the widths the reference's backbones use (timm vit_so400m / vit_large / the
# plumbing-size configs of engine.py); anything else is read as 64-wide heads
_VIT_HEADS = {1152: 16, 1024: 16, 768: 12, 384: 6, 256: 4, 192: 3, 128: 2}
_LLM_HEAD_DIM = {896: 64, 1536: 128, 2048: 128, 3584: 128}      # Qwen2.5-0.5B / 1.5B / 3B / 7B


def infer_config(sd: Dict[str, torch.Tensor]) -> VLACfg:
    """Model geometry read off a VLM state dict (HF-style or native Prismatic keys): which vision backbones (one, or DINOv2 + SigLIP
    fused), their widths / depths / prefix tokens / LayerScale, the projector form, the Qwen2 geometry.  The reference gets this
    from the checkpoint's config.json (vla-scripts/finetune.py:777-816); a bare state dict carries the same information in its
    keys and shapes, except the head counts (table above)."""
    from .engine import LLMCfg, ViTCfg, VLACfg as _V
    sd = strip_ddp_prefix(rename_prismatic_keys(sd))
    vits = []
    for pre in ("vision_backbone.featurizer.", "vision_backbone.fused_featurizer."):
        sub = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        if not sub:
            continue
        pos, pw = sub["pos_embed"], sub["patch_embed.proj.weight"]
        d, n_patches, patch = pos.shape[-1], pos.shape[-2], pw.shape[-1]
        depth = 1 + max(int(k.split(".")[1]) for k in sub if k.startswith("blocks."))
        n_prefix = (1 if "cls_token" in sub else 0) + (sub["reg_token"].shape[-2] if "reg_token" in sub else 0)
        ls = any(".ls1." in k for k in sub)
        side = int(round(n_patches ** 0.5))
        assert side * side == n_patches, "square patch grids only"
        vits.append(ViTCfg(d, depth, _VIT_HEADS.get(d, max(1, d // 64)), sub["blocks.0.mlp.fc1.weight"].shape[0], patch, patch * side, n_prefix, ls))
    assert vits, "state dict holds no vision_backbone.(fused_)featurizer.* weights"
    lm = {k[len("language_model.model."):]: v for k, v in sd.items() if k.startswith("language_model.model.")}
    D, vocab = lm["embed_tokens.weight"].shape[1], lm["embed_tokens.weight"].shape[0]
    n_layers = 1 + max(int(k.split(".")[1]) for k in lm if k.startswith("layers."))
    dh = _LLM_HEAD_DIM.get(D, 64)
    heads, kv = lm["layers.0.self_attn.q_proj.weight"].shape[0] // dh, lm["layers.0.self_attn.k_proj.weight"].shape[0] // dh
    llm = LLMCfg(D, n_layers, heads, kv, dh, lm["layers.0.mlp.gate_proj.weight"].shape[0], 1e-6, 1e6, vocab)
    cfg = _V(vit=vits, llm=llm, num_blocks=min(24, n_layers))
    assert ("projector.fc3.weight" in sd) == cfg.fused, "a fused (two-backbone) VLM has the 3-layer projector, a single-backbone one fc1 / fc2"
    return cfg


def load_lora_adapter(adapter_dir: str, with_config: bool = False):
    """``lora_adapter/adapter_model.safetensors`` (peft key names) of a run directory written by finetune.save_training_checkpoint
    or by peft's ``save_pretrained`` (vla-scripts/finetune.py:553-554).  with_config: also return ``adapter_config.json`` as a dict
    ({} when the file is absent) - ``r`` and ``lora_alpha`` decide the merge scale (lora_scaling)."""
    sd = load_file(os.path.join(adapter_dir, "adapter_model.safetensors"))
    if not with_config:
        return sd
    cfg_path = os.path.join(adapter_dir, "adapter_config.json")
    cfg = {}
    if os.path.exists(cfg_path):
        import json
        with open(cfg_path) as f:
            cfg = json.load(f)
    return sd, cfg


def lora_scaling(adapter_config: Optional[Dict] = None, scaling: Optional[float] = None) -> float:
    """lora_alpha / r of an adapter (peft's LoraLayer.scaling).  Every reference script uses lora_alpha = 2 r (finetune.py:835), which
    is the fallback when no adapter_config.json came with the adapter; an explicit ``scaling`` that disagrees with the config is an
    error rather than a silently wrong merge."""
    from_cfg = None
    if adapter_config and adapter_config.get("r") and adapter_config.get("lora_alpha") is not None:
        from_cfg = float(adapter_config["lora_alpha"]) / float(adapter_config["r"])
    if scaling is not None and from_cfg is not None and abs(scaling - from_cfg) > 1e-6 * max(1.0, abs(from_cfg)):
        raise ValueError(f"merge scale {scaling} disagrees with adapter_config.json (lora_alpha / r = {from_cfg})")
    return scaling if scaling is not None else from_cfg if from_cfg is not None else 2.0


def merge_lora_into_state_dict(base: Dict[str, torch.Tensor], adapter: Dict[str, torch.Tensor], scaling: Optional[float] = None,
                               adapter_config: Optional[Dict] = None) -> Dict[str, torch.Tensor]:
    """Offline merge of vla-scripts/merge_lora_weights_and_save.py:44-103 (PeftModel.from_pretrained(...).merge_and_unload()):
    W <- W + scaling * B @ A for every adapted Linear of an HF-style VLM state dict (fp32 product, one bf16 rounding); scaling =
    lora_alpha / r from ``adapter_config`` (load_lora_adapter(..., with_config=True)), 2 when there is none (finetune.py:835).
    Keys: '<prefix>base_model.model.<module>.lora_A.weight', or peft's in-memory form with the adapter name as an infix
    ('....lora_A.default.weight')."""
    scaling = lora_scaling(adapter_config, scaling)
    out = dict(base)
    pre = "base_model.model."
    for k, A in adapter.items():
        parts = k.split(".")
        if "lora_A" not in parts or parts[-1] != "weight":
            continue
        ia = parts.index("lora_A")
        mod, infix = ".".join(parts[:ia]), parts[ia + 1:-1]          # infix: [] on disk, [adapter name] in a live peft model
        kb = ".".join(parts[:ia] + ["lora_B"] + infix + ["weight"])
        if kb not in adapter:
            raise KeyError(f"adapter holds {k} without its {kb}")
        B = adapter[kb]
        name = (mod[len(pre):] if mod.startswith(pre) else mod) + ".weight"
        if name not in out:
            raise KeyError(f"adapter targets {name}, which the base state dict does not hold")
        W = out[name]
        out[name] = (W.float() + scaling * (B.float() @ A.float())).to(W.dtype)
    return out
