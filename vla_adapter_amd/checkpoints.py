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
