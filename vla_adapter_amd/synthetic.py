"""Seeded synthetic weights and batches: there is no network for checkpoints or datasets, so
the benchmark, smoke test and parity tests use random-init weights of the exact architecture and synthetic batches
with the collator's contract (prismatic/util/data_utils.py:165-172).  State-dict key names are the reference's."""
from __future__ import annotations

from typing import Dict

import torch

from .engine import NUM_TOKENS, LLMCfg, ViTCfg, VLACfg

ACTION_TOKEN_BEGIN_IDX = 151386
PAD_ID = 151643
IGNORE_INDEX = -100


def _rn(gen, shape, std, device):
    return (torch.randn(*shape, generator=gen, device=device) * std).to(torch.bfloat16)


def vit_weights(c: ViTCfg, gen, device, std=0.02) -> Dict[str, torch.Tensor]:
    d, P = c.d, c.patch
    sd = {"patch_embed.proj.weight": _rn(gen, (d, 3, P, P), std, device), "patch_embed.proj.bias": _rn(gen, (d,), std, device),
          "pos_embed": _rn(gen, (1, c.n_patches, d), std, device)}
    if c.n_prefix:
        sd["cls_token"] = _rn(gen, (1, 1, d), std, device)
        if c.n_prefix > 1:
            sd["reg_token"] = _rn(gen, (1, c.n_prefix - 1, d), std, device)
    for i in range(c.depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = (1 + _rn(gen, (d,), 0.05, device).float()).to(torch.bfloat16)
        sd[p + "norm1.bias"] = _rn(gen, (d,), 0.02, device)
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = _rn(gen, (3 * d, d), std, device), _rn(gen, (3 * d,), std, device)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = _rn(gen, (d, d), std, device), _rn(gen, (d,), std, device)
        sd[p + "norm2.weight"] = (1 + _rn(gen, (d,), 0.05, device).float()).to(torch.bfloat16)
        sd[p + "norm2.bias"] = _rn(gen, (d,), 0.02, device)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = _rn(gen, (c.mlp, d), std, device), _rn(gen, (c.mlp,), std, device)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = _rn(gen, (d, c.mlp), std, device), _rn(gen, (d,), std, device)
        if c.layerscale:
            sd[p + "ls1.scale_factor"] = (0.5 + _rn(gen, (d,), 0.1, device).float()).to(torch.bfloat16)
            sd[p + "ls2.scale_factor"] = (0.5 + _rn(gen, (d,), 0.1, device).float()).to(torch.bfloat16)
    return sd


def llm_weights(c: LLMCfg, gen, device, std=0.02) -> Dict[str, torch.Tensor]:
    D, H, KV, dh, I = c.d, c.heads, c.kv_heads, c.dh, c.inter
    sd = {"embed_tokens.weight": _rn(gen, (c.vocab, D), std, device), "norm.weight": (1 + _rn(gen, (D,), 0.05, device).float()).to(torch.bfloat16)}
    for i in range(c.n_layers):
        p = f"layers.{i}."
        sd[p + "input_layernorm.weight"] = (1 + _rn(gen, (D,), 0.05, device).float()).to(torch.bfloat16)
        sd[p + "post_attention_layernorm.weight"] = (1 + _rn(gen, (D,), 0.05, device).float()).to(torch.bfloat16)
        for n, o in (("q_proj", H * dh), ("k_proj", KV * dh), ("v_proj", KV * dh)):
            sd[p + f"self_attn.{n}.weight"], sd[p + f"self_attn.{n}.bias"] = _rn(gen, (o, D), std, device), _rn(gen, (o,), std, device)
        sd[p + "self_attn.o_proj.weight"] = _rn(gen, (D, H * dh), std, device)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = _rn(gen, (I, D), std, device), _rn(gen, (I, D), std, device)
        sd[p + "mlp.down_proj.weight"] = _rn(gen, (D, I), std, device)
    return sd


def head_weights(cfg: VLACfg, gen, device, std=0.02):
    D, Da = cfg.llm.d, cfg.action_dim
    sd = {"model.layer_norm1.weight": torch.ones(Da * D, device=device, dtype=torch.bfloat16),
          "model.layer_norm1.bias": _rn(gen, (Da * D,), 0.02, device),
          "model.fc1.weight": _rn(gen, (D, Da * D), std, device), "model.fc1.bias": _rn(gen, (D,), std, device),
          "model.layer_norm2.weight": torch.ones(D, device=device, dtype=torch.bfloat16), "model.layer_norm2.bias": _rn(gen, (D,), 0.02, device),
          "model.fc2.weight": _rn(gen, (Da, D), std, device), "model.fc2.bias": _rn(gen, (Da,), std, device)}
    for i in range(cfg.num_blocks):
        p = f"model.mlp_resnet_blocks.{i}."
        names = ("q_proj", "k_self", "v_self", "k_adapter", "v_adapter", "k_task", "v_task", "o_proj", "ffn.1") if cfg.pro else (
            "q_proj", "k_proj", "v_proj", "o_proj", "ffn.1")           # MLPResNetBlock_Pro (:287-335) / MLPResNetBlock (:195-214)
        for n in names:
            sd[p + n + ".weight"], sd[p + n + ".bias"] = _rn(gen, (D, D), std, device), _rn(gen, (D,), std, device)
        sd[p + "ffn.0.weight"] = (1 + _rn(gen, (D,), 0.05, device).float()).to(torch.bfloat16)
        sd[p + "ffn.0.bias"] = _rn(gen, (D,), 0.02, device)
        sd[p + "gating_factor"] = torch.full((1,), 0.1, device=device, dtype=torch.bfloat16)
        if cfg.pro:
            sd[p + "film_gen.0.weight"], sd[p + "film_gen.0.bias"] = _rn(gen, (2 * D, D), std, device), _rn(gen, (2 * D,), std, device)
    pp = {"fc1.weight": _rn(gen, (D, cfg.proprio_dim), 0.2, device), "fc1.bias": _rn(gen, (D,), std, device),
          "fc2.weight": _rn(gen, (D, D), std, device), "fc2.bias": _rn(gen, (D,), std, device)}
    return sd, pp


def make_weights(cfg: VLACfg, device="cuda", seed: int = 0, std: float = 0.02) -> Dict:
    gen = torch.Generator(device=device).manual_seed(seed)
    D = cfg.llm.d
    W = dict(vit=[vit_weights(c, gen, device, std) for c in cfg.vit], llm=llm_weights(cfg.llm, gen, device, std))
    vd = cfg.vis_dim
    if cfg.fused:
        W["proj"] = {"fc1.weight": _rn(gen, (4 * vd, vd), std, device), "fc1.bias": _rn(gen, (4 * vd,), std, device),
                     "fc2.weight": _rn(gen, (D, 4 * vd), std, device), "fc2.bias": _rn(gen, (D,), std, device),
                     "fc3.weight": _rn(gen, (D, D), std, device), "fc3.bias": _rn(gen, (D,), std, device)}
    else:
        W["proj"] = {"fc1.weight": _rn(gen, (D, vd), std, device), "fc1.bias": _rn(gen, (D,), std, device),
                     "fc2.weight": _rn(gen, (D, D), std, device), "fc2.bias": _rn(gen, (D,), std, device)}
    W["head"], W["proprio"] = head_weights(cfg, gen, device, std)
    W["action_queries"] = _rn(gen, (NUM_TOKENS, D), std, device)     # zero-init in the reference (:375-376); N(0,.02) makes parity non-trivial
    return W


def make_batch(cfg: VLACfg, B: int, device="cuda", seed: int = 0, P: int = 32, ragged: bool = False) -> Dict[str, torch.Tensor]:
    """pixel_values ~ N(0,1) clipped to +-3; input_ids = P prompt ids + 64 action-token ids; labels keep the last 65
    ids (datasets.py:124); right padding with 151643 / -100 when ``ragged`` (data_utils.py:114-134)."""
    g = torch.Generator().manual_seed(seed)
    img, nb = cfg.vit[0].img, len(cfg.vit)
    vocab = cfg.llm.vocab
    px = torch.randn(B, 3 * nb * cfg.n_img, img, img, generator=g).clamp_(-3, 3)
    L = P + NUM_TOKENS
    ids = torch.full((B, L), min(PAD_ID, vocab - 1), dtype=torch.int64)
    labels = torch.full((B, L), IGNORE_INDEX, dtype=torch.int64)
    hi_lo, hi_hi = (ACTION_TOKEN_BEGIN_IDX + 1, PAD_ID) if vocab > PAD_ID else (vocab - 300, vocab - 1)
    for b in range(B):
        p = P - (int(torch.randint(0, 9, (1,), generator=g)) if ragged and b > 0 else 0)
        row = torch.cat([torch.randint(0, min(ACTION_TOKEN_BEGIN_IDX, vocab - 300), (p,), generator=g),
                         torch.randint(hi_lo, hi_hi, (NUM_TOKENS,), generator=g)])
        ids[b, :p + NUM_TOKENS] = row
        labels[b, p - 1:p + NUM_TOKENS] = row[p - 1:]
    if vocab <= PAD_ID:   # tiny vocab: action ids are remapped above ACTION_TOKEN_BEGIN_IDX in the LABELS only
        labels = torch.where(labels >= vocab - 300, labels + (ACTION_TOKEN_BEGIN_IDX + 1 - (vocab - 300)), labels)
    am = ids != min(PAD_ID, vocab - 1)
    batch = dict(pixel_values=px, input_ids=ids, labels=labels, attention_mask=am,
                 actions=torch.rand(B, cfg.chunk, cfg.action_dim, generator=g) * 2 - 1,
                 proprio=torch.rand(B, cfg.proprio_dim, generator=g) * 2 - 1)
    return {k: v.to(device) for k, v in batch.items()}
