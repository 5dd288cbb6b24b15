"""Algorithmic FLOP model of the fine-tune hot path: the single source for roofline.achieved.

2 FLOPs per MAC.  Excluded (never computed by this build, dead work in the reference): the last ViT block, the
lm_head, film_gen, hidden_states[0].  Two backward conventions are reported side by side:

* ``step`` (what torch.autograd executes in the reference): dX GEMMs for every op downstream of a
  trainable tensor (action_queries sits at the LLM input -> the whole LLM gets dX over all S rows), dW GEMMs only for
  trainable weights, attention backward = 2x attention forward.
* ``step_live`` (what this build executes by default): the same, minus the gradient rows that can only reach frozen
  inputs - LLM backward over the rows >= row0 only (engine.LLM.backward) and no dX for the head's task-token
  projections.  Identical parameter gradients, fewer FLOPs; throughput fractions quoted against this number count
  only work that was really done.
"""
from __future__ import annotations

from .engine import NUM_TOKENS, VLACfg


def vit_fwd(c) -> float:
    T, Lu = c.n_patches + c.n_prefix, c.depth - 1
    return Lu * (2 * T * (4 * c.d ** 2 + 2 * c.d * c.mlp) + 4 * T * T * c.d) + 2 * c.n_patches * (3 * c.patch ** 2) * c.d


def proj_fwd(cfg: VLACfg) -> float:
    D, vd, Np = cfg.llm.d, cfg.vis_dim, cfg.n_patches
    per_tok = (vd * 4 * vd + 4 * vd * D + D * D) if cfg.fused else (vd * D + D * D)
    return 2 * Np * per_tok


def llm_linear_fwd(cfg: VLACfg, S: int) -> float:
    c = cfg.llm
    p_layer = c.d * (c.heads + 2 * c.kv_heads) * c.dh + c.heads * c.dh * c.d + 3 * c.d * c.inter
    return c.n_layers * 2 * S * p_layer


def llm_attn_fwd(cfg: VLACfg, S: int) -> float:
    c = cfg.llm
    return c.n_layers * 2 * S * S * c.heads * c.dh          # causal-useful half of 4 S^2 d


def head_fwd(cfg: VLACfg) -> float:
    D, T, A, Kt = cfg.llm.d, cfg.chunk, NUM_TOKENS, cfg.n_patches
    return cfg.num_blocks * (2 * D * D * (5 * T + 2 * (A + 1) + 2 * Kt) + 4 * T * (T + A + 1 + Kt) * D) + 2 * T * cfg.action_dim * D * D


def llm_attn_bwd_live(cfg: VLACfg, S: int, row0: int) -> float:
    """dQ for the queries >= row0 over their visible keys + dK/dV for the keys >= row0 from the queries >= them:
    (QK^T recompute, dP, dQ) on the [row0, S) x [0, S) causal trapezoid, (dK, dV) on the [row0, S)^2 causal triangle,
    2 * dh FLOPs per (query, key, head) pair and product."""
    c = cfg.llm
    R = S - row0
    trapezoid = R * row0 + R * (R + 1) / 2
    triangle = R * (R + 1) / 2
    return c.n_layers * c.heads * 2 * c.dh * (3 * trapezoid + 2 * triangle)


def step_flops_per_sample(cfg: VLACfg, L: int = 96, row0: int = 0) -> dict:
    """Adapter-only fine-tune (BASELINE configs 2/3): frozen ViT + projector forward only; LLM forward + dX; head x3.
    row0 = first live row of the LLM backward (engine.VLAEngine.live_row0)."""
    S = cfg.n_patches + L
    vit = sum(vit_fwd(c) for c in cfg.vit) * cfg.n_img
    proj, lin, att, head = proj_fwd(cfg), llm_linear_fwd(cfg, S), llm_attn_fwd(cfg, S), head_fwd(cfg)
    total = vit + proj + 2 * lin + 3 * att + 3 * head
    # executed: the LLM layers above the head's last block (Qwen2.5-1.5B: 4 of 28) reach neither the loss nor the actions and are not
    # run by the training step (engine.VLAEngine.n_act); the autograd convention above counts them, as the reference executes them
    fa = min(cfg.llm.n_layers, cfg.num_blocks) / cfg.llm.n_layers
    lin_a, att_a = lin * fa, att * fa
    task_dx = cfg.num_blocks * 2 * cfg.n_patches * 2 * cfg.llm.d * cfg.llm.d        # dX of k_task / v_task (dead when row0 > 0)
    live = vit + proj + lin_a * (1 + (S - row0) / S) + att_a + llm_attn_bwd_live(cfg, S, row0) * fa + 3 * head - (task_dx if row0 else 0)
    full_a = vit + proj + 2 * lin_a + 3 * att_a + 3 * head
    return dict(vit_fwd=vit, proj_fwd=proj, llm_linear_fwd=lin, llm_attn_fwd=att, head_fwd=head, forward=vit + proj + lin + att + head,
                step=total, step_live=live if row0 else full_a)


def source_digest() -> str:
    """sha256 over the kernel sources and the step schedule (csrc/*.hip, *.h, engine.py, ops.py, trainers.py): stamped into
    profiles/*_gemm_in_situ.json when a profile is reduced, compared by bench.py - a profile taken from other code is reported
    as stale instead of being quoted as a fresh measurement."""
    import glob
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h"))) + \
        [os.path.join(here, n) for n in ("engine.py", "ops.py", "trainers.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    from .engine import config2
    for k, v in step_flops_per_sample(config2(), row0=288).items():
        print(f"{k:16s} {v / 1e9:10.1f} GF/sample")
