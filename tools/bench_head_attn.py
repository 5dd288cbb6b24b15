#!/usr/bin/env python3
"""Isolated timing of the action-head attention kernels at the step's shape (B 32, T 8, Ka 65, Kt 256, 8 heads x 112)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B, T, Ka, Kt, H, D = 32, 8, 65, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 8, 896
    g = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(torch.bfloat16)
    x3, a2, t2 = g(B, T, 3 * D), g(B, Ka, 2 * D), g(B, Kt, 2 * D)
    gate, dout = torch.tensor([0.7], device="cuda").to(torch.bfloat16), g(B, T, D)
    args = (x3[:, :, :D], x3[:, :, D:2 * D], x3[:, :, 2 * D:], a2[:, :, :D], a2[:, :, D:], t2[:, :, :D], t2[:, :, D:])
    out, probs = ops.head_attn_fwd(*args, gate, H)
    g3, ga, gt, dg = torch.zeros_like(x3), torch.zeros_like(a2), torch.zeros_like(t2), torch.zeros(1, device="cuda")
    tabs = ops.rope_inter_tables(max(T, Ka, Kt), D // H, "cuda")
    bwd = lambda: ops.head_attn_bwd(dout, out, *args, gate, probs, dg, g3[:, :, :D], g3[:, :, D:2 * D], g3[:, :, 2 * D:], ga[:, :, :D],
                                    ga[:, :, D:], gt[:, :, :D], gt[:, :, D:], H, rope=tabs)
    for _ in range(2):
        print(f"head attn fwd {timeit(lambda: ops.head_attn_fwd(*args, gate, H)):6.1f} us   bwd {timeit(bwd):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
