#!/usr/bin/env python3
"""Tile choice on the batch-16 shapes of the LoRA / full fine-tune dX chain (M = 5632 LLM rows, 4096 ViT rows; N = 896 / 1152): a
128 x 128 tile list of 288-308 tiles fills 60 % of the chip's 512 workgroup slots.  Same process, interleaved rounds, TF/s per
forced tile (VLA_GEMM_TILE: 2 = 128x128 8 waves, 3 = 128x64 4 waves, 6 = 256x256), median of 5."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


shapes = [("llm o / dX qkv", 5632, 896, 896), ("llm dX o", 5632, 896, 896), ("llm down fwd", 5632, 896, 4864), ("llm dX gu", 5632, 896, 9728),
          ("llm dX qkvT", 5632, 896, 1152), ("llm qkv fwd", 5632, 1152, 896), ("vit proj", 4096, 1152, 1152), ("vit fc2 fwd", 4096, 1152, 4352),
          ("vit dX qkv", 4096, 1152, 3456), ("vit qkv fwd", 4096, 3456, 1152), ("vit fc1 / dX fc2", 4096, 4352, 1152), ("llm gate_up", 5632, 9728, 896),
          ("llm dH (swiglu bwd size)", 5632, 4864, 896)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    fn = lambda: ops.gemm_nt(a, w, out=out, split_k=0)
    res = {}
    for v in ("0", "2", "3", "6"):
        os.environ["VLA_GEMM_TILE"] = v
        fn()
    for _ in range(5):
        for v in ("0", "2", "3", "6"):
            os.environ["VLA_GEMM_TILE"] = v
            res.setdefault(v, []).append(timeit(fn))
    os.environ["VLA_GEMM_TILE"] = "0"
    fl = 2.0 * M * N * K
    print(f"{name:26s} {M:5d}x{N:4d}x{K:4d} " + " | ".join(f"{ {'0': 'auto', '2': '128x128', '3': '128x64', '6': '256x256'}[v]} {statistics.median(t) * 1e6:6.1f}us {fl / statistics.median(t) / 1e12:5.0f}TF"
                                                             for v, t in res.items()), flush=True)
