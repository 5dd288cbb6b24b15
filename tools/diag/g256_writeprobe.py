#!/usr/bin/env python3
"""Diagnostic: is the gemm256 epilogue bound by the memory side of its stores?  Same products, same tiles; C either the real
[batch, 256, N] tensor or ONE [256, N] tile row that every batch element overwrites (batch stride 0: the stores hit a footprint
that stays in the L2s / the Infinity Cache)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


os.environ["VLA_GEMM_TILE"] = "6"
for name, nb, N, K in [("vit qkv", 32, 3456, 1152), ("llm qkv-like", 44, 1152, 896), ("gate_up-like plain", 44, 9728, 896), ("down-like", 44, 896, 4864), ("sq", 32, 8192, 1024)]:
    a = torch.randn(nb, 256, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    bias = torch.randn(N, device=DEV).to(BF)
    real = torch.empty(nb, 256, N, device=DEV, dtype=BF)
    one = torch.empty(256, N, device=DEV, dtype=BF).unsqueeze(0).expand(nb, 256, N)
    f_real = lambda: ops.gemm_nt(a, w, bias=bias, out=real, split_k=0)
    f_one = lambda: ops.gemm_nt(a, w, bias=bias, out=one, split_k=0)
    r = {"real": [], "one": []}
    for _ in range(3):
        f_real(); r["real"].append(timeit(f_real))
        f_one(); r["one"].append(timeit(f_one))
    fl = 2.0 * nb * 256 * N * K
    tr, to = statistics.median(r["real"]), statistics.median(r["one"])
    print(f"{name:20s} batch {nb} x 256 x {N} x {K}: real C {tr:7.1f} us {fl / tr / 1e6:6.0f} TF | one-tile-row C {to:7.1f} us {fl / to / 1e6:6.0f} TF", flush=True)
