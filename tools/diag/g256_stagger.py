#!/usr/bin/env python3
"""Diagnostic: gemm256 launch time against the free-stagger fraction (VLA_GEMM256_STAGGER = percent of a tile time by which the
workgroups of the partial last round start late)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
os.environ["VLA_GEMM_TILE"] = "6"
B = 32
for name, M, N, K, act in [("gate_up live", B * 352, 9728, 896, 5), ("gate_up full", B * 352, 9728, 896, 4), ("vit qkv", B * 256, 3456, 1152, 0), ("vit fc1", B * 256, 4352, 1152, 1),
                           ("llm d->dh", B * 352, 4864, 896, 0), ("sq 8192 K1024", 8192, 8192, 1024, 0), ("8192x6400x1024 (25 col tiles)", 8192, 6400, 1024, 0)]:
    a = torch.randn(M, K, device=DEV).to(BF); w = (torch.randn(N, K, device=DEV) * 0.02).to(BF); bias = torch.randn(N, device=DEV).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    if act in (4, 5):
        out2 = torch.empty(M, N // 2, device=DEV, dtype=BF); live = (352, 288) if act == 5 else None
        fn = lambda: ops.gemm_nt(a, w, act=4, out=out, out2=out2, c_live=live)
    else:
        fn = lambda: ops.gemm_nt(a, w, bias=bias, act=act, out=out, split_k=0)
    res = {}
    for rnd in range(3):
        for st in (0, 25, 50, 75):
            os.environ["VLA_GEMM256_STAGGER"] = str(st)            # (0 = off; the library's default is 50)
            fn(); res.setdefault(st, []).append(timeit(fn))
    tiles = -(-M // 256) * -(-N // 256)
    print(f"{name:30s} tiles {tiles:5d} = {tiles / 256:5.2f} rounds" + "".join(f" | {st:2d}%: {statistics.median(v):7.1f}us" for st, v in res.items()), flush=True)
