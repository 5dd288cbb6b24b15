#!/usr/bin/env python3
"""fp8 (e4m3, per-row scales) GEMM against the bf16 kernels on the step's forward shapes: device-event time per launch incl.
and excl. the activation quantiser, effective TF/s."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    for name, M, N, K, act in [("llm gate_up", 11264, 9728, 896, 4), ("llm down", 11264, 896, 4864, 0), ("llm qkv", 11264, 1152, 896, 0),
                               ("vit qkv", 8192, 3456, 1152, 0), ("vit fc1", 8192, 4352, 1152, 1), ("vit fc2", 8192, 1152, 4352, 0),
                               ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0)]:
        a = torch.randn(M, K, device=DEV).to(BF)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        out2 = torch.empty(M, N // 2, device=DEV, dtype=BF) if act == 4 else None
        qb, sb = ops.quant_fp8_rows(w)
        qa, sa = ops.quant_fp8_rows(a)
        kw = dict(act=act, out=out) if act != 4 else dict(act=4, out=out, out2=out2)
        t16 = statistics.median([timeit(lambda: ops.gemm_nt(a, w, split_k=0, **kw)) for _ in range(3)])
        t8 = statistics.median([timeit(lambda: ops.gemm_nt(qa, qb, fp8=(sa, sb), **kw)) for _ in range(3)])
        tq = statistics.median([timeit(lambda: ops.quant_fp8_rows(a, out=qa, scale=sa)) for _ in range(3)])
        fl = 2.0 * M * N * K
        print(f"{name:12s} {M:5d}x{N:4d}x{K:4d} | bf16 {t16*1e6:7.1f}us {fl/t16/1e12:5.0f}TF | fp8 {t8*1e6:7.1f}us {fl/t8/1e12:5.0f}TF | "
              f"quantise A {tq*1e6:6.1f}us | fp8 incl. {fl/(t8+tq)/1e12:5.0f}TF", flush=True)


if __name__ == "__main__":
    main()
