#!/usr/bin/env python3
"""Split-K factor sweep on the few-tile long-K GEMMs of the step (same box, back-to-back)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    for M, N, K in [(2048, 896, 9728), (2048, 896, 4864), (1024, 896, 9728), (4096, 896, 9728), (2048, 1152, 4352)]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = None
        line = f"{M}x{N}x{K}:"
        for sk in (0, 2, 4, 8):
            if K % (64 * max(sk, 1)):
                continue
            t = timeit(lambda: ops.gemm_nt(a, w, out=out, split_k=sk))
            if ref is None:
                ref = out.float().clone()
            err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            line += f"  sk={sk} {t * 1e6:6.1f}us {2 * M * N * K / t / 1e12:5.0f}TF (d {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
