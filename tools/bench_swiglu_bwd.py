#!/usr/bin/env python3
"""The live-row backward's fused dH GEMM + SwiGLU backward in the step's form (M = 2048 live rows gathered from the
[B, 352, 2I] pre-activations through row-group addressing) and the full-sequence form: us per launch, TF/s of the GEMM."""
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
    B, S, D, I, r0 = 32, 352, 896, 4864, 288
    R = S - r0
    pre = torch.randn(B * S, 2 * I, device=DEV).to(BF)
    wdT = (torch.randn(I, D, device=DEV) * 0.02).to(BF)
    d_live = torch.randn(B * R, D, device=DEV).to(BF)
    d_full = torch.randn(B * S, D, device=DEV).to(BF)
    for name, fn, M in [("live rows (M 2048, row groups)", lambda: ops.gemm_swiglu_bwd(d_live, wdT, pre[r0:], gu_group=(R, S * 2 * I)), B * R),
                        ("full sequence (M 11264)", lambda: ops.gemm_swiglu_bwd(d_full, wdT, pre), B * S)]:
        for tile in ("0", "2", "6"):
            os.environ["VLA_GEMM_TILE"] = tile
            t = statistics.median([timeit(fn) for _ in range(5)])
            print(f"{name:32s} tile {tile} | {t*1e6:7.1f}us {2.0*M*I*D/t/1e12:5.0f}TF", flush=True)
    os.environ["VLA_GEMM_TILE"] = "0"


if __name__ == "__main__":
    main()
