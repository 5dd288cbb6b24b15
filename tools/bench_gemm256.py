#!/usr/bin/env python3
"""A/B of the GEMM tile choices on the step's hot shapes (same process, interleaved rounds, random operands):
auto (what the step runs) | 128x128 8-wave (VLA_GEMM_TILE=2) | 256x256 two-phase (VLA_GEMM_TILE=6) | vendor calibration (torch.matmul,
never on the product path).  Prints TF/s per variant: median over rounds."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    B = 32
    shapes = [("llm gate_up(swiglu)", B * 352, 9728, 896, 4), ("llm gate_up live-rows", B * 352, 9728, 896, 5), ("llm gate_up half", B * 176, 9728, 896, 4), ("llm down", B * 352, 896, 4864, 0),
              ("llm down half", B * 176, 896, 4864, 0), ("llm qkv(norope)", B * 352, 1152, 896, 0), ("llm o", B * 352, 896, 896, 0),
              ("vit qkv", B * 256, 3456, 1152, 0), ("vit proj", B * 256, 1152, 1152, 0), ("vit fc1(gelu)", B * 256, 4352, 1152, 1),
              ("vit fc2", B * 256, 1152, 4352, 0), ("vit fc2 +res", B * 256, 1152, 4352, 10), ("llm down +res", B * 352, 896, 4864, 10),
              ("llm o +res", B * 352, 896, 896, 10), ("head task kv", B * 256, 1792, 896, 0), ("llm d->dh", B * 352, 4864, 896, 0),
              ("llm dgu->dn", B * 352, 896, 9728, 0), ("live dgu->dn", 2048, 896, 9728, 0), ("live d->dh", 2048, 4864, 896, 0),
              ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0)]
    only = os.environ.get("SHAPES")
    for name, M, N, K, act in shapes:
        if only and not any(o in name for o in only.split(",")):
            continue
        a = torch.randn(M, K, device=DEV).to(BF)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
        bias = torch.randn(N, device=DEV).to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        if act in (4, 5):
            out2 = torch.empty(M, N // 2, device=DEV, dtype=BF)
            live = (352, 288) if act == 5 else None            # the step keeps pre-activations of the live rows only
            fn = lambda: ops.gemm_nt(a, w, act=4, out=out, out2=out2, c_live=live)
        elif act == 10:
            res_t = torch.randn(M, N, device=DEV).to(BF)
            fn = lambda: ops.gemm_nt(a, w, bias=bias, residual=res_t, out=out, split_k=0)
        else:
            fn = lambda: ops.gemm_nt(a, w, bias=bias, act=act, out=out, split_k=0)
        variants = [("auto", "0"), ("128x128", "2"), ("256x256", "6")]
        res = {k: [] for k, _ in variants}
        res["vendor"] = []
        wt = w.t()
        def setv(v):
            os.environ["VLA_GEMM_TILE"] = v.split(":")[0]
        for k, v in variants:
            setv(v)
            fn()
        torch.matmul(a, wt)
        for _ in range(5):
            for k, v in variants:
                setv(v)
                res[k].append(timeit(fn))
            if act not in (4, 5):
                res["vendor"].append(timeit(lambda: torch.matmul(a, wt)))
        os.environ["VLA_GEMM_TILE"] = "0"
        fl = 2.0 * M * N * K
        line = f"{name:22s} {M:5d}x{N:4d}x{K:4d}"
        for k in res:
            if res[k]:
                t = statistics.median(res[k])
                line += f" | {k} {t*1e6:7.1f}us {fl/t/1e12:5.0f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
