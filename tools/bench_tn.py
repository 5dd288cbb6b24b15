#!/usr/bin/env python3
"""A/B of the TN (weight-gradient) product's two tile geometries on the shapes of the training steps: 128 x 128 (VLA_TN_TILE=128),
256 x 256 two-phase (VLA_TN_TILE=256), auto (what the steps run).  Same process, interleaved rounds, random operands; TF/s = median."""
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


def setv(v):
    if v:
        os.environ["VLA_TN_TILE"] = v
    else:
        os.environ.pop("VLA_TN_TILE", None)


def ab(name, fn, flops):
    res = {}
    for v in ("128", "256", ""):
        setv(v)
        fn()
    for _ in range(5):
        for v in ("128", "256", ""):
            setv(v)
            res.setdefault(v or "auto", []).append(timeit(fn))
    setv("")
    print(f"{name:34s}" + "".join(f" | {k:4s} {statistics.median(t) * 1e6:8.1f} us {flops / statistics.median(t) / 1e12:5.0f} TF" for k, t in res.items()), flush=True)


def main():
    M = 16 * 352
    llm = [("llm dW qkv", 1152, 896), ("llm dW o", 896, 896), ("llm dW gate_up", 9728, 896), ("llm dW down", 896, 4864)]
    for name, n1, n2 in llm + [("vit dW qkv", 3456, 1152), ("vit dW proj", 1152, 1152), ("vit dW fc1", 4352, 1152), ("vit dW fc2", 1152, 4352)]:
        m = M if name.startswith("llm") else 16 * 256
        a, b = torch.randn(m, n1, device=DEV).to(BF), torch.randn(m, n2, device=DEV).to(BF)
        out = torch.empty(n1, n2, device=DEV, dtype=BF)
        ab(f"{name} {m}x{n1}x{n2}", lambda: ops.gemm_tn(a, b, out=out, split=0), 2.0 * m * n1 * n2)
    # one LLM layer's weight gradients as one grouped launch (the trainers' form), x 4 layers
    dys = [torch.randn(M, n1, device=DEV).to(BF) for _, n1, _ in llm]
    xs = [torch.randn(M, n2, device=DEV).to(BF) for _, _, n2 in llm]
    probs = []
    for _ in range(4):
        for (nm, n1, n2), dy, x in zip(llm, dys, xs):
            probs.append(ops.tn_problem(dy, x, torch.empty(n1, n2, device=DEV, dtype=BF)))
    fl = 4 * sum(2.0 * M * n1 * n2 for _, n1, n2 in llm)
    ab("grouped: 4 LLM layers (16 products)", lambda: ops.gemm_tn_grouped(probs), fl)
    # the action head's task-token weight gradient: 24 blocks batched, M = 32 x 256 rows
    a, b = torch.randn(24, 8192, 1792, device=DEV).to(BF), torch.randn(24, 8192, 896, device=DEV).to(BF)
    out = torch.empty(24, 1792, 896, device=DEV, dtype=BF)
    ab("head dW task k|v (24 x 8192x1792x896)", lambda: ops.gemm_tn(a, b, out=out, split=0), 24 * 2.0 * 8192 * 1792 * 896)
    a, b = torch.randn(24, 256, 2688, device=DEV).to(BF), torch.randn(24, 256, 896, device=DEV).to(BF)
    out = torch.empty(24, 2688, 896, device=DEV, dtype=BF)
    ab("head dW x-path (24 x 256x2688x896)", lambda: ops.gemm_tn(a, b, out=out, split=0), 24 * 2.0 * 256 * 2688 * 896)


if __name__ == "__main__":
    main()
