#!/usr/bin/env python3
"""Attention forward / backward on the step's shapes: us per launch and (causal-)effective TF/s (forward FLOPs 4 B H S^2 dh)."""
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
    for name, B, S, Hq, Hkv, dh, causal in [("llm 0.5B (B 32)", 32, 352, 14, 2, 64, True), ("llm 0.5B (B 16)", 16, 352, 14, 2, 64, True),
                                            ("siglip", 32, 256, 16, 16, 72, False), ("dinov2", 32, 261, 16, 16, 64, False)]:
        W = (Hq + 2 * Hkv) * dh
        qkv = (torch.randn(B, S, W, device=DEV) * 0.5).to(BF)
        q, k, v = qkv[:, :, :Hq * dh], qkv[:, :, Hq * dh:(Hq + Hkv) * dh], qkv[:, :, (Hq + Hkv) * dh:]
        ts = [timeit(lambda: ops.attn_fwd(q, k, v, Hq, Hkv, dh, causal, None, want_lse=True)) for _ in range(5)]
        t = statistics.median(ts)
        fl = 4.0 * B * Hq * S * S * dh * (0.5 if causal else 1.0)
        print(f"{name:18s} | fwd {t*1e6:7.1f}us {fl/t/1e12:5.0f}TF", flush=True)


if __name__ == "__main__":
    main()
