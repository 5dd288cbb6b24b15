#!/usr/bin/env python3
"""Micro-benchmarks of the individual kernels at the hot-path shapes (config 2: SigLIP + Qwen2.5-0.5B, B=32)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
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
    res = []
    B = 32
    shapes = [("llm qkv", B * 352, 1152, 896, 0), ("llm o", B * 352, 896, 896, 0), ("llm gate_up(swiglu)", B * 352, 9728, 896, 4),
              ("llm down", B * 352, 896, 4864, 0), ("vit qkv", B * 256, 3456, 1152, 0), ("vit proj", B * 256, 1152, 1152, 0),
              ("vit fc1(gelu)", B * 256, 4352, 1152, 1), ("vit fc2", B * 256, 1152, 4352, 0), ("head task kv", B * 256, 1792, 896, 0),
              ("head x-path", B * 8, 2688, 896, 0), ("llm dgu->dn (K=9728)", B * 352, 896, 9728, 0), ("llm d->dh", B * 352, 4864, 896, 0),
              ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0)]
    for name, M, N, K, act in shapes:
        a = torch.randn(M, K, device=DEV).to(BF)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
        bias = torch.randn(N, device=DEV).to(BF)
        if act == 4:
            out2 = torch.empty(M, N // 2, device=DEV, dtype=BF)
            out = torch.empty(M, N, device=DEV, dtype=BF)
            fn = lambda: ops.gemm_nt(a, w, act=4, out=out, out2=out2)
        else:
            out = torch.empty(M, N, device=DEV, dtype=BF)
            fn = lambda: ops.gemm_nt(a, w, bias=bias, act=act, out=out)
        line = f"gemm {name:22s} {M:5d}x{N:4d}x{K:4d}"
        for tile, tn in ((0, "auto"), (1, "256x128w16"), (4, "256x256"), (3, "128x64")):
            os.environ["VLA_GEMM_TILE"] = str(tile)
            t = timeit(fn)
            tf = 2.0 * M * N * K / t / 1e12
            res.append(dict(kernel="gemm", tile=tn, name=name, M=M, N=N, K=K, us=t * 1e6, tflops=tf))
            line += f" | {tn} {t*1e6:6.1f}us {tf:5.0f}TF"
        os.environ["VLA_GEMM_TILE"] = "0"
        print(line, flush=True)
    # attention
    for name, Bn, S, Hq, Hkv, dh, causal in [("llm attn", B, 352, 14, 2, 64, True), ("vit attn", B, 256, 16, 16, 72, False)]:
        W = (Hq + 2 * Hkv) * dh
        qkv = torch.randn(Bn, S, W, device=DEV).to(BF)
        q, k, v = qkv[:, :, :Hq * dh], qkv[:, :, Hq * dh:(Hq + Hkv) * dh], qkv[:, :, (Hq + Hkv) * dh:]
        t = timeit(lambda: ops.attn_fwd(q, k, v, Hq, Hkv, dh, causal, None, want_lse=True))
        fl = 4.0 * Bn * Hq * S * S * dh * (0.5 if causal else 1.0)
        print(f"attn fwd {name:12s} {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
        res.append(dict(kernel="attn_fwd", name=name, us=t * 1e6, tflops=fl / t / 1e12))
        if dh == 64:
            o, lse = ops.attn_fwd(q, k, v, Hq, Hkv, dh, causal, None, want_lse=True)
            do = torch.randn_like(o)
            t = timeit(lambda: ops.attn_bwd(do, q, k, v, o, lse, Hq, Hkv, dh, causal, None))
            print(f"attn bwd {name:12s} {t*1e6:9.1f} us  {2.5*fl/t/1e12:7.1f} TF/s", flush=True)
            res.append(dict(kernel="attn_bwd", name=name, us=t * 1e6, tflops=2.5 * fl / t / 1e12))
    # norms (HBM-bound): bytes = read + write
    x = torch.randn(B * 352, 896, device=DEV).to(BF)
    w = torch.ones(896, device=DEV, dtype=BF)
    t = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6))
    print(f"rmsnorm fwd {t*1e6:9.1f} us  {2*x.numel()*2/t/1e9:7.1f} GB/s", flush=True)
    res.append(dict(kernel="rmsnorm_fwd", us=t * 1e6, gbps=2 * x.numel() * 2 / t / 1e9))
    x = torch.randn(B * 256, 1152, device=DEV).to(BF)
    w = torch.ones(1152, device=DEV, dtype=BF)
    b = torch.zeros(1152, device=DEV, dtype=BF)
    t = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-6))
    print(f"layernorm fwd {t*1e6:9.1f} us  {2*x.numel()*2/t/1e9:7.1f} GB/s", flush=True)
    res.append(dict(kernel="layernorm_fwd", us=t * 1e6, gbps=2 * x.numel() * 2 / t / 1e9))
    n = 218_000_000
    p = torch.randn(n, device=DEV).to(BF); g = torch.randn(n, device=DEV).to(BF); m = torch.zeros_like(p); v = torch.zeros_like(p)
    t = timeit(lambda: ops.adamw_(p, g, m, v, 1, 5e-4), iters=5)
    print(f"adamw 218M {t*1e6:9.1f} us  {7*n*2/t/1e9:7.1f} GB/s", flush=True)
    res.append(dict(kernel="adamw", us=t * 1e6, gbps=7 * n * 2 / t / 1e9))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_kernels.json", "w"), indent=1)


if __name__ == "__main__":
    main()
