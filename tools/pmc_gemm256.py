#!/usr/bin/env python3
"""The four big products of the adapter-only step on gemm256_kernel, with the step's own epilogues, three launches each (for the
rocprofv3 --pmc passes of tools/collect_gemm256_pmc.sh; the program itself follows `--` directly).  Shapes at batch 32, S = 352:
LLM gate|up (SwiGLU epilogue, pre-activations kept for the live rows only), LLM down (+ residual), ViT qkv (+ bias), ViT fc2
(+ bias + residual)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SHAPES = [("llm_gate_up", 11264, 9728, 896), ("llm_down", 11264, 896, 4864), ("vit_qkv", 8192, 3456, 1152), ("vit_fc2", 8192, 1152, 4352)]
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(BF)
meta = []
for name, M, N, K in SHAPES:
    a, w = rn(M, K), rn(N, K, sc=0.02)
    for _ in range(3):
        if name == "llm_gate_up":
            pre, h = torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N // 2, device=DEV, dtype=BF)
            ops.gemm_nt(a, w, act=ops.ACT_SWIGLU, out=pre, out2=h, c_live=(352, 288))
            alg = 2 * (M * K + N * K + M * N * 64 // 352 + M * N // 2)
        elif name == "llm_down":
            r = rn(M, N)
            ops.gemm_nt(a, w, residual=r, out=torch.empty(M, N, device=DEV, dtype=BF))
            alg = 2 * (M * K + N * K + 2 * M * N)
        elif name == "vit_qkv":
            ops.gemm_nt(a, w, bias=rn(N), out=torch.empty(M, N, device=DEV, dtype=BF))
            alg = 2 * (M * K + N * K + M * N)
        else:
            r = rn(M, N)
            ops.gemm_nt(a, w, bias=rn(N), residual=r, out=torch.empty(M, N, device=DEV, dtype=BF))
            alg = 2 * (M * K + N * K + 2 * M * N)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    meta.append(dict(name=name, M=M, N=N, K=K, flops=2.0 * M * N * K, algorithmic_bytes=alg, tiles=tiles, mfma_per_launch=M * N * K / (16 * 16 * 32)))
torch.cuda.synchronize()
out = os.environ.get("VLA_PMC_META")
if out:
    json.dump(meta, open(out, "w"), indent=1)
