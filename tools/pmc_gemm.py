#!/usr/bin/env python3
"""Run a few GEMM shapes once each (for rocprofv3 --pmc collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
for (M, N, K, act) in [(11264, 9728, 896, 4), (11264, 896, 4864, 0), (8192, 4352, 1152, 1), (8192, 8192, 8192, 0)]:
    a = torch.randn(M, K, device=DEV).to(BF); w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    for tile in (2, 3):
        os.environ["VLA_GEMM_TILE"] = str(tile)
        for _ in range(3):
            if act == 4: ops.gemm_nt(a, w, act=4)
            else: ops.gemm_nt(a, w, act=act)
torch.cuda.synchronize()
