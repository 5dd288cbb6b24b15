#!/usr/bin/env python3
"""Ablation of the dK/dV attention-backward kernel (timing only): VLA_DKV_DBG bit0 skip LDS reduction, bit1 skip
compute, bit2 one iteration."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import ops
from tools.bench_kernels import timeit
B, S, Hq, Hkv, dh = 32, 352, 14, 2, 64
W = (Hq + 2 * Hkv) * dh
qkv = torch.randn(B, S, W, device="cuda").bfloat16()
q, k, v = qkv[:, :, :Hq * dh], qkv[:, :, Hq * dh:(Hq + Hkv) * dh], qkv[:, :, (Hq + Hkv) * dh:]
o, lse = ops.attn_fwd(q, k, v, Hq, Hkv, dh, True, None, want_lse=True)
do = torch.randn_like(o)
for dbg in (0, 1, 2, 3, 4, 6, 7):
    os.environ["VLA_DKV_DBG"] = str(dbg)
    t = timeit(lambda: ops.attn_bwd(do, q, k, v, o, lse, Hq, Hkv, dh, True, None))
    print(f"dbg={dbg} (skip_reduce={dbg&1} skip_compute={(dbg>>1)&1} one_iter={(dbg>>2)&1}): attn_bwd total {t*1e6:.1f} us", flush=True)
