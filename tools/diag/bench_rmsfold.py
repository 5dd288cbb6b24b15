#!/usr/bin/env python3
"""Diagnostic: the LLM layer's four GEMMs with and without the folded-RMSNorm fields, and the stand-alone norm kernel (isolated)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M, D, I, W, S = 32 * 352, 896, 4864, 1152, 352
x = torch.randn(M, D, device=DEV).to(BF); ao = torch.randn(M, D, device=DEV).to(BF); hb = torch.randn(M, I, device=DEV).to(BF)
wqkv = (torch.randn(W, D, device=DEV) * .02).to(BF); bq = torch.randn(W, device=DEV).to(BF); wo = (torch.randn(D, D, device=DEV) * .02).to(BF)
wgu = (torch.randn(2 * I, D, device=DEV) * .02).to(BF); wd = (torch.randn(D, I, device=DEV) * .02).to(BF)
qkv = torch.empty(M, W, device=DEV, dtype=BF); x1 = torch.empty(M, D, device=DEV, dtype=BF); gu = torch.empty(M, 2 * I, device=DEV, dtype=BF); h = torch.empty(M, I, device=DEV, dtype=BF)
ss = torch.ones(4, M, device=DEV); rstd = torch.empty(M, device=DEV)
cos, sin = ops.rope_half_tables(S, 64, 1e6, DEV)
rope = (1, cos, sin, S, 64, 1024)
nw = torch.ones(D, device=DEV).to(BF); nb = torch.empty(M, D, device=DEV, dtype=BF)
cases = {
 "qkv": (lambda: ops.gemm_nt(x, wqkv, bias=bq, out=qkv, rope=rope), lambda: ops.gemm_nt(x, wqkv, bias=bq, out=qkv, rope=rope, rownorm=(ss, 1e-6, rstd))),
 "o": (lambda: ops.gemm_nt(ao, wo, residual=x, out=x1), lambda: ops.gemm_nt(ao, wo, residual=x, out=x1, ssq_out=ss)),
 "gate_up": (lambda: ops.gemm_nt(x1, wgu, act=ops.ACT_SWIGLU, out=gu, out2=h, c_live=(S, 288)), lambda: ops.gemm_nt(x1, wgu, act=ops.ACT_SWIGLU, out=gu, out2=h, c_live=(S, 288), rownorm=(ss, 1e-6, rstd))),
 "down": (lambda: ops.gemm_nt(hb, wd, residual=x1, out=x), lambda: ops.gemm_nt(hb, wd, residual=x1, out=x, ssq_out=ss)),
}
for k, (a, b) in cases.items():
    ta = statistics.median([timeit(a) for _ in range(3)]); tb = statistics.median([timeit(b) for _ in range(3)])
    print(f"{k:8s} plain {ta:7.1f} us   with the RMSNorm fields {tb:7.1f} us")
rs = torch.empty(M, device=DEV)
f = lambda: ops.N.check(ops._lib().vla_rmsnorm_fwd(ops._st(), ops._p(x), ops._p(nw), ops._p(nb), ops._p(rs), M, D, 1e-6), "rms")
print(f"rmsnorm kernel {statistics.median([timeit(f) for _ in range(3)]):7.1f} us")
