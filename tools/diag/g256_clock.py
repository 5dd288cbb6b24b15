"""Shader clock a gemm256 launch holds: a DIAGNOSTIC build (not in the tree: kernel start/end s_memtime + s_memrealtime of every
workgroup written behind the h output) run on the gate/up shape.  Prints cycles per workgroup, wall time, MHz."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
M, N, K = 32 * 352, 9728, 896
a = torch.randn(M, K, device=DEV).to(BF); w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
out = torch.empty(M, N, device=DEV, dtype=BF)
buf = torch.zeros(M * (N // 2) + 4096, device=DEV, dtype=BF)
out2 = buf[: M * (N // 2)].view(M, N // 2)
for live in ((352, 288), None):
    for _ in range(5):
        ops.gemm_nt(a, w, act=4, out=out, out2=out2, c_live=live)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_nt(a, w, act=4, out=out, out2=out2, c_live=live)
    e1.record(); torch.cuda.synchronize()
    st = buf[M * (N // 2):].view(torch.int64).cpu().numpy()[:512].reshape(256, 2)
    cyc, rt, ndef = st[:, 0], st[:, 1] & ((1 << 48) - 1), st[:, 1] >> 48
    print(f"live={live}: {e0.elapsed_time(e1) * 100:.1f} us/launch | WG cycles median {np.median(cyc):.0f} max {cyc.max()} | realtime ticks median {np.median(rt):.0f} "
          f"(100 MHz -> {np.median(rt) / 100:.1f} us) | shader clock {np.median(cyc) / (np.median(rt) / 100):.0f} MHz | deferred tiles per WG {np.median(ndef):.0f}")
