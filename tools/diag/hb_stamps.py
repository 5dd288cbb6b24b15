"""Per-phase cycle medians of head_bwd_tiles from in-kernel s_memtime stamps.

Needs a DIAGNOSTIC build of libvla_native.so that stamps the kernel and exports vla_hb_read_stamps (not part of the
product ABI; the stamped kernel is not kept in the tree - add `s_memtime` reads at the phase boundaries of head_bwd_tiles
into a __device__ array and a host entry that copies it out).  Used once in round 3: 33k of 70k cycles sat in the
row-per-lane dK/dV stores, which is what moved them behind an LDS transpose.
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import native, ops
lib = native.load()
lib.vla_hb_read_stamps.argtypes = [C.c_void_p]; lib.vla_hb_read_stamps.restype = C.c_int
DEV, BF = "cuda", torch.bfloat16
B, T, Ka, Kt, D, H = 32, 8, 65, 256, 896, 8
x3 = (torch.randn(B, T, 3 * D, device=DEV) * .3).to(BF); a2 = (torch.randn(B, Ka, 2 * D, device=DEV) * .3).to(BF); t2 = (torch.randn(B, Kt, 2 * D, device=DEV) * .3).to(BF)
gate = torch.tensor([0.7]).to(BF).to(DEV); dout = (torch.randn(B, T, D, device=DEV)).to(BF)
args = (x3[:, :, :D], x3[:, :, D:2 * D], x3[:, :, 2 * D:], a2[:, :, :D], a2[:, :, D:], t2[:, :, :D], t2[:, :, D:])
out, probs = ops.head_attn_fwd(*args, gate, H)
tabs = ops.rope_inter_tables(max(T, Ka, Kt), D // H, DEV)
g3, ga, gt = torch.zeros_like(x3), torch.zeros_like(a2), torch.zeros_like(t2); dg = torch.zeros(1, device=DEV)
for _ in range(5):
    ops.head_attn_bwd(dout, out, *args, gate, probs, dg, g3[:, :, :D], g3[:, :, D:2 * D], g3[:, :, 2 * D:], ga[:, :, :D], ga[:, :, D:], gt[:, :, :D], gt[:, :, D:], H, rope=tabs)
torch.cuda.synchronize()
buf = np.zeros(256 * 16, dtype=np.uint64); assert lib.vla_hb_read_stamps(buf.ctypes.data) == 0
st = buf.reshape(256, 16).astype(np.int64)[:, :9]
d = np.diff(st, axis=1); med = np.median(d, axis=0)
names = ["loads issued + staging", "barrier", "S, dP MFMAs", "score VALU", "dV / dK blocks + stores", "S^T, dP^T MFMAs", "score VALU 2", "dQ blocks + stores"]
for n, m in zip(names, med): print(f"{n:28s} {int(m):7d} cycles")
print("total", int(np.median(st[:, 8] - st[:, 0])), " spread of start times across WGs (cycles):", int(st[:, 0].max() - st[:, 0].min()), " end spread", int(st[:, 8].max() - st[:, 8].min()))
