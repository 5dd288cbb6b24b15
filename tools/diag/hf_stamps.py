"""Per-phase cycle medians of head_fwd_mfma from in-kernel s_memtime stamps (DIAGNOSTIC build exporting vla_hf_read_stamps: stamps at
kernel start / after the LDS zero-fill / after the first prefetch is issued / after the first tile is in LDS / after the tile loop /
after the merge barrier / after the merge / end; per wave)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import native, ops
lib = native.load()
lib.vla_hf_read_stamps.argtypes = [C.c_void_p]; lib.vla_hf_read_stamps.restype = C.c_int
B, T, Ka, Kt, H, D = 32, 8, 65, 256, 8, 896
g = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(torch.bfloat16)
x3, a2, t2 = g(B, T, 3 * D), g(B, Ka, 2 * D), g(B, Kt, 2 * D)
gate = torch.tensor([0.7], device="cuda").to(torch.bfloat16)
args = (x3[:, :, :D], x3[:, :, D:2 * D], x3[:, :, 2 * D:], a2[:, :, :D], a2[:, :, D:], t2[:, :, :D], t2[:, :, D:])
for _ in range(5):
    ops.head_attn_fwd(*args, gate, H)
torch.cuda.synchronize()
buf = np.zeros(256 * 4 * 8, dtype=np.uint64); assert lib.vla_hf_read_stamps(buf.ctypes.data) == 0
st = buf.reshape(256, 4, 8).astype(np.int64)
names = ["zero-fill", "prefetch issue", "first tile in LDS", "tile loop", "merge barrier", "merge", "store"]
for w in range(4):
    d = np.diff(st[:, w, :], axis=1)
    med = np.median(d, axis=0)
    tot = np.median(st[:, w, 7 if w == 0 else 5] - st[:, w, 0])
    print(f"wave {w}: " + "  ".join(f"{n} {int(m):6d}" for n, m in zip(names, med)) + f"  | total {int(tot)}")
print("kernel span (first start -> last end), cycles:", int(st[:, 0, 7].max() - st[:, :, 0].min()))
