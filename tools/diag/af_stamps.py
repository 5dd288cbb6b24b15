"""Per-phase cycle medians of attn_fwd_kernel<64> on the LLM's shape (B 32, S 352, 14 / 2 heads, causal) from in-kernel s_memtime stamps
(DIAGNOSTIC build exporting vla_af_read_stamps: start / prologue done / first barrier passed / second tile's barrier / loop end / end)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import native, ops
lib = native.load()
lib.vla_af_read_stamps.argtypes = [C.c_void_p]; lib.vla_af_read_stamps.restype = C.c_int
B, S, Hq, Hkv, dh = 32, 352, 14, 2, 64
W = (Hq + 2 * Hkv) * dh
qkv = (torch.randn(B, S, W, device="cuda") * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :, :Hq * dh], qkv[:, :, Hq * dh:(Hq + Hkv) * dh], qkv[:, :, (Hq + Hkv) * dh:]
for _ in range(3):
    ops.attn_fwd(q, k, v, Hq, Hkv, dh, True, None, want_lse=True)
torch.cuda.synchronize()
n = B * Hq * 3 * 4
buf = np.zeros(32 * 16 * 4 * 4 * 8, dtype=np.uint64); assert lib.vla_af_read_stamps(buf.ctypes.data) == 0
st = buf[: n * 8].reshape(B, Hq, 3, 4, 8).astype(np.int64)
for qb in range(3):
    x = st[:, :, qb].reshape(-1, 4, 8)
    tiles = np.median(x[:, :, 7])
    pro, first, second, loop, epi = (np.median(x[:, :, 1] - x[:, :, 0]), np.median(x[:, :, 2] - x[:, :, 1]), np.median(x[:, :, 3] - x[:, :, 2]),
                                     np.median(x[:, :, 4] - x[:, :, 2]), np.median(x[:, :, 5] - x[:, :, 4]))
    print(f"q block {qb}: tiles {tiles:.0f}  prologue {pro:.0f}  to first barrier {first:.0f}  first tile {second:.0f}  loop {loop:.0f} ({loop / max(tiles, 1):.0f}/tile)  epilogue {epi:.0f}  total {np.median(x[:, :, 5] - x[:, :, 0]):.0f}")
print("kernel span cycles", int(st[..., 5].max() - st[..., 0][st[..., 0] > 0].min()))
