"""Calibration only (never on the product path): what does the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on
the step's hot shapes?  Sets the headroom for gemm_nt_kernel."""
import sys
import time

import torch

sys.path.insert(0, ".")
from vla_adapter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


B = 32
shapes = [("llm qkv", B * 352, 1152, 896), ("llm o", B * 352, 896, 896), ("llm gate_up", B * 352, 9728, 896),
          ("llm down", B * 352, 896, 4864), ("vit qkv", B * 256, 3456, 1152), ("vit proj", B * 256, 1152, 1152),
          ("vit fc1", B * 256, 4352, 1152), ("vit fc2", B * 256, 1152, 4352), ("head task kv", B * 256, 1792, 896),
          ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    t_mine = timeit(lambda: ops.gemm_nt(a, w, out=out))
    wt = w.t()
    t_vend = timeit(lambda: torch.matmul(a, wt, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:14s} {M:5d}x{N:4d}x{K:4d} | gemm_nt {t_mine*1e6:7.1f} us {fl/t_mine/1e12:6.0f} TF | vendor {t_vend*1e6:7.1f} us {fl/t_vend/1e12:6.0f} TF", flush=True)
