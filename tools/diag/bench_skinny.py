import os, sys, statistics, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
def timeit(fn, iters=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K in [("llm t qkv", 5632, 192, 896), ("llm t o", 5632, 64, 896), ("llm t gate_up", 5632, 128, 896), ("llm t down", 5632, 64, 4864),
                      ("llm dt gate_up", 5632, 128, 9728), ("llm dt qkv", 5632, 192, 1152), ("vit t qkv", 4096, 192, 1152), ("vit t fc1", 4096, 64, 1152), ("vit t fc2", 4096, 64, 4352), ("vit dt fc1", 4096, 64, 4352)]:
    a = torch.randn(M, K, device=DEV).to(BF); b = torch.randn(N, K, device=DEV).to(BF); out = torch.empty(M, N, device=DEV, dtype=BF)
    line = f"{name:16s} {M}x{N}x{K}"
    for sk in (0, 2, 4, 7, 8, None):
        if sk and K % (64 * sk): continue
        try:
            t = statistics.median([timeit(lambda: ops.gemm_nt(a, b, alpha=2.0, out=out, split_k=sk)) for _ in range(3)])
            line += f" | sk={sk}: {t:5.1f}us"
        except Exception as e:
            line += f" | sk={sk}: err"
    print(line, flush=True)
