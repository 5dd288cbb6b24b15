"""In-graph time per launch of the LoRA fine-tune's low-rank products t = 2 x A_cat^T / dt = 2 dy B_blk (batch 16, config-2 backbone):
gemm_skinny.hip against the 128-row tiles (VLA_NO_SKINNY=1), split-K as ops.gemm_nt picks it."""
import os, sys, torch
sys.path.insert(0, ".")
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
g = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).to(BF)
NREP = 20
def timeit(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(NREP): fn()
    for _ in range(2): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / NREP * 1e3
shapes = [("ViT t qkv", 4096, 192, 1152), ("ViT t proj/fc1, dt proj/fc2", 4096, 64, 1152), ("ViT t fc2, dt fc1", 4096, 64, 4352), ("ViT dt qkv", 4096, 192, 3456),
          ("LLM t qkv", 5632, 192, 896), ("LLM t o, dt o/down", 5632, 64, 896), ("LLM t gate/up", 5632, 128, 896), ("LLM t down", 5632, 64, 4864),
          ("LLM dt qkv", 5632, 192, 1152), ("LLM dt gate/up", 5632, 128, 9728)]
for name, M, N, K in shapes:
    a, b, out = g(M, K), g(N, K, sc=.05), torch.empty(M, N, device=DEV, dtype=BF)
    fn = lambda: ops.gemm_nt(a, b, alpha=2.0, out=out)
    os.environ.pop("VLA_NO_SKINNY", None)
    t1 = timeit(fn)
    os.environ["VLA_NO_SKINNY"] = "1"
    t0 = timeit(fn)
    os.environ.pop("VLA_NO_SKINNY", None)
    fn0 = lambda: ops.gemm_nt(a, b, alpha=2.0, out=out, split_k=0)
    t2 = timeit(fn0)
    print(f"{name:30s} {M}x{N}x{K}: 128-row tiles (auto split) {t0:6.1f} us | default routing {t1:6.1f} us | skinny kernel, no split {t2:6.1f} us   [A pass at 5 TB/s: {M*K*2/5e6:5.1f} us]", flush=True)
