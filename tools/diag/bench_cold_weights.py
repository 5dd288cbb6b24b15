"""Cold-weight cost of the batch-1 products: the same launch chain over (a) two alternating weights (hot) and (b) a rotation of > 256 MB
of weights (cold: HBM), two-stage and deep operand ring.  (Round 4 also measured a touch kernel warming the memory-side cache ahead of
each launch, on a side stream and in-stream: 23-29 / 12-19 us per launch against 10-14 cold - dropped, the entry point removed.)"""
import sys, torch
sys.path.insert(0, ".")
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
g = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).to(BF)
def run(name, M, N, K, ncopy, hint, mode):
    x = g(M, K)
    W = g(ncopy, N, K, sc=.03)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    st = torch.cuda.Stream()
    def chain():
        for i in range(ncopy):
            ops.gemm_nt(x, W[i], out=out)
    with torch.cuda.stream(st):
        chain()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    ctx = ops.latency_hint() if hint else None
    if ctx: ctx.__enter__()
    with torch.cuda.graph(gr, stream=st):
        chain()
    if ctx: ctx.__exit__()
    for _ in range(2): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / ncopy * 1e3
for name, M, N, K in [("LLM gate/up 369x9728x896", 369, 9728, 896), ("LLM qkv 369x1152x896", 369, 1152, 896), ("ViT fc1 256x4352x1152", 256, 4352, 1152),
                      ("head w_x 8x2688x896", 8, 2688, 896)]:
    per = N * K * 2
    ncold = max(8, (400 << 20) // per)
    ncold = min(ncold, 160)
    hot = run(name, M, N, K, 2, True, "none")
    cold2 = run(name, M, N, K, ncold, False, "none")
    cold = run(name, M, N, K, ncold, True, "none")
    print(f"{name:28s} weights {per/1e6:5.1f} MB x {ncold:3d}: hot {hot:5.1f} | cold two-stage {cold2:5.1f} | cold deep {cold:5.1f} us/launch", flush=True)
