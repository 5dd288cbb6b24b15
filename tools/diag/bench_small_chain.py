"""In-graph time per launch of the batch-1 pass's kernels (un-profiled): N dependent copies of one launch captured into a hipGraph."""
import sys, torch
sys.path.insert(0, ".")
from vla_adapter_amd import ops
DEV, BF = "cuda", torch.bfloat16
g = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).to(BF)
N = 40
def timeit(name, fn, hint):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    ctx = ops.latency_hint() if hint else None
    if ctx: ctx.__enter__()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(N): fn()
    if ctx: ctx.__exit__()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / N * 1e3
S, D, I = 369, 896, 4864
x, xo = g(S, D), g(S, D)
w_qkv, b_qkv = g(1152, D, sc=.03), g(1152)
w_o, w_gu, w_dn = g(D, D, sc=.03), g(2 * I, D, sc=.03), g(D, I, sc=.03)
h = g(S, I)
nw = g(D)
qkv = g(1, S, 1152)
cos, sin = ops.rope_half_tables(S, 64, 1e6, DEV)
out_qkv, out_o, out_h, out_d = torch.empty(S, 1152, device=DEV, dtype=BF), torch.empty(S, D, device=DEV, dtype=BF), torch.empty(S, I, device=DEV, dtype=BF), torch.empty(S, D, device=DEV, dtype=BF)
x8, w_x, w_f = g(8, D), g(3 * D, D, sc=.03), g(D, D, sc=.03)
o8a, o8b = torch.empty(8, 3 * D, device=DEV, dtype=BF), torch.empty(8, D, device=DEV, dtype=BF)
xv, w_vqkv, w_fc1, w_fc2 = g(256, 1152), g(3456, 1152, sc=.03), g(4352, 1152, sc=.03), g(1152, 4352, sc=.03)
ov1, ov2, ov3 = torch.empty(256, 3456, device=DEV, dtype=BF), torch.empty(256, 4352, device=DEV, dtype=BF), torch.empty(256, 1152, device=DEV, dtype=BF)
hv = g(256, 4352)
cases = [
    ("rmsnorm 369x896", lambda: ops.rmsnorm_fwd(x, nw, 1e-6)),
    ("LLM qkv+rope 369x1152x896", lambda: ops.gemm_nt(x, w_qkv, bias=b_qkv, out=out_qkv, rope=(1, cos, sin, S, 64, 1024))),
    ("LLM attn S=369", lambda: ops.attn_fwd(qkv[:, :, :896], qkv[:, :, 896:1024], qkv[:, :, 1024:], 14, 2, 64, True, None)),
    ("LLM o 369x896x896 +res", lambda: ops.gemm_nt(x, w_o, residual=xo, out=out_o)),
    ("LLM gate/up swiglu 369x9728x896", lambda: ops.gemm_nt(x, w_gu, act=ops.ACT_SWIGLU, out=None)),
    ("LLM down 369x896x4864 +res (auto split)", lambda: ops.gemm_nt(h, w_dn, residual=xo, out=out_d)),
    ("LLM down, no split", lambda: ops.gemm_nt(h, w_dn, residual=xo, out=out_d, split_k=0)),
    ("head w_x 8x2688x896", lambda: ops.gemm_nt(x8, w_x, out=o8a)),
    ("head w_o 8x896x896", lambda: ops.gemm_nt(x8, w_f, residual=x8, out=o8b)),
    ("ViT qkv 256x3456x1152", lambda: ops.gemm_nt(xv, w_vqkv, out=ov1)),
    ("ViT fc1 gelu 256x4352x1152", lambda: ops.gemm_nt(xv, w_fc1, act=1, out=ov2)),
    ("ViT fc2 256x1152x4352 +res (auto split)", lambda: ops.gemm_nt(hv, w_fc2, residual=xv, out=ov3)),
    ("ViT fc2, no split", lambda: ops.gemm_nt(hv, w_fc2, residual=xv, out=ov3, split_k=0)),
]
for name, fn in cases:
    try:
        a, b = timeit(name, fn, False), timeit(name, fn, True)
        print(f"{name:45s} two-stage {a:6.1f} us   latency hint {b:6.1f} us", flush=True)
    except Exception as e:
        print(name, "FAILED", repr(e)[:200], flush=True)
