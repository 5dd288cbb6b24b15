"""Un-profiled in-graph time of the pieces of the batch-1 pass (config2 backbone): LLM layers alone, ViT alone, head blocks alone."""
import sys, torch
sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S, ops
from vla_adapter_amd.modeling_prismatic import OpenVLAForActionPrediction
dev = "cuda"
cfg = E.config2()
vla = OpenVLAForActionPrediction(cfg, S.make_weights(cfg, dev, seed=0), dev)
ids = torch.randint(0, 151000, (1, 48))
px = torch.randn(1, 3 * len(cfg.vit) * cfg.n_img, 224, 224).clamp_(-3, 3).to(torch.bfloat16)
i2, am, lab = vla.prepare_inference_inputs(ids.to(dev), torch.ones_like(ids, dtype=torch.bool).to(dev))
batch = dict(input_ids=i2, labels=lab, attention_mask=am.bool(), pixel_values=px.to(dev), proprio=torch.zeros(1, 8, device=dev))
eng = vla.engine
for _ in range(3):
    eng.predict(batch)
torch.cuda.synchronize()
llm, head = eng.llm, eng.head
def timeg(fn, hint=True, n=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    ctx = ops.latency_hint() if hint else None
    if ctx: ctx.__enter__()
    with torch.cuda.graph(gr, stream=st):
        fn()
    if ctx: ctx.__exit__()
    for _ in range(2): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
nl = cfg.llm.n_layers
for hint in (False, True):
    t = timeg(lambda: [llm.fwd_layer(i) for i in range(nl)], hint)
    print(f"hint={hint}: LLM {nl} layers {t:8.1f} us = {t/nl:6.1f} us/layer")
    t = timeg(lambda: eng._vision_backbone(0, batch), hint)
    print(f"hint={hint}: vision backbone 0 {t:8.1f} us")
    t = timeg(lambda: [head.fwd_layer(i) for i in range(cfg.num_blocks)], hint)
    print(f"hint={hint}: head {cfg.num_blocks} blocks {t:8.1f} us = {t/cfg.num_blocks:6.1f} us/block")
# pieces of one LLM layer
i = 5
c = llm.cfg; D = c.d; Sq = llm.S
L = llm.layers[i]
x = llm.HS[i].view(-1, D)
pieces = {
  "rms": lambda: llm._rms(x, L["n1"], llm.nbuf, llm.R1[i]),
  "qkv": lambda: ops.gemm_nt(llm.nbuf, L["wqkv"], bias=L["bqkv"], out=llm.QKV[i], rope=(1, llm.cos, llm.sin, Sq, c.dh, (c.heads + c.kv_heads) * c.dh)),
  "attn": lambda: llm._attn_fwd(llm.QKV[i].view(1, Sq, -1), i, 0, 1, Sq),
  "o": lambda: ops.gemm_nt(llm.AO[i], L["wo"], residual=x, out=llm.X1[i]),
  "gu": lambda: ops.gemm_nt(llm.nbuf, L["wgu"], act=ops.ACT_SWIGLU, out=llm.GU[i], out2=llm.hbuf),
  "gu(no pre-act store)": lambda: ops.gemm_nt(llm.nbuf, L["wgu"], act=ops.ACT_SWIGLU, out=None, out2=llm.hbuf),
  "down": lambda: ops.gemm_nt(llm.hbuf, L["wd"], residual=llm.X1[i], out=llm.HS[llm.out_slot(i)].view(-1, D)),
}
for k, fn in pieces.items():
    t = timeg(lambda: [fn() for _ in range(20)], True)
    print(f"  LLM layer piece {k:24s} {t/20:6.1f} us (x20 chain, hot)")
