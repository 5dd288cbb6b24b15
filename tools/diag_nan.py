import sys, math, torch
sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S
dev = "cuda:0"; cfg = E.tiny_config(); W = S.make_weights(cfg, dev, seed=3, std=0.05); batch = S.make_batch(cfg, 8, dev, seed=100, P=40)
eng = E.VLAEngine(cfg, W, dev)
eng.capture({k: v.clone() for k, v in batch.items()}, None)
for step in range(3):
    l = eng.train_step_graphed(1e-3)
    torch.cuda.synchronize()
    P = eng.head.P
    g = P.grad.float()
    bad = []
    for name, (off, shape) in P.offsets.items():
        n = math.prod(shape)
        if not torch.isfinite(g[off:off + n]).all():
            idx = (~torch.isfinite(g[off:off + n])).nonzero().flatten()
            bad.append((name, shape, int(idx.numel()), idx[:6].tolist()))
    print(step, l.tolist(), "row0", eng._row0, "bad regions:", bad[:10])
    h = eng.head
    for nm in ("d_pf32", "dpad", "dgate", "ln_dw", "guard", "acc32"):
        t = getattr(h, nm, None)
        if t is not None: print("   ", nm, torch.isfinite(t.float()).all().item(), t.float().abs().max().item())
