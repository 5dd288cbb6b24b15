"""Diagnostic: captured step vs eager step, B = 8 (two LLM pipelines), several repetitions (race screen)."""
import sys
import torch
sys.path.insert(0, ".")
from vla_adapter_amd import engine as E, synthetic as S
dev = "cuda:0"
cfg = E.tiny_config()
W = S.make_weights(cfg, dev, seed=3, std=0.05)
batch = S.make_batch(cfg, 8, dev, seed=100, P=40)
def eager():
    eng = E.VLAEngine(cfg, W, dev)
    return [eng.train_step(batch, 1e-3)[0].item() for _ in range(4)], eng.head.P.data.float().clone()
def graphed():
    eng = E.VLAEngine(cfg, W, dev)
    eng.capture({k: v.clone() for k, v in batch.items()}, None)
    l = [eng.train_step_graphed(1e-3)[0].item() for _ in range(4)]
    eng.flush(); torch.cuda.synchronize()
    return l, eng.head.P.data.float().clone()
le, pe = eager()
print("eager  ", le)
for i in range(4):
    lg, pg = graphed()
    print("graphed", lg, "param rel diff vs eager", ((pg - pe).norm() / pe.norm()).item())
