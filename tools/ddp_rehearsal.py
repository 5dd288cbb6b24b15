"""Functional rehearsal of the multi-rank captured step on ONE GPU (RCCL refuses two ranks per device, so the
collectives go through gloo): launched with torch.distributed.run --nproc-per-node 2.  Checks that every rank ends
with identical parameters AND that these equal a single-process run that computes both ranks' gradients itself, adds them
and applies AdamW with the 1/N scale (what synchronous data parallelism means; the reference's DDP, finetune.py:215-227).
VLA_DDP_ALGO=rs_ag runs the exchange as reduce-scatter + all-gather."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
os.environ.setdefault("VLA_DIST_BACKEND", "gloo")
from vla_adapter_amd import ddp, engine as E, synthetic as S  # noqa: E402


def log(*a):
    sys.stdout.write(f"[rank {os.environ.get('RANK')}] " + " ".join(str(x) for x in a) + "\n")   # one write: lines stay whole
    sys.stdout.flush()


rank, local, world = ddp.init_process_group_from_env()
torch.cuda.set_device(0)
dev = "cuda:0"
log("group up", world)
t = torch.ones(1000, device=dev, dtype=torch.bfloat16) * (rank + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
log("plain all_reduce of a bf16 cuda tensor:", t[0].item())
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dist.all_reduce(t)
torch.cuda.synchronize()
log("side-stream all_reduce:", t[0].item())
cfg = E.tiny_config()
W = S.make_weights(cfg, dev, seed=3, std=0.05)
eng = E.VLAEngine(cfg, W, dev)
eng.reducer = ddp.FlatGradReducer(algo=os.environ.get("VLA_DDP_ALGO", "allreduce"))
batches = [S.make_batch(cfg, 2, dev, seed=100 + r, P=40) for r in range(world)]
batch = batches[rank]
eng.capture(batch, None)
log("captured")
for it in range(3):
    l = eng.train_step_graphed(1e-3)
    log("step", it, "enqueued")
eng.flush()
torch.cuda.synchronize()
log("loss", l[0].item())
p = eng.head.P.data.float()
ref = p.clone()
dist.all_reduce(ref)
ref /= world
err = (p - ref).abs().max().item()
log("max |param - mean over ranks| =", err)
assert err == 0.0, "ranks diverged"
# single-process reference: both ranks' gradients from the SAME parameters, summed in bf16, AdamW with gscale = 1/world
from vla_adapter_amd import ops  # noqa: E402
ref_eng = E.VLAEngine(cfg, W, dev)
P = ref_eng.head.P
for it in range(3):
    gs = []
    for b in batches:
        pred = ref_eng.forward(b, None, for_training=True)
        ref_eng.loss_and_backward(pred, b["actions"])
        gs.append(P.grad.clone())
    P.grad.copy_(gs[0])
    for g in gs[1:]:
        ops.add_(P.grad, g)
    ops.adamw_(P.data, P.grad, P.m, P.v, it + 1, 1e-3, gscale=1.0 / world)
    ref_eng.head.dirty = True
torch.cuda.synchronize()
pr = P.data.float()
rel = ((p - pr).norm() / pr.norm()).item()
log("rel |param(DDP) - param(single process, summed gradients)| =", rel)
assert rel <= 1e-3, "the two-rank run must reproduce the single-process run on the mean gradient (up to the fp32 atomic order of the bias reductions)"
dist.barrier()
dist.destroy_process_group()
log("ranks-in-sync-ok")
