"""Functional rehearsal of the multi-rank LoRA / full fine-tune step on ONE GPU (gloo: RCCL refuses two ranks per device), launched
with torch.distributed.run --nproc-per-node 2.  VLA_TRAINER=full|lora, VLA_CAPTURED=1 for the segment-graph replay.
Checks (as tools/ddp_rehearsal.py does for the adapter-only engine): every rank ends with identical parameters AND these equal a
single-process run that computes both ranks' gradients itself, adds them in bf16 and applies AdamW with the 1/N scale - with the
exchange started range by range DURING the backward (trainers.BackboneTrainer._backward_gen; vla-scripts/finetune.py:215-227, 869).
UNMEASURED on multi-GPU hardware: this is a correctness rehearsal of the schedule, not of the links."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
os.environ.setdefault("VLA_DIST_BACKEND", "gloo")
from vla_adapter_amd import ddp, engine as E, ops, synthetic as S  # noqa: E402
from vla_adapter_amd.trainers import FullFinetune, LoRAFinetune  # noqa: E402


def log(*a):
    sys.stdout.write(f"[rank {os.environ.get('RANK')}] " + " ".join(str(x) for x in a) + "\n")
    sys.stdout.flush()


def make(cfg, W, dev, mode):
    eng = E.VLAEngine(cfg, W, dev)
    return eng, (FullFinetune(eng) if mode == "full" else LoRAFinetune(eng, rank=8, seed=5))


rank, local, world = ddp.init_process_group_from_env()
torch.cuda.set_device(0)
dev, mode, captured = "cuda:0", os.environ.get("VLA_TRAINER", "full"), bool(int(os.environ.get("VLA_CAPTURED", "0")))
cfg = E.tiny_fused_config()
W = S.make_weights(cfg, dev, seed=3, std=0.05)
batches = [S.make_batch(cfg, 2, dev, seed=100 + r, P=24) for r in range(world)]
lr, steps = 1e-3, 3
eng, tr = make(cfg, W, dev, mode)
eng.reducer = ddp.FlatGradReducer(bucket_bytes=1 << 16, algo=os.environ.get("VLA_DDP_ALGO", "allreduce"))   # small buckets: many collectives per range
n_ex = [0]
orig = tr._exchange
tr._exchange = lambda ranges, after_event=None: (n_ex.__setitem__(0, n_ex[0] + sum(1 for _, lo, hi in ranges if hi > lo)), orig(ranges, after_event=after_event))[1]
if captured:
    tr.capture(batches[rank], None)
    log("captured", len(tr._segs), "segment graphs")
for it in range(steps):
    l = tr.train_step_graphed(lr) if captured else tr.train_step(batches[rank], lr)
torch.cuda.synchronize()
log("loss", l[0].item(), "gradient ranges handed to the exchange per step:", n_ex[0] // steps)
assert n_ex[0] // steps >= 4, "the exchange must start range by range during the backward, not once at its end"
for name, buf in (("vlm", tr.P.data), ("head", eng.head.P.data)):
    p = buf.float()
    ref = p.clone()
    dist.all_reduce(ref)
    err = (p - ref / world).abs().max().item()
    log(f"max |{name} param - mean over ranks| =", err)
    assert err == 0.0, "ranks diverged"
# single-process reference: both ranks' gradients from the SAME parameters, summed in bf16, AdamW with gscale = 1 / world
eng2, ref = make(cfg, W, dev, mode)
for it in range(steps):
    gs = []
    for b in batches:
        ref.backward(ref.forward(b, None), b["actions"])
        gs.append((ref.P.grad.clone(), eng2.head.P.grad.clone()))
    for k, P in enumerate((ref.P, eng2.head.P)):
        P.grad.copy_(gs[0][k])
        for g in gs[1:]:
            ops.add_(P.grad, g[k])
        ops.adamw_(P.data, P.grad, P.m, P.v, it + 1, lr, gscale=1.0 / world)
    eng2.head.dirty = True
    ref.refresh()
torch.cuda.synchronize()
for name, a, b in (("vlm", tr.P.data, ref.P.data), ("head", eng.head.P.data, eng2.head.P.data)):
    rel = ((a.float() - b.float()).norm() / b.float().norm()).item()
    log(f"rel |{name} param(DDP) - param(single process, summed gradients)| =", rel)
    assert rel <= 1e-3, "the two-rank run must reproduce the single-process run on the mean gradient (up to the fp32 atomic order of the bias reductions)"
dist.barrier()
dist.destroy_process_group()
log("ranks-in-sync-ok")
