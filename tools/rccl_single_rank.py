"""RCCL path rehearsal on ONE GPU: a one-rank "nccl" process group, with the gradient reducer forced to issue its
collectives (a one-rank all-reduce is the identity).  Exercises what the gloo rehearsal cannot: RCCL communicator creation
next to the captured segment graphs, collectives on the reducer's own stream between graph replays, the per-slice
completion events that flush() hands to the LLM and head streams.  Checks the result against a run without the exchange."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from vla_adapter_amd import ddp, engine as E, synthetic as S  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = "cuda:0"
cfg = E.tiny_config()
W = S.make_weights(cfg, dev, seed=3, std=0.05)
batch = S.make_batch(cfg, 8, dev, seed=100, P=40)


def run(with_exchange: bool):
    eng = E.VLAEngine(cfg, W, dev)
    if with_exchange:
        eng.reducer = ddp.FlatGradReducer()
        eng.reducer.world = 2                      # force the collectives of a multi-rank job ...
        type(eng.reducer).grad_scale = property(lambda self: 1.0)   # ... a one-rank sum is the identity: no averaging
    eng.capture({k: v.clone() for k, v in batch.items()}, None)
    losses = [eng.train_step_graphed(1e-3)[0].item() for _ in range(4)]
    eng.flush()
    torch.cuda.synchronize()
    return losses, eng.head.P.data.float().clone()


l0, p0 = run(False)
l1, p1 = run(True)
print("losses without / with the RCCL exchange:", l0, l1)
assert l0 == l1 and torch.equal(p0, p1), "the one-rank exchange must not change the result"
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("rccl-single-rank-ok")
