"""World-size-2 gloo test of the data-parallel gradient exchange logic (CPU; the GPU path uses the same class
over RCCL).  Each rank holds a different flat gradient; after the bucketed all-reduce + 1/N scale both ranks hold
the mean, which is what torch DDP (finetune.py:215-227) computes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vla_adapter_amd import ddp


def test_bucket_ranges_cover_exactly():
    for n, b in ((0, 64), (7, 64), (1000, 64), (1 << 20, 1 << 18), (12345, 1000)):
        r = ddp.bucket_ranges(n, b)
        assert (r == []) if n == 0 else (r[0][0] == 0 and r[-1][1] == n)
        assert all(x[1] == y[0] for x, y in zip(r, r[1:]))
        assert all((s % 8 == 0) for s, _ in r)


def _worker(rank, world, port, q, algo="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = ddp.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(100_003, generator=g)
    red = ddp.FlatGradReducer(bucket_bytes=64 * 1024, algo=algo)
    red.reduce_async(flat, 0, 100_000)          # head region
    red.reduce_async(flat, 100_000, None)       # small tail (action queries)
    red.wait()
    flat *= red.grad_scale
    q.put((rank, flat))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_flat_grad_allreduce_world2_gloo(algo):
    """Both exchange algorithms (one all-reduce per bucket; reduce-scatter + all-gather per bucket with a ragged tail) leave the
    mean of the two ranks' gradients on BOTH ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, algo)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = (torch.randn(100_003, generator=torch.Generator().manual_seed(100)) + torch.randn(100_003, generator=torch.Generator().manual_seed(101))) / 2
    assert torch.allclose(got[0], ref, atol=1e-6) and torch.equal(got[0], got[1])


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    ddp.init_process_group_from_env("gloo")
    p = torch.randn(10_001, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    q2 = torch.randn(33, generator=torch.Generator().manual_seed(8)).to(torch.bfloat16)
    ddp.assert_ranks_in_sync([p, q2], what="identical replicas")          # same seeds on both ranks: passes
    if rank == 1:
        p[1234] += 1.0                                                    # one parameter on one rank
    try:
        ddp.assert_ranks_in_sync([p, q2], what="after the injected divergence")
        q.put((rank, "not-detected"))
    except RuntimeError as e:
        q.put((rank, "detected" if "diverged" in str(e) else str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_desync_guard_detects_one_diverged_parameter_on_every_rank():
    """ddp.assert_ranks_in_sync (finetune --sync_check_freq): identical replicas pass; one parameter changed on one rank raises on
    BOTH ranks (min / max of the checksum differ everywhere), so no rank runs on alone."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == {0: "detected", 1: "detected"}, got
