"""Data-parallel gradient exchange over RCCL/xGMI (one process per GPU; torch.distributed backend "nccl" IS RCCL).

Reference behaviour (vla-scripts/finetune.py:215-227, 869, 284): torch DDP sum-all-reduces every trainable
gradient in 25 MiB buckets and divides by the world size; every rank draws its own samples (no DistributedSampler).
Divergence: the reference calls ``action_head.module.predict_action`` and therefore never synchronises
the head's gradients; this build all-reduces them (mathematically correct DP).

MI355X-first design: all trainable parameters live in ONE flat bf16 buffer (engine.FlatParams), so the exchange is
a handful of large collectives instead of thousands of small ones.  In the captured step the head/proprio gradients
(437 MB) are final when the head stream ends - before the LLM backward and the next step's vision stage have
finished: their all-reduce is launched from that event on its own stream and runs underneath them.  The 64x896
action-query gradient (final when the LLM backward ends, which is earlier) is reduced FIRST: it is all the next step's
LLM stream waits for, while the big exchange keeps running beside that step's first forward segment and is joined by the
head stream only.  The update itself is applied at the start of the next step.
Bucket size defaults to 64 MiB (xGMI is point-to-point, 7 links x ~153 GB/s: large messages amortise the per-
collective latency; ring all-reduce is per-link bound).  The 1/N scale is folded into the AdamW kernel.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:   # VLA_DIST_BACKEND=gloo: functional rehearsal of the multi-rank path where RCCL cannot run
            backend = os.environ.get("VLA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def bucket_ranges(numel: int, bucket_elems: int, align: int = 8) -> List[Tuple[int, int]]:
    """Split [0, numel) into contiguous ranges of at most ``bucket_elems`` (multiples of ``align``)."""
    assert numel >= 0 and bucket_elems > 0
    step = max(align, bucket_elems // align * align)
    return [(s, min(numel, s + step)) for s in range(0, numel, step)]


class FlatGradReducer:
    """Sum-all-reduce of a flat gradient buffer in large buckets, optionally on a side stream (overlap).

    algo="allreduce" (default): one ``all_reduce`` per bucket, algorithm left to RCCL (a ring is bound by ONE xGMI link:
    ~153 GB/s -> 437 MB x 2 x 7/8 / 153 GB/s ~ 5 ms at 8 GPUs).
    algo="rs_ag": every bucket as ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` - each rank sends 1/N of the bucket to
    each of its N-1 peers directly, all 7 links of the fully connected xGMI node busy at once (~0.7 ms).  The
    same sum in a different association order than a ring, so results may differ from "allreduce" in the last bf16 bit; bucket
    bounds are aligned to N x 8 elements, the ragged tail of the buffer goes through a small all_reduce.  UNMEASURED on
    hardware (no multi-GPU node in this round); covered functionally by tests/test_ddp_cpu.py on gloo."""

    def __init__(self, group=None, bucket_bytes: int = 64 << 20, algo: str = "allreduce"):
        assert algo in ("allreduce", "rs_ag")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.algo = algo
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._pending = False
        self._shards = {}

    def _reduce_bucket(self, t: torch.Tensor):
        """In-place sum over the ranks of the contiguous 1-D tensor ``t``."""
        N = self.world
        if self.algo == "allreduce" or N == 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return
        n = t.numel() // (8 * N) * (8 * N)              # equal 16-B aligned shards
        if n:
            key = (n // N, t.dtype, str(t.device))
            shard = self._shards.get(key)
            if shard is None:
                shard = self._shards[key] = torch.empty(n // N, dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(shard, t[:n], op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(t[:n], shard, group=self.group)
        if n < t.numel():
            dist.all_reduce(t[n:], op=dist.ReduceOp.SUM, group=self.group)

    def reduce_async(self, flat: torch.Tensor, start: int = 0, end: Optional[int] = None, after_event=None):
        """Launch the exchange of flat[start:end] after everything already enqueued on the current stream - or, with
        ``after_event``, as soon as that event fires (the slice was produced on another stream and is final there)."""
        if self.world == 1:
            return None
        end = flat.numel() if end is None else end
        view = flat[start:end]
        ranges = bucket_ranges(view.numel(), self.bucket_bytes // view.element_size(), align=8 * (self.world if self.algo == "rs_ag" else 1))
        if self.stream is not None and flat.is_cuda:
            if after_event is not None:
                self.stream.wait_event(after_event)
            else:
                self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for a, b in ranges:
                    self._reduce_bucket(view[a:b])
                done = torch.cuda.Event()
                done.record()
            self._pending = True
            return done        # fires when THIS slice is reduced (later slices queue behind it on the same stream)
        for a, b in ranges:
            self._reduce_bucket(view[a:b])
        return None

    def wait(self, *streams):
        """Make the given streams (default: the current one) wait for the side-stream collectives (no host sync)."""
        if self._pending:
            for s in (streams or (torch.cuda.current_stream(),)):
                s.wait_stream(self.stream)
            self._pending = False

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def params_checksum(flats) -> torch.Tensor:
    """int64 [2]: (sum of the 16-bit patterns of every parameter, number of parameters) over the given flat bf16 buffers.  An
    integer sum: exact, order-independent - two ranks holding the same parameters produce the same pair."""
    dev = flats[0].device
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    for t in flats:
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
        acc[0] += t.view(torch.int16).to(torch.int32).sum(dtype=torch.int64)
        acc[1] += t.numel()
    return acc


def assert_ranks_in_sync(flats, what: str = "parameters", group=None) -> None:
    """Desync guard of a data-parallel run (every rank must hold the same replica after every update: the reference relies on
    DDP's construction-time broadcast and identical all-reduced gradients, vla-scripts/finetune.py:869).  The ranks exchange the
    MIN and the MAX of a parameter checksum (two 16-byte all-reduces); any difference raises on every rank.  Off the hot path:
    called every --sync_check_freq optimizer steps (one host sync)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    c = params_checksum(flats)
    if dist.get_backend(group) == "gloo":
        c = c.cpu()
    lo, hi = c.clone(), c.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"data-parallel ranks diverged: {what} differ between ranks (checksum min {lo.tolist()} != max {hi.tolist()}, "
                           f"this rank {c.tolist()})")
