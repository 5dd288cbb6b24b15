#!/usr/bin/env python3
"""Diagnostic (never on the product path): per-tile timeline of gemm256_kernel from in-kernel s_memtime stamps.
Build: `patch -p0 -d vla_adapter_amd/csrc < tools/diag/gemm256_stamps.patch`, `make -C vla_adapter_amd/csrc CXXFLAGS="... -DG256_STAMPS"`,
run this on the GPU box, `patch -R`, rebuild clean (the stamped build is ~10 % slower and never committed).
Intervals (wave 0 of every workgroup, shader cycles): top wait (setup + K-tile 1 issue + counted wait) | K loop | phase A (next tile's
K-tile 0 issue + alpha / bias / activation / pack) | phase B (stage + stores) | closing barrier.  Round 3 (profiles/r03_gemm256_stamps.txt):
K loop 2 090 - 2 130 cycles per K-tile (2 048 = the MFMA pipe), 13 - 17 k cycles per tile around it."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vla_adapter_amd import native, ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
lib = native.load()
lib.vla_g256_read_stamps.argtypes = [C.c_void_p]
lib.vla_g256_read_stamps.restype = C.c_int


def run(name, M, N, K, act, live=None, residual=False):
    a = torch.randn(M, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    bias = torch.randn(N, device=DEV).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    os.environ["VLA_GEMM_TILE"] = "6"
    if act == 4:
        out2 = torch.empty(M, N // 2, device=DEV, dtype=BF)
        fn = lambda: ops.gemm_nt(a, w, act=4, out=out, out2=out2, c_live=live)
    else:
        r = torch.randn(M, N, device=DEV).to(BF) if residual else None
        fn = lambda: ops.gemm_nt(a, w, bias=bias, act=act, residual=r, out=out, split_k=0)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    assert lib.vla_g256_clear_stamps() == 0
    fn()
    torch.cuda.synchronize()
    buf = np.zeros(256 * 16 * 8, dtype=np.uint64)
    assert lib.vla_g256_read_stamps(buf.ctypes.data) == 0
    st = buf.reshape(256, 16, 8).astype(np.int64)
    tiles = -(-M // 256) * -(-N // 256)
    iters = min(15, tiles // 256)               # full rounds only
    names = ["top wait", "K loop", "phase A", "phase B", "barrier"]
    print(f"{name}: {M}x{N}x{K}  tiles {tiles}  K-tiles {K // 64}")
    for it in range(max(iters, 1)):
        d = st[:, it, 1:6] - st[:, it, 0:5]
        ok = (st[:, it, 4] > 0)
        med = np.median(d[ok], axis=0)
        tot = np.median(st[ok, it, 4] - st[ok, it, 0])
        nxt = np.median(st[ok, it + 1, 0] - st[ok, it, 0]) if it + 1 < 16 and (st[:, it + 1, 0] > 0).any() else float("nan")
        k10 = np.median(st[ok, it, 6] - st[ok, it, 1]) if (st[ok, it, 6] > 0).any() else float("nan")      # (deferred-h build: first ten K-tiles)
        print(f"  tile {it}: " + "  ".join(f"{n} {int(m):6d}" for n, m in zip(names, med)) + f"   | tile total {int(tot)} cycles, next top after {nxt}  first-10-K-tiles {k10}")


if not os.environ.get("ONLY_GATE_UP"):
    run("vit qkv", 32 * 256, 3456, 1152, 0)
    run("sq 8192 K1024", 8192, 8192, 1024, 0)
    run("sq 8192 K1024 +res", 8192, 8192, 1024, 0, residual=True)
run("llm gate_up live", 32 * 352, 9728, 896, 4, live=(352, 288))
run("llm gate_up full", 32 * 352, 9728, 896, 4)
