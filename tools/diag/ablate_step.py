#!/usr/bin/env python3
"""Diagnostic, timing only (results are WRONG by construction): the captured training step with one kernel family compiled out of
the graphs.  CAUTION - NOT an upper bound as it stands: a skipped producer leaves its output buffer at zeros (or at whatever it held),
and GEMMs on zero operands run the chip at a higher clock (MI355X_MICROARCH.md, DVFS): "no RMSNorm" read -1.2 ms this way while the
real fold of those norms into their GEMMs measured +0.1 ms.  The buffers of the skipped producers are therefore filled with random data
once (rms, attn, head_fwd, head_bwd - the fills happen during the eager warm-up / capture pass, not in the timed replays).
usage: ablate_step.py {none|rms|ln|attn|head_fwd|head_bwd|nostore} [bench.py args]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
what = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
from vla_adapter_amd import engine as E, ops  # noqa: E402

_buf = {}


def _like(key, shape, dtype, dev):
    k = (key, tuple(shape), dtype)
    if k not in _buf:
        _buf[k] = (torch.randn(shape, device=dev) * 0.5).to(dtype)      # never zeros: see the caution above
    return _buf[k]


_alloc0 = E.LLM._alloc


def _alloc_random(self, B, S):
    fresh = self._buf_key != (B, S)
    _alloc0(self, B, S)
    if fresh:                                                            # buffers whose producer is skipped hold random data
        self.nbuf.copy_((torch.randn(self.nbuf.shape, device=self.nbuf.device) * 0.5).to(self.nbuf.dtype))
        self.AO.copy_((torch.randn(self.AO.shape, device=self.AO.device) * 0.5).to(self.AO.dtype))
        self.R1.fill_(1.0); self.R2.fill_(1.0)
        if what == "nostore":                                             # (a library built without the 256-row kernel's stores, through VLA_NATIVE_LIB)
            for t in (self.HS, self.X1, self.QKV, self.GU, self.hbuf):
                t.copy_((torch.randn(t.shape, device=t.device) * 0.5).to(t.dtype))


E.LLM._alloc = _alloc_random


if what == "rms":
    E.LLM._rms = lambda self, x, w, out, rstd: None
elif what == "ln":
    _ln = ops.layernorm_fwd

    def ln(x, w, b, eps, want_stats=False, **kw):
        if want_stats:
            return _ln(x, w, b, eps, want_stats=True, **kw)
        return x                                              # (ViT blocks: the GEMM reads the un-normalised rows)
    ops.layernorm_fwd = ln
elif what == "attn":
    _af = ops.attn_fwd

    def af(q, k, v, Hq, Hkv, dh, causal, kmask=None, want_lse=False, **kw):
        o = _like("o", (q.shape[0], q.shape[1], Hq * dh), q.dtype, q.device)
        if want_lse:
            return o, _like("lse", (q.shape[0], Hq, q.shape[1]), torch.float32, q.device)
        return o
    ops.attn_fwd = af
    E.LLM._attn_fwd = lambda self, q3, i, b0, b1, S: None     # (the LLM's attention output buffer keeps what it held)
elif what in ("head_fwd", "head_bwd"):
    _attn0 = E.Head._attn
    _filled = set()

    def _attn(self, i, fwd, dout=None):
        if fwd != (what == "head_fwd"):
            return _attn0(self, i, fwd, dout)
        for name in (("AOx",) if fwd else ("dQKVx", "dKV_adp", "dKV_task")):      # the skipped kernel's outputs: random, once
            t = getattr(self, name)[i]
            if (name, i) not in _filled:
                _filled.add((name, i))
                t.copy_((torch.randn(t.shape, device=t.device) * 0.05).to(t.dtype))
    E.Head._attn = _attn
import bench  # noqa: E402

bench.main()
