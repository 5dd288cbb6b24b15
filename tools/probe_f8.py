#!/usr/bin/env python3
"""v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit scales): lane l holds row l % 16 and k = 32 * (l / 16) + byte.
Random matrices against the fp32 product of the decoded values."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vla_adapter_amd import native as N, ops  # noqa: E402

lib = N.load()
lib.vla_probe_mfma_f8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.vla_probe_mfma_f8.restype = C.c_int


def run(a, b):
    out = torch.zeros(64 * 4, device="cuda")
    ad, bd = a.cuda(), b.cuda()
    N.check(lib.vla_probe_mfma_f8(ops._st(), ad.data_ptr(), bd.data_ptr(), out.data_ptr()), "probe")
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4)
    c = torch.zeros(16, 16)
    for lane in range(64):
        for r in range(4):
            c[(lane >> 4) * 4 + r, lane & 15] = o[lane, r]
    return c


def lay(M):
    u = M.view(torch.uint8)
    out = torch.zeros(64, 32, dtype=torch.uint8)
    for lane in range(64):
        out[lane] = u[lane % 16, 32 * (lane // 16):32 * (lane // 16) + 32]
    return out


def main():
    g = torch.Generator().manual_seed(0)
    for name, fa, fb in [("positive", lambda x: x.abs(), lambda x: x.abs()), ("A signed", lambda x: x, lambda x: x.abs()),
                         ("both signed", lambda x: x, lambda x: x), ("small", lambda x: x * 0.01, lambda x: x * 0.01)]:
        A = fa(torch.randn(16, 128, generator=g)).to(torch.float8_e4m3fn)
        B = fb(torch.randn(16, 128, generator=g)).to(torch.float8_e4m3fn)
        ref = A.float() @ B.float().t()
        c = run(lay(A), lay(B))
        print(f"{name:12s} max|C - A.B^T| {(c - ref).abs().max().item():.3e}  max|C - (A.B^T)^T| {(c - ref.t()).abs().max().item():.3e}  |ref| {ref.abs().max().item():.2f}")


if __name__ == "__main__":
    main()
