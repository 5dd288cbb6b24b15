"""Tensor-level wrappers over the C ABI: pointer/stride plumbing only (PyTorch owns memory and the stream).

Every function launches on ``torch.cuda.current_stream()`` and raises on failure.  Shapes are validated here
(host side) before any kernel is launched.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Optional, Tuple

import torch

from . import native as N
from .native import ACT_GELU, ACT_GELU_TANH, ACT_NONE, ACT_RELU, ACT_SWIGLU, ACT_SWIGLU_BWD  # noqa: F401

BF16 = torch.bfloat16


def _lib():
    return N.load()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == BF16, f"expected CUDA bf16 tensor, got {t.device} {t.dtype}"


def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, bias=None, residual=None, res_mod: int = 0, act: int = ACT_NONE,
            out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, alpha: float = 1.0,
            want_pre: bool = True, a_group=None, c_group=None, r_group=None, rope=None, c_live=None,
            split_k: Optional[int] = None, bias_post_round: bool = False, fp8=None, ext=None, query_256: bool = False) -> torch.Tensor:
    """C = epilogue(A @ B^T).  a: [M,K] or [batch,M,K] (row stride = a.stride(-2)); b: [N,K] or [batch,N,K].
    SwiGLU: returns (pre [.., N] or None, h [.., N/2]).  split_k: None = automatic, 0/1 = off, k = forced.
    fp8=(a_scale [M] f32, b_scale [N] f32): a and b are uint8 tensors of OCP e4m3 codes (quant_fp8_rows).
    ext=(a2 [M, K2], b2 [N, K2]) bf16: K extension, C = epilogue(A @ B^T + A2 @ B2^T) in one fp32 accumulator (LoRA branch); with fp8
    the base pair is e4m3 and dequantised before the extension adds to it.  query_256=True: no launch, returns whether this call
    would run on the 256-row kernel."""
    if fp8 is not None:
        assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and a.dim() == 2 and split_k in (None, 0, 1)
        _chk_bf16(bias, residual, out, out2)
        split_k = 0
    else:
        _chk_bf16(a, b, bias, residual, out, out2)
    assert a.stride(-1) == 1 and b.stride(-1) == 1
    batched = a.dim() == 3
    if batched:
        nb, M, K = a.shape
        sA = a.stride(0)
        sB = b.stride(0) if b.dim() == 3 else 0
        Nn = b.shape[-2]
    else:
        nb, (M, K), sA, sB, Nn = 1, a.shape, 0, 0, b.shape[0]
    assert b.shape[-1] == K, f"K mismatch {a.shape} x {b.shape}"
    d = N.GemmDesc()
    d.A, d.B = a.data_ptr(), b.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.batch = M, Nn, K, a.stride(-2), b.stride(-2), nb
    d.sA, d.sB, d.act, d.alpha, d.res_mod = sA, sB, act, alpha, res_mod
    shape = (nb, M, Nn) if batched else (M, Nn)
    if act == ACT_SWIGLU:
        if out is None and want_pre:
            out = torch.empty(shape, device=a.device, dtype=BF16)
        hshape = shape[:-1] + (Nn // 2,)
        if out2 is None:
            out2 = torch.empty(hshape, device=a.device, dtype=BF16)
        assert out2.shape == hshape and out2.stride(-1) == 1
        d.C2, d.ldc2, d.sC2 = out2.data_ptr(), out2.stride(-2), (out2.stride(0) if batched else 0)
    elif out is None:
        out = torch.empty(shape, device=a.device, dtype=BF16)
    if out is not None:
        assert tuple(out.shape) == tuple(shape) and out.stride(-1) == 1, f"out shape {out.shape} != {shape}"
        d.C, d.ldc, d.sC = out.data_ptr(), out.stride(-2), (out.stride(0) if batched else 0)
    if bias is not None:
        assert bias.shape[-1] == Nn and bias.stride(-1) == 1
        d.bias = bias.data_ptr()
        d.sBias = bias.stride(0) if bias.dim() == 2 else 0
    if residual is not None:
        assert residual.stride(-1) == 1 and residual.shape[-1] == Nn
        assert res_mod > 0 or r_group is not None or tuple(residual.shape) == tuple(shape)
        d.R, d.ldr = residual.data_ptr(), residual.stride(-2)
        d.sR = residual.stride(0) if (batched and residual.dim() == 3) else 0
    if a_group is not None:      # (rows per group, stride between groups): a = first row-group view [g, K]
        d.a_group, d.a_group_stride = a_group
    if c_group is not None:
        d.c_group, d.c_group_stride = c_group
    if r_group is not None:
        d.r_group, d.r_group_stride = r_group
    if c_live is not None:       # (period, first live row): rows m of C with m % period < first are not stored
        d.c_live_mod, d.c_live_from = c_live
    if rope is not None:         # (mode, cos, sin, T, dh, ncols): fused rotary embedding on output columns [0, ncols)
        mode, cos_t, sin_t, T, dh, ncols = rope
        assert cos_t.dtype == torch.float32 and cos_t.is_contiguous() and sin_t.is_contiguous() and cos_t.shape[0] >= T
        assert cos_t.shape[1] == (dh // 2 if mode == 1 else dh)
        d.rope_mode, d.rope_T, d.rope_dh, d.rope_cols = mode, T, dh, ncols
        d.rope_cos, d.rope_sin = cos_t.data_ptr(), sin_t.data_ptr()
    # split-K for few-tile long-K problems (batch-1 inference, and the live-row backward's M = B*64 GEMM over K = 2I): the
    # K loop of a tile is serial, so a problem with fewer tiles than the chip has workgroup slots (2 x 256) runs at the
    # latency of ONE long loop on part of the CUs; K slices meet in a per-stream fp32 workspace and are summed by
    # splitk_finalize_kernel
    if fp8 is not None:
        sa, sb = fp8
        assert sa.dtype == torch.float32 and sb.dtype == torch.float32 and sa.numel() == M and sb.numel() == Nn and sa.is_contiguous() and sb.is_contiguous()
        d.fp8, d.a_scale, d.b_scale = 1, sa.data_ptr(), sb.data_ptr()
    if ext is not None:
        a2, b2 = ext
        _chk_bf16(a2, b2)
        assert not batched and a2.dim() == 2 and b2.dim() == 2 and a2.shape[0] == M and b2.shape[0] == Nn and a2.shape[1] == b2.shape[1]
        assert a2.stride(1) == 1 and b2.stride(1) == 1 and a2.shape[1] % 64 == 0 and split_k in (None, 0, 1)
        d.A2, d.B2, d.K2, d.lda2, d.ldb2 = a2.data_ptr(), b2.data_ptr(), a2.shape[1], a2.stride(0), b2.stride(0)
        split_k = 0
    if bias_post_round:          # C = bf16(bf16(A.B^T) + bias): torch CPU Linear on a strided bf16 input (vla_native.h)
        assert bias is not None
        d.bias_post_round = 1
    plain = (not bias_post_round and not batched and act in (ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_TANH) and out is not None and a_group is None
             and c_group is None and r_group is None and rope is None and c_live is None and res_mod == 0 and Nn % 4 == 0)
    if split_k is None:
        split_k = 0
        tiles = ((M + 127) // 128) * ((Nn + 127) // 128)
        t64 = ((M + 63) // 64) * ((Nn + 127) // 128)
        if (plain and K >= 2048 and t64 * 2 <= 256 and not os.environ.get("VLA_NO_SPLITK") and not os.environ.get("VLA_NO_SPLITK_UNEVEN")
                and _lib().vla_gemm_latency_hint(-1) > 0):
            # the batch-1 pass (latency hint): as many K slices as give every CU one 64 x 128 workgroup, >= 4 K-tiles each (the slices need
            # not divide K: the last one is shorter) - ViT fc2 256 x 1152 x 4352: 7 slices of 10 K-tiles on 252 workgroups instead of 4 on 144
            split_k = max(2, min(256 // t64, (K // 64) // 4, 16))
        elif plain and K >= 2048 and tiles <= 256 and not os.environ.get("VLA_NO_SPLITK"):
            cap = 512                                                   # workgroup slots a split may fill (two per CU)
            for sk in (8, 4, 2):
                if tiles * sk <= cap and K % (64 * sk) == 0 and K // sk >= 512:
                    split_k = sk
                    break
    if split_k > 1:
        assert plain, "split-K needs a plain epilogue"
        d.split_k, d.ws = split_k, _splitk_ws(split_k * M * Nn, a.device).data_ptr()
    if query_256:
        return bool(_lib().vla_gemm_uses_256(C.byref(d)))
    N.check(_lib().vla_gemm_bf16_nt(_st(), C.byref(d)), "gemm_bf16_nt")
    if act == ACT_SWIGLU:
        return out, out2
    return out


class latency_hint:
    """``with ops.latency_hint():`` - the products launched (or captured into a hipGraph) inside run on an otherwise idle chip and are
    latency-bound (batch-1 predict_action): vla_gemm_latency_hint(1) for the block.  Kernel selection only, bit-identical results."""

    def __enter__(self):
        self.prev = _lib().vla_gemm_latency_hint(1)
        return self

    def __exit__(self, *exc):
        _lib().vla_gemm_latency_hint(self.prev)
        return False


_SPLITK_WS = {}


def _splitk_ws(numel: int, device) -> torch.Tensor:
    """fp32 workspace of the CURRENT stream (streams overlap, so each has its own)."""
    key = (torch.cuda.current_stream().cuda_stream, str(device))
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < numel:
        ws = _SPLITK_WS[key] = torch.empty(max(numel, 8 << 20), device=device, dtype=torch.float32)
    return ws


def gemm_swiglu_bwd(d: torch.Tensor, w_downT: torch.Tensor, gu: torch.Tensor, out: Optional[torch.Tensor] = None,
                    gu_group=None, ext=None, fp8=None) -> torch.Tensor:
    """dGU[M, 2I] = swiglu'(GU) * (d[M, D] @ w_downT[I, D]^T): the dH GEMM with the SwiGLU backward in its epilogue.
    gu_group=(rows per group, element stride between groups): ``gu`` is then the first row-group window of a larger
    tensor (row m of the product reads gu row (m // g) * stride + (m % g) * ld).
    fp8=(d_scale [M], w_scale [I]) (with ext only): d and w_downT are uint8 e4m3 codes (quant_fp8_rows)."""
    if fp8 is not None:
        assert ext is not None and d.dtype == torch.uint8 and w_downT.dtype == torch.uint8
        _chk_bf16(gu, out)
    else:
        _chk_bf16(d, w_downT, gu, out)
    M, K = d.shape
    I = w_downT.shape[0]
    assert gu.shape[-1] == 2 * I and gu.stride(-1) == 1 and w_downT.shape[1] == K
    assert gu_group is not None or gu.shape[0] == M
    if out is None:
        out = torch.empty(M, 2 * I, device=d.device, dtype=BF16)
    assert out.shape == (M, 2 * I)
    desc = N.GemmDesc()
    desc.A, desc.B, desc.C, desc.R = d.data_ptr(), w_downT.data_ptr(), out.data_ptr(), gu.data_ptr()
    desc.M, desc.N, desc.K, desc.lda, desc.ldb, desc.ldc, desc.ldr, desc.batch = M, I, K, d.stride(0), w_downT.stride(0), out.stride(0), gu.stride(0), 1
    desc.act, desc.alpha = ACT_SWIGLU_BWD, 1.0
    if gu_group is not None:
        desc.r_group, desc.r_group_stride = gu_group
    if ext is not None:          # K extension: dH = d . W_down + dt . A_down (LoRA on down_proj)
        a2, b2 = ext
        _chk_bf16(a2, b2)
        assert a2.shape == (M, b2.shape[1]) and b2.shape[0] == I and a2.stride(1) == 1 and b2.stride(1) == 1 and a2.shape[1] % 64 == 0
        desc.A2, desc.B2, desc.K2, desc.lda2, desc.ldb2 = a2.data_ptr(), b2.data_ptr(), a2.shape[1], a2.stride(0), b2.stride(0)
    if fp8 is not None:
        sa, sb = fp8
        assert sa.dtype == torch.float32 and sb.dtype == torch.float32 and sa.numel() == M and sb.numel() == I and sa.is_contiguous() and sb.is_contiguous()
        desc.fp8, desc.a_scale, desc.b_scale = 1, sa.data_ptr(), sb.data_ptr()
    N.check(_lib().vla_gemm_bf16_nt(_st(), C.byref(desc)), "gemm_bf16_nt(swiglu_bwd)")
    return out


_TN_WS = {}


def gemm_tn(a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None, alpha: float = 1.0, accumulate: bool = False,
            a_group=None, b_group=None, a_cols=None, rows: Optional[int] = None, split: Optional[int] = None) -> torch.Tensor:
    """C[.., N1, N2] = alpha * a[.., M, N1]^T @ b[.., M, N2]  (contraction over the ROWS: dW = dY^T X, no operand transposes).
    a / b: [M, N] or [batch, M, N] views with last stride 1.  accumulate: C = bf16(bf16(product) + C).
    rows: contraction length when it differs from a.shape[-2] (row groups).  a_group / b_group = (rows per group, element stride
    between groups): a / b is then the first group's [g, N] view of a larger tensor.  a_cols = (n1, group, stride, offset): the
    product's N1 axis is the n1 columns {offset + (c // group) * stride + c % group} of a (gate / up halves of an interleaved dY).
    split: None = automatic (few-tile long-M products), 0 / 1 = off."""
    _chk_bf16(a, b, out)
    assert a.stride(-1) == 1 and b.stride(-1) == 1 and a.dim() == b.dim()
    batched = a.dim() == 3
    nb = a.shape[0] if batched else 1
    M = rows if rows is not None else a.shape[-2]
    assert rows is not None or b.shape[-2] == M
    N1, N2 = a.shape[-1], b.shape[-1]
    d = N.GemmTnDesc()
    a_ptr = a.data_ptr()
    if a_cols is not None:
        N1, cg, cgs, off = a_cols
        assert cg % 8 == 0 and cgs % 8 == 0 and off % 8 == 0
        d.a_col_group, d.a_col_group_stride = cg, cgs
        a_ptr += off * 2
    shape = (nb, N1, N2) if batched else (N1, N2)
    if out is None:
        assert not accumulate
        out = torch.empty(shape, device=a.device, dtype=BF16)
    assert tuple(out.shape) == shape and out.stride(-1) == 1, f"out shape {tuple(out.shape)} != {shape}"
    d.A, d.B, d.C = a_ptr, b.data_ptr(), out.data_ptr()
    d.M, d.N1, d.N2, d.lda, d.ldb, d.ldc, d.batch = M, N1, N2, a.stride(-2), b.stride(-2), out.stride(-2), nb
    d.sA, d.sB, d.sC = (a.stride(0), b.stride(0), out.stride(0)) if batched else (0, 0, 0)
    d.alpha = alpha
    if accumulate:
        d.R, d.ldr, d.sR = out.data_ptr(), out.stride(-2), d.sC
    if a_group is not None:
        d.a_group, d.a_group_stride = a_group
    if b_group is not None:
        d.b_group, d.b_group_stride = b_group
    if split is None:
        split = 0
        tiles = ((N1 + 127) // 128) * ((N2 + 127) // 128) * nb
        if tiles <= 128 and M >= 1024:
            split = max(1, min(16, 512 // tiles, M // 256))
    if split > 1:
        per = ((M + split - 1) // split + 63) // 64 * 64
        split = (M + per - 1) // per                  # no empty slice
    if split > 1:
        key = (torch.cuda.current_stream().cuda_stream, str(a.device))
        need = nb * split * N1 * N2
        ws = _TN_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _TN_WS[key] = torch.empty(max(need, 4 << 20), device=a.device, dtype=torch.float32)
        d.split, d.ws = split, ws.data_ptr()
    N.check(_lib().vla_gemm_bf16_tn(_st(), C.byref(d)), "gemm_bf16_tn")
    return out


TN_GROUP_MAX = 48


def tn_problem(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, alpha: float = 1.0, a_cols=None) -> "N.GemmTnDesc":
    """Descriptor of one plain product out[N1, N2] = alpha a[M, N1]^T b[M, N2] for gemm_tn_grouped (2-D views, last stride 1)."""
    _chk_bf16(a, b, out)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1 and a.shape[0] == b.shape[0]
    d = N.GemmTnDesc()
    N1, a_ptr = a.shape[1], a.data_ptr()
    if a_cols is not None:
        N1, cg, cgs, off = a_cols
        d.a_col_group, d.a_col_group_stride = cg, cgs
        a_ptr += off * 2
    assert tuple(out.shape) == (N1, b.shape[1]), f"out shape {tuple(out.shape)} != {(N1, b.shape[1])}"
    d.A, d.B, d.C = a_ptr, b.data_ptr(), out.data_ptr()
    d.M, d.N1, d.N2, d.lda, d.ldb, d.ldc, d.batch, d.alpha = a.shape[0], N1, b.shape[1], a.stride(0), b.stride(0), out.stride(0), 1, alpha
    d._keep = (a, b, out)                      # the descriptor holds raw pointers: keep the tensors alive with it
    return d


def gemm_tn_grouped(problems):
    """ONE launch (per 48 problems) over the concatenated tile lists of independent TN products (tn_problem descriptors)."""
    for i in range(0, len(problems), TN_GROUP_MAX):
        chunk = problems[i:i + TN_GROUP_MAX]
        arr = (N.GemmTnDesc * len(chunk))(*chunk)
        N.check(_lib().vla_gemm_bf16_tn_grouped(_st(), arr, len(chunk)), "gemm_bf16_tn_grouped")


def copy_rows3d(src: torch.Tensor, dst: torch.Tensor, groups: int, rows: int, cols: int, s_sg: int, s_sr: int, d_sg: int, d_sr: int):
    """dst[g][r][:cols] = src[g][r][:cols] over raw element strides (src / dst: tensors whose data_ptr is element [0][0][0])."""
    _chk_bf16(src, dst)
    N.check(_lib().vla_copy_rows3d(_st(), _p(src), _p(dst), groups, rows, cols, s_sg, s_sr, d_sg, d_sr), "copy_rows3d")
    return dst


def layerscale_fwd(a: torch.Tensor, ls: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = bf16(x + bf16(a * ls))  (modeling_prismatic.py:58-66 + the residual add); out may be x."""
    _chk_bf16(a, ls, x, out)
    assert a.is_contiguous() and x.is_contiguous() and a.shape == x.shape and ls.numel() == a.shape[-1]
    out = torch.empty_like(x) if out is None else out
    N.check(_lib().vla_layerscale_fwd(_st(), _p(a), _p(ls), _p(x), _p(out), a.numel() // a.shape[-1], a.shape[-1]), "layerscale_fwd")
    return out


def layerscale_bwd(dy: torch.Tensor, a: Optional[torch.Tensor], ls: torch.Tensor, dls_f32: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """da = bf16(dy * ls); dls_f32 (optional, +=) += column sums of dy * a."""
    _chk_bf16(dy, a, ls, out)
    assert dy.is_contiguous() and (a is None or (a.is_contiguous() and a.shape == dy.shape))
    out = torch.empty_like(dy) if out is None else out
    N.check(_lib().vla_layerscale_bwd(_st(), _p(dy), _p(a), _p(ls), _p(out), _p(dls_f32), dy.numel() // dy.shape[-1], dy.shape[-1]), "layerscale_bwd")
    return out


def transpose(x: torch.Tensor, ld_out: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [.., R, Cc] -> [.., Cc, ld_out] (columns R..ld_out-1 zero).  Returns the padded buffer."""
    _chk_bf16(x, out)
    assert x.stride(-1) == 1
    batched = x.dim() == 3
    nb = x.shape[0] if batched else 1
    R, Cc = x.shape[-2:]
    ld = ld_out or R
    if out is None:
        shape = (nb, Cc, ld) if batched else (Cc, ld)
        out = torch.empty(shape, device=x.device, dtype=BF16)
        if ld != R:
            zero_(out)                 # (native fill: the padding columns; torch.zeros would put an ATen kernel on the step)
    N.check(_lib().vla_transpose_bf16(_st(), _p(x), _p(out), R, Cc, x.stride(-2), out.stride(-2), nb,
                                      x.stride(0) if batched else 0, out.stride(0) if batched else 0), "transpose")
    return out


def layernorm_fwd(x, w, b, eps: float, want_stats: bool = False):
    _chk_bf16(x, w, b)
    cols = x.shape[-1]
    x2 = x.reshape(-1, cols)
    assert x2.stride(-1) == 1
    y = torch.empty_like(x2)
    stats = torch.empty(x2.shape[0], 2, device=x.device, dtype=torch.float32) if want_stats else None
    N.check(_lib().vla_layernorm_fwd(_st(), _p(x2), _p(w), _p(b), _p(y), _p(stats), x2.shape[0], cols, x2.stride(0),
                                     y.stride(0), eps), "layernorm_fwd")
    return (y.view(x.shape), stats) if want_stats else y.view(x.shape)


def layernorm_bwd(dy, x, w, stats, dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None,
                  want_dx: bool = True):
    _chk_bf16(dy, x, w)
    cols = x.shape[-1]
    x2, dy2 = x.reshape(-1, cols), dy.reshape(-1, cols)
    dx = torch.empty_like(x2) if want_dx else None
    N.check(_lib().vla_layernorm_bwd(_st(), _p(dy2), _p(x2), _p(w), _p(stats), _p(dx), _p(dw), _p(db), x2.shape[0], cols,
                                     x2.stride(0), dy2.stride(0), dx.stride(0) if dx is not None else cols), "layernorm_bwd")
    return dx.view(x.shape) if dx is not None else None


def rmsnorm_fwd(x, w, eps: float, want_rstd: bool = False, out=None):
    _chk_bf16(x, w)
    cols = x.shape[-1]
    assert x.is_contiguous()
    rows = x.numel() // cols
    y = torch.empty_like(x) if out is None else out
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if want_rstd else None
    N.check(_lib().vla_rmsnorm_fwd(_st(), _p(x), _p(w), _p(y), _p(rstd), rows, cols, eps), "rmsnorm_fwd")
    return (y, rstd) if want_rstd else y


def rmsnorm_fwd_q8(x, w, eps: float, q8: torch.Tensor, qscale: torch.Tensor, rstd: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None):
    """RMSNorm with the row-wise e4m3 quantisation of its bf16 output fused behind it: fills q8 (uint8 [rows, cols]) and
    qscale (f32 [rows]); rstd / y (the bf16 output) are optional."""
    _chk_bf16(x, w, y)
    cols = x.shape[-1]
    assert x.is_contiguous() and q8.dtype == torch.uint8 and q8.stride(-1) == 1
    rows = x.numel() // cols
    N.check(_lib().vla_rmsnorm_fwd_q8(_st(), _p(x), _p(w), _p(y), _p(rstd), _p(q8), _p(qscale), rows, cols, q8.stride(-2), eps), "rmsnorm_fwd_q8")
    return q8, qscale


def layernorm_fwd_q8(x2, w, b, eps: float, q8: torch.Tensor, qscale: torch.Tensor, y: Optional[torch.Tensor] = None, stats=None):
    """LayerNorm (rows of x2 [rows, cols], row stride x2.stride(0)) + fused row-wise e4m3 quantisation of its bf16 output."""
    _chk_bf16(x2, w, b, y)
    rows, cols = x2.shape
    assert x2.stride(-1) == 1 and q8.dtype == torch.uint8 and q8.stride(-1) == 1
    N.check(_lib().vla_layernorm_fwd_q8(_st(), _p(x2), _p(w), _p(b), _p(y), _p(stats), _p(q8), _p(qscale), rows, cols, x2.stride(0),
                                        y.stride(0) if y is not None else 0, q8.stride(0), eps), "layernorm_fwd_q8")
    return q8, qscale


def rmsnorm_bwd(dy, x, w, rstd, dres=None, out=None, x_rows=None):
    """dy/dres/out compact [rows, cols].  x_rows=(group, group_rows, row0): x / rstd are the forward's full tensors and
    compact row r maps to row (r // group) * group_rows + row0 + r % group (live-row window of every sequence)."""
    _chk_bf16(dy, x, w, dres)
    cols = x.shape[-1]
    assert x.is_contiguous() and dy.is_contiguous()
    rows = dy.numel() // cols
    g, gr, r0 = x_rows if x_rows is not None else (0, 0, 0)
    if x_rows is None:
        assert x.numel() == dy.numel()
    else:
        assert rows % g == 0 and (rows // g) * gr * cols <= x.numel() and rstd.numel() * cols >= (rows // g) * gr * cols
    dx = torch.empty_like(dy) if out is None else out
    if dres is not None:
        assert dres.is_contiguous() and dres.numel() >= dy.numel()
    N.check(_lib().vla_rmsnorm_bwd(_st(), _p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), rows, cols, g, gr, r0), "rmsnorm_bwd")
    return dx


def _attn_desc(q, k, v, o, lse, kmask, causal, scale, Hq, Hkv, dh):
    """q [B,Sq,>=Hq*dh], k/v [B,Sk,>=Hkv*dh] views (last-dim stride 1), o [B,Sq,Hq*dh]."""
    d = N.AttnDesc()
    B, Sq, Sk = q.shape[0], q.shape[1], k.shape[1]
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    d.lse = lse.data_ptr() if lse is not None else None
    d.kmask = kmask.data_ptr() if kmask is not None else None
    d.q_sb, d.k_sb, d.v_sb, d.o_sb = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    d.q_ss, d.k_ss, d.v_ss, d.o_ss = q.stride(1), k.stride(1), v.stride(1), o.stride(1)
    d.B, d.Sq, d.Sk, d.Hq, d.Hkv, d.dh, d.causal, d.scale = B, Sq, Sk, Hq, Hkv, dh, int(causal), scale
    return d


def attn_fwd(q, k, v, Hq: int, Hkv: int, dh: int, causal: bool, kmask=None, scale: Optional[float] = None,
             want_lse: bool = False):
    _chk_bf16(q, k, v)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
    B, Sq = q.shape[:2]
    if kmask is not None:
        assert kmask.dtype == torch.uint8 and kmask.is_contiguous() and tuple(kmask.shape) == (B, k.shape[1])
    o = torch.empty(B, Sq, Hq * dh, device=q.device, dtype=BF16)
    lse = torch.empty(B, Hq, Sq, device=q.device, dtype=torch.float32) if want_lse else None
    d = _attn_desc(q, k, v, o, lse, kmask, causal, scale if scale is not None else dh ** -0.5, Hq, Hkv, dh)
    N.check(_lib().vla_attn_fwd(_st(), C.byref(d)), "attn_fwd")
    return (o, lse) if want_lse else o


def attn_bwd(dout, q, k, v, o, lse, Hq: int, Hkv: int, dh: int, causal: bool, kmask=None,
             scale: Optional[float] = None, dq=None, dk=None, dv=None, rope=None, row0: int = 0):
    """row0 > 0 (causal only, multiple of 32): live-row backward.  q/o/dout/dq are the [B, Sk - row0, ...] windows
    (views) of the rows >= row0, k/v the full [B, Sk, ...] tensors, lse the forward's full [B, Hq, Sk]; dk/dv are
    produced for the keys >= row0 only ([B, Sk - row0, ...]).  Exactly the gradients the rows >= row0 receive."""
    _chk_bf16(dout, q, k, v, o)
    B, Sq, Sk = q.shape[0], q.shape[1], k.shape[1]
    assert Sq + row0 == Sk or row0 == 0
    dq = torch.empty(B, Sq, Hq * dh, device=q.device, dtype=BF16) if dq is None else dq
    dk = torch.empty(B, Sk - row0, Hkv * dh, device=q.device, dtype=BF16) if dk is None else dk
    dv = torch.empty(B, Sk - row0, Hkv * dh, device=q.device, dtype=BF16) if dv is None else dv
    delta = torch.empty(B, Hq, Sq, device=q.device, dtype=torch.float32)
    d = _attn_desc(q, k, v, o, lse, kmask, causal, scale if scale is not None else dh ** -0.5, Hq, Hkv, dh)
    if row0:
        assert causal and row0 % 32 == 0 and lse.shape == (B, Hq, Sk) and lse.is_contiguous()
        d.q_off, d.dkv_k0, d.lse_hs = row0, row0, Sk
        d.lse = lse.data_ptr() + 4 * row0
    d.dout, d.dq, d.dk, d.dv, d.delta = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
    d.do_sb, d.dq_sb, d.dk_sb, d.dv_sb = dout.stride(0), dq.stride(0), dk.stride(0), dv.stride(0)
    d.do_ss, d.dq_ss, d.dk_ss, d.dv_ss = dout.stride(1), dq.stride(1), dk.stride(1), dv.stride(1)
    if rope is not None:          # (cos, sin) f32 [S, dh/2]: dq/dk come back through the inverse rotate_half RoPE
        assert rope[0].shape == (Sk, dh // 2) and rope[0].dtype == torch.float32
        d.rope_cos, d.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
    N.check(_lib().vla_attn_bwd(_st(), C.byref(d)), "attn_bwd")
    return dq, dk, dv


def rope_half_tables(S: int, dh: int, theta: float, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """f32 [S, dh/2] tables of bf16-rounded cos/sin (HF casts cos/sin to the activation dtype)."""
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    f = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
    return (f.cos().to(BF16).float().to(device).contiguous(), f.sin().to(BF16).float().to(device).contiguous())


def rope_half_(x2d: torch.Tensor, cos_t, sin_t, S: int, nheads: int, dh: int, sign: int = 1):
    """In place on x2d [rows, >= nheads*dh] (a column window of the fused qkv buffer)."""
    _chk_bf16(x2d)
    assert x2d.stride(-1) == 1 and cos_t.shape == (S, dh // 2)
    N.check(_lib().vla_rope_half(_st(), _p(x2d), _p(cos_t), _p(sin_t), x2d.shape[0], S, nheads, dh, x2d.stride(0), sign),
            "rope_half")
    return x2d


def rope_inter_tables(T: int, dh: int, device, base: float = 10000.0):
    """action_heads.py:150-164: cos/sin of cat([f, f]) -> f32 [T, dh] tables holding the bf16 values the reference's bf16
    head computes.  finetune.py:280-281 casts the head with .to(torch.bfloat16), which casts the registered ``inv_freq``
    buffer too: positions are a bf16 arange (above 256 they collapse onto representable even numbers), the angle
    t * inv_freq is rounded to bf16 BEFORE cos / sin.  Host-side table construction (torch CPU as the array library),
    pinned by the reference-run fixtures tests/golden/head_bf16_*.npz."""
    inv = (1.0 / (base ** (torch.arange(0, dh, 2).float() / dh))).to(BF16)
    f = torch.einsum("i,j->ij", torch.arange(T, dtype=BF16), inv)
    e = torch.cat([f, f], dim=-1)
    return (e.cos().float().to(device).contiguous(), e.sin().float().to(device).contiguous())


def rope_inter_(x2d: torch.Tensor, cos_t, sin_t, T: int, nheads: int, dh: int, mode: int = 0):
    _chk_bf16(x2d)
    assert x2d.stride(-1) == 1 and cos_t.shape[0] >= T and cos_t.shape[1] == dh
    N.check(_lib().vla_rope_interleaved(_st(), _p(x2d), _p(cos_t), _p(sin_t), x2d.shape[0], T, nheads, dh, x2d.stride(0),
                                        mode), "rope_interleaved")
    return x2d


def im2col_patch(pixels: torch.Tensor, c0: int, P: int, ldo: int) -> torch.Tensor:
    assert pixels.is_cuda and pixels.is_contiguous() and pixels.dtype in (BF16, torch.float32)
    B, Ct, H, W = pixels.shape
    out = torch.empty(B * (H // P) * (W // P), ldo, device=pixels.device, dtype=BF16)
    N.check(_lib().vla_im2col_patch(_st(), _p(pixels), _p(out), B, Ct, c0, H, W, P, ldo,
                                    int(pixels.dtype == torch.float32)), "im2col_patch")
    return out


def action_mask(labels: torch.Tensor, shift: int):
    """-> qidx int32 [B, L-shift], pos int32 [B,64], count int32 [B] (device tensors; no host sync)."""
    assert labels.is_cuda and labels.dtype == torch.int64 and labels.is_contiguous()
    B, L = labels.shape
    qidx = torch.empty(B, L - shift, device=labels.device, dtype=torch.int32)
    pos = torch.empty(B, 64, device=labels.device, dtype=torch.int32)
    cnt = torch.empty(B, device=labels.device, dtype=torch.int32)
    N.check(_lib().vla_action_mask(_st(), _p(labels), _p(qidx), _p(pos), _p(cnt), B, L, shift), "action_mask")
    return qidx, pos, cnt


def embed_splice(ids, attn_mask_u8, qidx, table, action_queries, out, mm_mask, Np: int):
    B, L = ids.shape
    D = table.shape[1]
    assert ids.dtype == torch.int64 and ids.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (B, L + Np, D)
    assert tuple(action_queries.shape) == (64, D) and qidx.shape == (B, L)
    N.check(_lib().vla_embed_splice(_st(), _p(ids), _p(attn_mask_u8), _p(qidx), _p(table), _p(action_queries), _p(out),
                                    _p(mm_mask), B, L, Np, D, table.shape[0]), "embed_splice")


def action_query_grad(dx, pos, Np: int, row0: int = 0) -> torch.Tensor:
    """dx [B, S - row0, D]: the rows >= row0 of the gradient w.r.t. inputs_embeds."""
    B, S, D = dx.shape
    assert dx.is_contiguous()
    dq = torch.empty(64, D, device=dx.device, dtype=torch.float32)
    N.check(_lib().vla_action_query_grad(_st(), _p(dx), _p(pos), _p(dq), B, S, Np, D, row0), "action_query_grad")
    return dq


def gather_rows(src2d, idx, out2d):
    N.check(_lib().vla_gather_rows(_st(), _p(src2d), _p(idx), _p(out2d), idx.numel(), out2d.shape[-1], src2d.stride(0),
                                   out2d.stride(0)), "gather_rows")
    return out2d


def scatter_add_rows(src2d, idx, out2d):
    N.check(_lib().vla_scatter_add_rows(_st(), _p(src2d), _p(idx), _p(out2d), idx.numel(), src2d.shape[-1], src2d.stride(0),
                                        out2d.stride(0)), "scatter_add_rows")
    return out2d


def add_(a, b):
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    N.check(_lib().vla_add_bf16(_st(), _p(a), _p(b), _p(a), a.numel()), "add_bf16")
    return a


def gelu_fwd(x):
    y = torch.empty_like(x)
    N.check(_lib().vla_gelu_fwd(_st(), _p(x), _p(y), x.numel()), "gelu_fwd")
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    N.check(_lib().vla_gelu_bwd(_st(), _p(dy), _p(x), _p(dx), x.numel()), "gelu_bwd")
    return dx


def relu_bwd(dy, y):
    dx = torch.empty_like(y)
    N.check(_lib().vla_relu_bwd(_st(), _p(dy), _p(y), _p(dx), y.numel()), "relu_bwd")
    return dx


def swiglu_bwd(dh, gu, out=None):
    M, I = dh.shape
    assert gu.shape == (M, 2 * I) and dh.is_contiguous() and gu.is_contiguous()
    dgu = torch.empty_like(gu) if out is None else out
    N.check(_lib().vla_swiglu_bwd(_st(), _p(dh), _p(gu), _p(dgu), M, I), "swiglu_bwd")
    return dgu


def swiglu_fwd(gu, out=None):
    M, I2 = gu.shape
    assert gu.is_contiguous() and I2 % 32 == 0
    h = torch.empty(M, I2 // 2, device=gu.device, dtype=BF16) if out is None else out
    N.check(_lib().vla_swiglu_fwd(_st(), _p(gu), _p(h), M, I2 // 2), "swiglu_fwd")
    return h


def colsum_(x, out_f32):
    """x [rows, cols] or [batch, rows, cols] (bf16) -> out_f32 [cols] / [batch, cols] += column sums."""
    if x.dim() == 3:
        assert out_f32.shape == (x.shape[0], x.shape[2]) and out_f32.stride(-1) == 1 and x.stride(-1) == 1
        N.check(_lib().vla_colsum_bf16(_st(), _p(x), _p(out_f32), x.shape[1], x.shape[2], x.stride(1), x.shape[0], x.stride(0),
                                       out_f32.stride(0)), "colsum")
    else:
        N.check(_lib().vla_colsum_bf16(_st(), _p(x), _p(out_f32), x.shape[0], x.shape[1], x.stride(0), 1, 0, 0), "colsum")
    return out_f32


def cast_f32_bf16(x, out=None):
    y = torch.empty(x.shape, device=x.device, dtype=BF16) if out is None else out
    N.check(_lib().vla_cast_f32_bf16(_st(), _p(x), _p(y), x.numel()), "cast")
    return y


def head_attn_desc(q, ks, vs, ka, va, kt, vt, gate, probs, out, H: int):
    d = N.HeadAttnDesc()
    B, T = q.shape[:2]
    d.q, d.k_self, d.v_self, d.k_adp, d.v_adp, d.k_task, d.v_task = (t.data_ptr() for t in (q, ks, vs, ka, va, kt, vt))
    d.gate, d.probs, d.out = gate.data_ptr(), probs.data_ptr(), out.data_ptr()
    d.B, d.T, d.Ka, d.Kt, d.H, d.dh = B, T, ka.shape[1], kt.shape[1], H, out.shape[-1] // H
    d.ld_q, d.ld_self, d.ld_adp, d.ld_task, d.ld_out = q.stride(1), ks.stride(1), ka.stride(1), kt.stride(1), out.stride(1)
    for t, n in ((vs, ks), (va, ka), (vt, kt)):
        assert t.stride(1) == n.stride(1) and t.stride(-1) == 1
    return d


def head_attn_fwd(q, ks, vs, ka, va, kt, vt, gate, H: int = 8, ref_softmax: bool = False):
    """q/ks/vs [B,T,D*], ka/va [B,Ka,D*], kt/vt [B,Kt,D*] (views, last stride 1) -> out [B,T,D], probs f32.
    ref_softmax: weights rounded to bf16 after normalisation, as ATen's bf16 softmax emits them (two passes over the keys)."""
    _chk_bf16(q, ks, vs, ka, va, kt, vt, gate)
    B, T = q.shape[:2]
    D = q.shape[-1]
    Nn = T + ka.shape[1] + kt.shape[1]
    out = torch.empty(B, T, D, device=q.device, dtype=BF16)
    probs = torch.empty(B, H, T, Nn, device=q.device, dtype=torch.float32)
    d = head_attn_desc(q, ks, vs, ka, va, kt, vt, gate, probs, out, H)
    d.ref_softmax = int(ref_softmax)
    N.check(_lib().vla_head_attn_fwd(_st(), C.byref(d)), "head_attn_fwd")
    return out, probs


def head_attn_bwd(dout, out, q, ks, vs, ka, va, kt, vt, gate, probs, dgate_f32, dq, dks, dvs, dka, dva, dkt, dvt, H: int = 8,
                  rope=None):
    """``out`` = the forward output, ``probs`` = the forward's workspace.  Gradient tensors are caller-provided views
    with the SAME row strides as their forward counterparts."""
    _chk_bf16(dout, out, dq, dks, dvs, dka, dva, dkt, dvt)
    d = head_attn_desc(q, ks, vs, ka, va, kt, vt, gate, probs, out, H)
    d.dout = dout.data_ptr()
    assert dout.stride(1) == out.stride(1) and dout.stride(-1) == 1
    for g, f in ((dq, q), (dks, ks), (dvs, vs), (dka, ka), (dva, va), (dkt, kt), (dvt, vt)):
        assert g.stride(1) == f.stride(1) and g.stride(-1) == 1, "grad views must mirror forward strides"
    d.dq, d.dk_self, d.dv_self, d.dk_adp, d.dv_adp, d.dk_task, d.dv_task = (t.data_ptr() for t in (dq, dks, dvs, dka, dva, dkt, dvt))
    d.dgate = dgate_f32.data_ptr()
    if rope is not None:          # (cos, sin) f32 [>= max(T,Ka,Kt), dh]: dq / dk come back through the RoPE transpose
        assert rope[0].shape[0] >= max(q.shape[1], ka.shape[1], kt.shape[1]) and rope[0].shape[1] == d.dh
        d.rope_cos, d.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
    # workspace of the tile-uniform MFMA backward (dQ / gate partials per 32-key tile): B*H*ceil(N/32)*(T*dh + 1) floats
    ntile = (d.T + d.Ka + d.Kt + 31) // 32
    ws = _splitk_ws(d.B * d.H * ntile * (d.T * d.dh + 1), q.device)
    d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    N.check(_lib().vla_head_attn_bwd(_st(), C.byref(d)), "head_attn_bwd")


def l1_loss(pred, target, want_grad: bool = True, gscale: float = 1.0):
    _chk_bf16(pred, target)
    B, Cc, Da = pred.shape
    loss3 = torch.empty(3, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    N.check(_lib().vla_l1_loss(_st(), _p(pred.contiguous()), _p(target.contiguous()), _p(loss3), _p(dpred), B, Cc, Da, gscale),
            "l1_loss")
    return loss3, dpred


def adamw_(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01, gscale: float = 1.0):
    assert p.dtype == BF16 and m.dtype == BF16 and v.dtype == BF16 and g.dtype in (BF16, torch.float32)
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    assert p.numel() == g.numel() == m.numel() == v.numel()
    N.check(_lib().vla_adamw_bf16(_st(), _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, step,
                                  int(g.dtype == torch.float32), gscale), "adamw")


# ---- host-glue replacements: no ATen kernel between the hand-written ones on the training step ---------------------------
_DT = {BF16: 0, torch.float32: 1}


def copy2d(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, ld_src: int, ld_dst: int, src_mod: int = 0, d_group=None):
    """dst[r, :cols] = cast(src[r % src_mod or r, :cols]) over raw row strides (elements); d_group=(rows per group, group stride)."""
    g, gs = d_group if d_group is not None else (0, 0)
    N.check(_lib().vla_copy2d(_st(), _p(src), _p(dst), rows, cols, ld_src, ld_dst, _DT[src.dtype], _DT[dst.dtype], src_mod, g, gs), "copy2d")
    return dst


def quant_fp8_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None):
    """x bf16 [rows, cols] -> (uint8 [rows, cols] of OCP e4m3 codes, f32 [rows] dequantisation scales = amax / 448)."""
    assert x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=torch.uint8)
    if scale is None:
        scale = torch.empty(rows, device=x.device, dtype=torch.float32)
    assert out.dtype == torch.uint8 and out.stride(1) == 1 and tuple(out.shape) == (rows, cols)
    N.check(_lib().vla_quant_fp8_rows(_st(), _p(x), _p(out), _p(scale), rows, cols, x.stride(0), out.stride(0)), "quant_fp8_rows")
    return out, scale


def dropout(x: torch.Tensor, out: torch.Tensor, p: float, seed: int, step: Optional[torch.Tensor] = None):
    """out = dropout(x, p) on 2-D bf16 views (last stride 1): vla_dropout_bf16.  step: device int32 [1] (the mask's second key) or None."""
    _chk_bf16(x, out)
    assert x.dim() == 2 and x.shape == out.shape and x.stride(1) == 1 and out.stride(1) == 1
    N.check(_lib().vla_dropout_bf16(_st(), _p(x), _p(out), x.shape[0], x.shape[1], x.stride(0), out.stride(0), float(p), int(seed) & (2 ** 64 - 1),
                                    _p(step) if step is not None else None), "dropout")
    return out


def dropout_bwd_add_(dx: torch.Tensor, u: torch.Tensor, p: float, seed: int, step: Optional[torch.Tensor] = None):
    """dx += dropout'(u) with the mask of dropout(., p, seed, step): vla_dropout_bwd_add_bf16."""
    _chk_bf16(dx, u)
    assert u.dim() == 2 and u.shape == dx.shape and u.stride(1) == 1 and dx.stride(1) == 1
    N.check(_lib().vla_dropout_bwd_add_bf16(_st(), _p(u), _p(dx), u.shape[0], u.shape[1], u.stride(0), dx.stride(0), float(p), int(seed) & (2 ** 64 - 1),
                                            _p(step) if step is not None else None), "dropout_bwd_add")
    return dx


def inc_i32_(t: torch.Tensor):
    assert t.dtype == torch.int32 and t.numel() >= 1
    N.check(_lib().vla_inc_i32(_st(), _p(t)), "inc_i32")
    return t


def zero_(t: torch.Tensor):
    assert t.is_contiguous()
    N.check(_lib().vla_fill_zero(_st(), _p(t), t.numel() * t.element_size()), "fill_zero")
    return t


def head_index_prep(pos1, pos0, cnt0, gather, scatter, guard, B: int, S: int, Np: int, row0: int):
    N.check(_lib().vla_head_index_prep(_st(), _p(pos1), _p(pos0), _p(cnt0), _p(gather), _p(scatter), _p(guard), B, S, Np, row0), "head_index_prep")


def add_scalar_f32_(x: torch.Tensor, s: torch.Tensor):
    N.check(_lib().vla_add_scalar_f32(_st(), _p(x), _p(s), x.numel()), "add_scalar_f32")
    return x


# ---- input stage ------------------------------------------------------------------------------------
def image_normalize_u8_(img_u8: torch.Tensor, out: torch.Tensor, c0: int, mean, std):
    """img_u8 [B, H, W, 3] uint8 -> out[:, c0:c0+3] = ((img / 255) - mean) / std  (out [B, Ctot, H, W] bf16 or f32)."""
    assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.dim() == 4 and img_u8.shape[-1] == 3
    B, H, W, _ = img_u8.shape
    assert out.is_contiguous() and out.shape[0] == B and tuple(out.shape[2:]) == (H, W) and out.dtype in (BF16, torch.float32)
    m3, s3 = (C.c_float * 3)(*[float(x) for x in mean]), (C.c_float * 3)(*[float(x) for x in std])
    N.check(_lib().vla_image_normalize_u8(_st(), _p(img_u8), _p(out), B, H, W, out.shape[1], c0, m3, s3,
                                          int(out.dtype == torch.float32)), "image_normalize_u8")
    return out


def action_tokenize(actions_f32: torch.Tensor, bins_f64: torch.Tensor, tokenizer_len: int, lo: float = -1.0, hi: float = 1.0):
    """ActionTokenizer (use_minivlm): int64 ids of the same shape = tokenizer_len - np.digitize(np.clip(a, lo, hi), bins)."""
    assert actions_f32.dtype == torch.float32 and actions_f32.is_contiguous() and bins_f64.dtype == torch.float64 and bins_f64.is_contiguous()
    ids = torch.empty(actions_f32.shape, device=actions_f32.device, dtype=torch.int64)
    N.check(_lib().vla_action_tokenize(_st(), _p(actions_f32), _p(bins_f64), _p(ids), actions_f32.numel(), bins_f64.numel(), lo, hi,
                                       tokenizer_len), "action_tokenize")
    return ids
