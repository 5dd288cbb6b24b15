"""Round-3 kernels of the backbone-training steps against plain fp32 restatements of the same op on the same seeded bf16 inputs:
the TN GEMM (dW = dY^T X on un-transposed operands), the K extension of the NT GEMM (LoRA branch inside the base product),
LayerScale forward / backward, the strided 3-level row copy.  Tolerances as tests/test_kernels_gpu.py: outputs are bf16 with fp32
accumulation -> rel-L2 <= 2e-3, max |diff| <= 2^-6 max|ref|; copies bit-exact."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))

from oracle import vla_oracle as O  # noqa: E402
from test_kernels_gpu import BF, DEV, check, gen, ops  # noqa: E402,F401


# ------------------------------------------------------------------ TN GEMM
@pytest.fixture(params=["128", "256"])
def tn_tile(request, monkeypatch):
    """Both tile geometries of the TN product on every case: the 128 x 128 kernel and the 256 x 256 two-phase kernel (forced through
    VLA_TN_TILE; contraction splits always run on the 128-tile kernel)."""
    monkeypatch.setenv("VLA_TN_TILE", request.param)
    return request.param


@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (256, 2688, 896), (2080, 1792, 896), (300, 200, 136), (1000, 64, 896), (5000, 896, 64),
                                     (31, 8, 8), (129, 264, 72), (128, 512, 512), (192, 256, 256), (5632, 1152, 896)])
@pytest.mark.parametrize("split", [0, None, 3])
def test_gemm_tn_matches_transposed_product(ops, M, N1, N2, split, tn_tile):
    a, b = gen(M, N1, seed=1), gen(M, N2, seed=2, scale=0.1)
    if split == 3 and M < 192:
        pytest.skip("three slices need three K-tiles")
    out = ops.gemm_tn(a.to(DEV), b.to(DEV), split=split)
    check(out, O.rnd(a.float().t() @ b.float(), True), name=f"gemm_tn {M}x{N1}x{N2} split {split}")


def test_gemm_tn_is_deterministic_and_matches_the_nt_kernel_on_transposes(ops, tn_tile):
    """Same fp32 sums in the same K order as the NT kernel on explicitly transposed operands (the round-2 form of dW): the
    two agree to the last bf16 bit up to fp32 summation order inside a K-tile (k permutation) - asserted at 1e-3, and the TN
    kernel itself is run-to-run identical."""
    M, N1, N2 = 2080, 1792, 896
    a, b = gen(M, N1, seed=3).to(DEV), gen(M, N2, seed=4, scale=0.1).to(DEV)
    o1, o2 = ops.gemm_tn(a, b, split=0), ops.gemm_tn(a, b, split=0)
    assert torch.equal(o1, o2)
    Mp = (M + 63) // 64 * 64
    nt = ops.gemm_nt(ops.transpose(a, ld_out=Mp), ops.transpose(b, ld_out=Mp))
    check(o1, nt.float().cpu(), rel=1e-3, name="tn vs nt-on-transposes")


@pytest.mark.parametrize("M,N1,N2", [(64, 256, 256), (128, 256, 512), (192, 512, 256), (2080, 1792, 896), (5632, 896, 4864), (1000, 640, 136), (333, 264, 200)])
def test_gemm_tn256_bit_identical_to_the_128_tile_kernel(ops, M, N1, N2, monkeypatch):
    """Same rows per K-tile, same m order inside a k-step, K-tiles added in the same order: the two tile geometries must agree
    to the last bit - on one- to three-K-tile contractions (the prologue / tail cases of the counted waits), ragged contraction
    tails and ragged outputs, and while a second stream keeps the memory system busy (race screen by repetition)."""
    a, b = gen(M, N1, seed=41).to(DEV), gen(M, N2, seed=42, scale=0.1).to(DEV)
    monkeypatch.setenv("VLA_TN_TILE", "128")
    ref = ops.gemm_tn(a, b, split=0)
    monkeypatch.setenv("VLA_TN_TILE", "256")
    junk, side = torch.empty(64 << 20, dtype=torch.uint8, device=DEV), torch.cuda.Stream()
    outs = []
    for _ in range(10):
        with torch.cuda.stream(side):
            junk.add_(1)
        outs.append(ops.gemm_tn(a, b, split=0))
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
    assert not bad, f"gemm_tn 256 vs 128 tiles {M}x{N1}x{N2}: launches {bad} differ"


def test_gemm_tn_batched_alpha_accumulate(ops, tn_tile):
    nb, M, N1, N2 = 3, 520, 256, 192
    a, b, c0 = gen(nb, M, N1, seed=5), gen(nb, M, N2, seed=6, scale=0.1), gen(nb, N1, N2, seed=7)
    out = c0.to(DEV).clone()
    ops.gemm_tn(a.to(DEV), b.to(DEV), out=out, alpha=2.0, accumulate=True)
    ref = O.rnd(O.rnd(2.0 * torch.einsum("bmi,bmj->bij", a.float(), b.float()), True) + c0.float(), True)
    check(out, ref, name="gemm_tn batched accumulate")
    # strided views (a column window of a wider buffer, as dQKV inside the fused qkv gradient)
    wide = gen(M, 3 * N1, seed=8).to(DEV)
    out2 = ops.gemm_tn(wide[:, N1:2 * N1], b[0].to(DEV))
    check(out2, O.rnd(wide[:, N1:2 * N1].float().cpu().t() @ b[0].float(), True), name="gemm_tn column window")


def test_gemm_tn_row_groups_and_column_groups(ops, tn_tile):
    """Row groups on the contraction: X = the first Kt rows of every sequence of a [B, S, D] tensor, read in place.  Column groups
    on A: the gate (or up) columns of a gate/up-interleaved dY."""
    Bn, S, Kt, D, N1 = 5, 352, 256, 128, 192
    hs = gen(Bn, S, D, seed=9).to(DEV)
    dy = gen(Bn * Kt, N1, seed=10, scale=0.1).to(DEV)
    out = ops.gemm_tn(dy, hs[0, :Kt], rows=Bn * Kt, b_group=(Kt, S * D))
    ref = O.rnd(dy.float().cpu().t() @ hs[:, :Kt].reshape(Bn * Kt, D).float().cpu(), True)
    check(out, ref, name="gemm_tn row groups on B")
    out = ops.gemm_tn(hs[0, :Kt], dy, rows=Bn * Kt, a_group=(Kt, S * D))
    check(out, ref.t(), name="gemm_tn row groups on A")
    I, M = 320, 700
    dgu = gen(M, 2 * I, seed=11).to(DEV)                          # 16-column interleave: [gate 0..15 | up 0..15 | gate 16..31 | ...]
    t = gen(M, 64, seed=12, scale=0.1).to(DEV)
    g3 = dgu.view(M, I // 16, 2, 16)
    for j, name in ((0, "gate"), (1, "up")):
        out = ops.gemm_tn(dgu, t, a_cols=(I, 16, 32, 16 * j))
        ref = O.rnd(g3[:, :, j].reshape(M, I).float().cpu().t() @ t.float().cpu(), True)
        check(out, ref, name=f"gemm_tn column groups ({name})")


# ------------------------------------------------------------------ K extension of the NT GEMM
@pytest.mark.parametrize("M,N,K,K2", [(300, 200, 192, 64), (1000, 896, 896, 128), (2048, 1152, 896, 192), (64, 8, 64, 64)])
def test_gemm_k_extension(ops, M, N, K, K2):
    a, b, a2, b2 = gen(M, K, seed=1), gen(N, K, seed=2, scale=0.05), gen(M, K2, seed=3), gen(N, K2, seed=4, scale=0.05)
    bias, r = gen(N, seed=5), gen(M, N, seed=6)
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=r.to(DEV), ext=(a2.to(DEV), b2.to(DEV)))
    y = O.rnd(a.float() @ b.float().t() + a2.float() @ b2.float().t() + bias.float(), True)
    check(out, O.rnd(y + r.float(), True), name=f"gemm ext {M}x{N}x{K}+{K2}")
    # the extension is the same accumulator: identical to one GEMM over the concatenated operands
    cat = ops.gemm_nt(torch.cat([a, a2], 1).to(DEV), torch.cat([b, b2], 1).to(DEV), bias=bias.to(DEV), residual=r.to(DEV))
    if M < 1024 or N < 768:                    # (large problems route the concatenated form to the 256-row kernel: same sums, same order)
        assert torch.equal(out, cat)
    else:
        check(out, cat.float().cpu(), rel=1e-3, name="ext vs concatenated")


def test_gemm_k_extension_keeps_the_fused_epilogues(ops):
    """SwiGLU forward, rotate_half RoPE and the SwiGLU-backward epilogue on an extended product == the same epilogue on the
    concatenated operands (bit for bit: same kernel geometry, same K order)."""
    M, I, K, K2 = 330, 320, 256, 128
    x, w, x2, w2 = gen(M, K, seed=12), gen(2 * I, K, seed=13, scale=0.1), gen(M, K2, seed=14), gen(2 * I, K2, seed=15, scale=0.1)
    d = lambda t: t.to(DEV)
    import os
    os.environ["VLA_GEMM_TILE"] = "2"          # the 128-row 8-wave geometry for the concatenated reference too
    try:
        pre, h = ops.gemm_nt(d(x), d(w), act=ops.ACT_SWIGLU, ext=(d(x2), d(w2)))
        pre_c, h_c = ops.gemm_nt(d(torch.cat([x, x2], 1)), d(torch.cat([w, w2], 1)), act=ops.ACT_SWIGLU)
        assert torch.equal(pre, pre_c) and torch.equal(h, h_c)
        S, H, dh = 33, 5, 64
        cos, sin = ops.rope_half_tables(S, dh, 1e6, DEV)
        Mr, N = 10 * S, (H + 2) * dh
        a, b, a2, b2 = gen(Mr, K, seed=16), gen(N, K, seed=17, scale=0.1), gen(Mr, K2, seed=18), gen(N, K2, seed=19, scale=0.1)
        bias = gen(N, seed=20)
        o = ops.gemm_nt(d(a), d(b), bias=d(bias), rope=(1, cos, sin, S, dh, (H + 1) * dh), ext=(d(a2), d(b2)))
        o_c = ops.gemm_nt(d(torch.cat([a, a2], 1)), d(torch.cat([b, b2], 1)), bias=d(bias), rope=(1, cos, sin, S, dh, (H + 1) * dh))
        assert torch.equal(o, o_c)
        gu = gen(M, 2 * I, seed=21)
        dy, wdT, dt, AT = gen(M, K, seed=22), gen(I, K, seed=23, scale=0.1), gen(M, K2, seed=24), gen(I, K2, seed=25, scale=0.1)
        g1 = ops.gemm_swiglu_bwd(d(dy), d(wdT), d(gu), ext=(d(dt), d(AT)))
        g2 = ops.gemm_swiglu_bwd(d(torch.cat([dy, dt], 1)), d(torch.cat([wdT, AT], 1)), d(gu))
        assert torch.equal(g1, g2)
    finally:
        del os.environ["VLA_GEMM_TILE"]


@pytest.mark.parametrize("M,N,K,K2,kind", [(4096, 1152, 1152, 64, "plain"), (5632, 896, 896, 192, "res"), (2100, 1000, 448, 128, "res"), (1024, 768, 64, 64, "gelu"),
                                           (5632, 1152, 896, 192, "rope"), (2200, 2432, 896, 128, "swiglu"), (4096, 4352, 1152, 64, "gelu")])
def test_gemm256_k_extension_bit_identical(ops, M, N, K, K2, kind, monkeypatch):
    """Round 4: the K extension on the 256 x 256 kernel (its EXT instantiations: the K-tiles behind K / 64 staged from A2 / B2) - bit for bit
    the 128-row kernel's extended product (same K order, same MFMA sequence, same epilogue), every epilogue a LoRA-wrapped Linear uses."""
    d = lambda t: t.to(DEV)
    a, b, a2, b2 = d(gen(M, K, seed=31)), d(gen(N, K, seed=32, scale=0.05)), d(gen(M, K2, seed=33)), d(gen(N, K2, seed=34, scale=0.05))
    kw = dict(ext=(a2, b2))
    if kind in ("plain", "res", "gelu", "rope"):
        kw["bias"] = d(gen(N, seed=35))
    if kind == "res":
        kw["residual"] = d(gen(M, N, seed=36))
    if kind == "gelu":
        kw["act"] = 1
    if kind == "rope":
        S = 352
        cos, sin = ops.rope_half_tables(S, 64, 1e6, DEV)
        kw["rope"] = (1, cos, sin, S, 64, 1024)
    if kind == "swiglu":
        kw["act"] = ops.ACT_SWIGLU
    monkeypatch.setenv("VLA_GEMM_TILE", "2")
    ref = ops.gemm_nt(a, b, **kw)
    monkeypatch.setenv("VLA_GEMM_TILE", "6")
    assert ops.gemm_nt(a, b, query_256=True, **kw), "the 256-row kernel refused the extended product"
    out = ops.gemm_nt(a, b, **kw)
    monkeypatch.delenv("VLA_GEMM_TILE")
    if kind == "swiglu":
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    else:
        assert torch.equal(out, ref), f"{(out.float() - ref.float()).abs().max().item()}"
    auto = ops.gemm_nt(a, b, **kw)                                        # whichever kernel the routing picks: the same bits
    assert torch.equal(auto[1] if kind == "swiglu" else auto, ref[1] if kind == "swiglu" else ref)


def _deq(q, s):
    return q.cpu().view(torch.float8_e4m3fn).float() * s.cpu()[:, None]


@pytest.mark.parametrize("M,N,K,K2", [(300, 200, 256, 64), (1000, 896, 896, 128), (2048, 1152, 1152, 192), (64, 8, 128, 64), (640, 1536, 8960, 64)])
def test_gemm_fp8_base_with_bf16_k_extension(ops, M, N, K, K2):
    """ABI 5: e4m3 base operands + bf16 extension in ONE accumulator (a LoRA-wrapped Linear whose frozen base weight runs on the fp8
    MFMA): C = epilogue(sa[m] sb[n] (Aq . Bq^T) + A2 . B2^T) against the fp32 product of the DEQUANTISED base operands plus the bf16
    extension.  The reference has no fp8 code: parity unpinned, the kernel is held to the arithmetic it claims."""
    a, b, a2, b2 = gen(M, K, seed=31), gen(N, K, seed=32, scale=0.05), gen(M, K2, seed=33), gen(N, K2, seed=34, scale=0.05)
    bias, r = gen(N, seed=35), gen(M, N, seed=36)
    qa, sa = ops.quant_fp8_rows(a.to(DEV))
    qb, sb = ops.quant_fp8_rows(b.to(DEV))
    out = ops.gemm_nt(qa, qb, bias=bias.to(DEV), residual=r.to(DEV), fp8=(sa, sb), ext=(a2.to(DEV), b2.to(DEV)))
    y = O.rnd(_deq(qa, sa) @ _deq(qb, sb).t() + a2.float() @ b2.float().t() + bias.float(), True)
    check(out, O.rnd(y + r.float(), True), name=f"fp8 gemm + bf16 ext {M}x{N}x{K}+{K2}")
    # the extension really adds to the DEQUANTISED product: with unit scales forced, the result must differ
    wrong = ops.gemm_nt(qa, qb, bias=bias.to(DEV), residual=r.to(DEV), fp8=(torch.ones_like(sa), torch.ones_like(sb)), ext=(a2.to(DEV), b2.to(DEV)))
    assert not torch.equal(out, wrong)


def test_gemm_fp8_k_extension_keeps_the_fused_epilogues(ops):
    """SwiGLU forward, rotate_half RoPE and the SwiGLU-backward epilogue on the fp8 + extension form."""
    M, I, K, K2 = 330, 320, 256, 128
    x, w, x2, w2 = gen(M, K, seed=42), gen(2 * I, K, seed=43, scale=0.1), gen(M, K2, seed=44), gen(2 * I, K2, seed=45, scale=0.1)
    d = lambda t: t.to(DEV)
    qa, sa = ops.quant_fp8_rows(d(x))
    qb, sb = ops.quant_fp8_rows(d(w))
    pre, h = ops.gemm_nt(qa, qb, act=ops.ACT_SWIGLU, fp8=(sa, sb), ext=(d(x2), d(w2)))
    y = O.rnd(_deq(qa, sa) @ _deq(qb, sb).t() + x2.float() @ w2.float().t(), True)
    check(pre, y, name="fp8 + ext swiglu pre-activations")
    g, u = y.view(M, I // 16, 2, 16)[:, :, 0].reshape(M, I), y.view(M, I // 16, 2, 16)[:, :, 1].reshape(M, I)
    check(h, O.rnd(O.rnd(g * torch.sigmoid(g), True) * u, True), rel=6e-3, name="fp8 + ext swiglu h")
    S, H, dh = 33, 5, 64
    cos, sin = ops.rope_half_tables(S, dh, 1e6, DEV)
    Bq = 10
    Mr, N = Bq * S, (H + 2) * dh
    a, b, a2, b2 = gen(Mr, K, seed=46), gen(N, K, seed=47, scale=0.1), gen(Mr, K2, seed=48), gen(N, K2, seed=49, scale=0.1)
    bias = gen(N, seed=50)
    qa, sa = ops.quant_fp8_rows(d(a))
    qb, sb = ops.quant_fp8_rows(d(b))
    o = ops.gemm_nt(qa, qb, bias=d(bias), rope=(1, cos, sin, S, dh, (H + 1) * dh), fp8=(sa, sb), ext=(d(a2), d(b2)))
    y = O.rnd(_deq(qa, sa) @ _deq(qb, sb).t() + a2.float() @ b2.float().t() + bias.float(), True)
    c, s_ = O.rope_half_tables(S, dh, 1e6, True)
    rot = O.rope_half(y[:, :(H + 1) * dh].view(Bq, S, H + 1, dh).transpose(1, 2), c, s_, True).transpose(1, 2).reshape(Mr, (H + 1) * dh)
    check(o, torch.cat([rot, y[:, (H + 1) * dh:]], 1), name="fp8 + ext gemm + rope_half")
    gu = gen(M, 2 * I, seed=51)
    dy, wdT, dt, AT = gen(M, K, seed=52), gen(I, K, seed=53, scale=0.1), gen(M, K2, seed=54), gen(I, K2, seed=55, scale=0.1)
    qd, sd = ops.quant_fp8_rows(d(dy))
    qw, sw = ops.quant_fp8_rows(d(wdT))
    g1 = ops.gemm_swiglu_bwd(qd, qw, d(gu), ext=(d(dt), d(AT)), fp8=(sd, sw))
    dh_ = O.rnd(_deq(qd, sd) @ _deq(qw, sw).t() + dt.float() @ AT.float().t(), True)          # dH [M, I], rounded as the epilogue does
    gg = gu.float().view(M, I // 16, 2, 16)
    gate, up = gg[:, :, 0].reshape(M, I), gg[:, :, 1].reshape(M, I)
    sg = torch.sigmoid(gate)
    dgate, dup = dh_ * up * (sg * (1 + gate * (1 - sg))), dh_ * gate * sg
    ref = torch.stack([dgate.view(M, I // 16, 16), dup.view(M, I // 16, 16)], 2).reshape(M, 2 * I)
    check(g1, O.rnd(ref, True), rel=6e-3, name="fp8 + ext swiglu backward")


# ------------------------------------------------------------------ LayerScale, row copies
@pytest.mark.parametrize("rows,cols", [(261 * 3, 1024), (40, 192), (1000, 8)])
def test_layerscale_forward_backward(ops, rows, cols):
    a, x, ls, dy = gen(rows, cols, seed=1), gen(rows, cols, seed=2), gen(cols, seed=3, scale=0.3), gen(rows, cols, seed=4)
    out = ops.layerscale_fwd(a.to(DEV), ls.to(DEV), x.to(DEV))
    ref = O.rnd(x.float() + O.rnd(a.float() * ls.float(), True), True)
    assert torch.equal(out.cpu(), ref.to(BF)), "elementwise with the reference's two roundings: bit-exact"
    dls = torch.zeros(cols, device=DEV, dtype=torch.float32)
    da = ops.layerscale_bwd(dy.to(DEV), a.to(DEV), ls.to(DEV), dls)
    assert torch.equal(da.cpu(), O.rnd(dy.float() * ls.float(), True).to(BF))
    ref_dls = (dy.float() * a.float()).sum(0)
    assert (dls.cpu() - ref_dls).abs().max().item() <= 1e-4 * (ref_dls.abs().max().item() + 1.0) * rows ** 0.5
    da2 = ops.layerscale_bwd(dy.to(DEV), None, ls.to(DEV), None)                # frozen scale (LoRA): dx only
    assert torch.equal(da2, da)


def test_copy_rows3d(ops):
    Bn, Np, vis, d, npi, T = 3, 32, 320, 192, 16, 21
    feats = gen(Bn, Np, vis, seed=5).to(DEV)
    # image 1 of backbone columns [128, 320) -> rows n_prefix.. of a [B, T, d] token buffer
    dst = torch.zeros(Bn, T, d, device=DEV, dtype=BF)
    ops.copy_rows3d(feats[0, npi:, 128:], dst[0, T - npi:], Bn, npi, d, Np * vis, vis, T * d, d)
    assert torch.equal(dst[:, T - npi:], feats[:, npi:, 128:]) and bool((dst[:, :T - npi] == 0).all())
    # odd widths take the element path
    src = gen(4, 7, 13, seed=6).to(DEV)
    dst = torch.zeros(4, 9, 13, device=DEV, dtype=BF)
    ops.copy_rows3d(src, dst[0, 1], 4, 7, 13, 7 * 13, 13, 9 * 13, 13)
    assert torch.equal(dst[:, 1:8], src)


def test_gemm_tn_grouped_equals_the_single_launches(ops, tn_tile):
    """One grouped launch over several products (different shapes, a column-grouped operand, a ragged contraction) == the same
    products launched one by one, bit for bit (same tiles, same K order)."""
    M = 1000
    dy = gen(M, 640, seed=31).to(DEV)
    xs = [gen(M, n, seed=32 + i, scale=0.1).to(DEV) for i, n in enumerate((896, 64, 136))]
    probs, refs, outs = [], [], []
    for i, x in enumerate(xs):
        o = torch.empty(640, x.shape[1], device=DEV, dtype=BF)
        probs.append(ops.tn_problem(dy, x, o, alpha=1.0 + i))
        refs.append(ops.gemm_tn(dy, x, alpha=1.0 + i, split=0))
        outs.append(o)
    o = torch.empty(320, 64, device=DEV, dtype=BF)                       # the "up" half of a gate/up-interleaved dY
    probs.append(ops.tn_problem(dy, xs[1], o, a_cols=(320, 16, 32, 16)))
    refs.append(ops.gemm_tn(dy, xs[1], a_cols=(320, 16, 32, 16), split=0))
    outs.append(o)
    short = gen(200, 128, seed=40).to(DEV)                                 # another contraction length in the same launch
    o = torch.empty(128, 128, device=DEV, dtype=BF)
    probs.append(ops.tn_problem(short, short, o))
    refs.append(ops.gemm_tn(short, short, split=0))
    outs.append(o)
    ops.gemm_tn_grouped(probs)
    for a, b in zip(outs, refs):
        assert torch.equal(a, b)
    many = [ops.tn_problem(short, short, torch.empty(128, 128, device=DEV, dtype=BF)) for _ in range(ops.TN_GROUP_MAX + 5)]   # > one table
    ops.gemm_tn_grouped(many)
    assert all(torch.equal(p._keep[2], refs[-1]) for p in many)
