// bf16 TN GEMM for gfx950:  C[N1,N2] = alpha * sum_m A[m,N1] * B[m,N2]  (+ R)  - the contraction runs over the ROWS of both
// operands.  This is the weight-gradient product of every nn.Linear, dW = dY^T . X (torch.autograd's `grad_output.t().mm(input)`
// behind vla-scripts/finetune.py:1039-1042 for the action head's Linears, the LoRA pairs (:832-844) and - full fine-tune,
// :846-849 - every Linear of the VLM), on the operands exactly as the forward / backward left them: dY [rows, out] and
// X [rows, in], both row-major.  Rounds 1-2 fed these products to the NT kernel through explicit transposes of dY and X
// (228 MB per LLM layer in the full fine-tune, 0.7 ms per adapter-only step); here the transposition happens in the LDS read.
//
// Structure:
//   * block tile 128 (n1) x 128 (n2) x 64 (m); 8 waves as 2 x 4, wave tile 64 x 32, v_mfma_f32_16x16x32_bf16;
//   * both operand tiles are [64 m][128 columns] row-major in LDS (256-B rows), filled by global_load_lds_dwordx4 (one piece =
//     4 rows) with the 16-B-chunk XOR of cdna_hip_programming.md T10 image (b) applied on the SOURCE address: LDS chunk c of
//     row r holds global chunk c ^ f(r), f(r) = ((r & 3) << 2) | ((r >> 2) & 3);
//   * MFMA fragments (8 consecutive m for one column) come from ds_read_b64_tr_b16 - the hardware transposed read: group kq of
//     16 lanes reads rows 32 ks + 8 kq + {0..3} and {4..7} of a 16-column block, lane i receives column i.  A and B fragments
//     use the same m permutation, so any order of m inside a k-step is fine.  With a 32-lane half's two blocks 8 rows apart in
//     the same columns the read is conflict-free on this image (T10);
//   * 2 stages, one barrier per K-tile, two workgroups per CU (64 KiB LDS each) hide each other's fills and epilogues;
//   * the contraction tail (M % 64) is zeroed in the B-side fragments of the last K-tile (sources clamped to valid rows);
//   * row groups on the contraction rows (row m at (m / g) * stride + (m % g) * ld, g % 64 == 0: a K-tile never straddles a
//     group): X = the first Kt rows of every sequence of a [B, S, D] hidden state, read in place;
//   * column groups on A (column c at (c / g) * stride + c % g): the gate or the up columns of an interleaved dGU;
//   * split over the contraction (blockIdx.z slices, fp32 planes + a finalize pass) for few-tile long-M products.
#include "common.h"
#include "../../include/vla_native.h"

namespace {

constexpr int TBK = 64;                       // contraction rows per K-tile
constexpr int TILE_BYTES = TBK * 256;         // one operand tile: 64 rows x 128 bf16
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 64 KiB

struct TnP {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const bf16_t* R; float* ws;
  int M, N1, N2, lda, ldb, ldc, ldr;
  long long sA, sB, sC, sR;
  float alpha;
  int gA, gB; long long sgA, sgB;             // contraction-row groups
  int cgA, cgsA;                              // column groups on A (elements)
  int tiles_n2, ntiles, split, mslice;
};

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

// XCD-aware bijective order over `nwg` workgroups (as gemm.hip): workgroups b and b + 8 share an XCD (round-robin dispatch); every
// XCD gets a contiguous run of the tile list, so consecutive tiles (same A column panel) hit the same L2
__device__ __forceinline__ int xcd_order(int bid, int nwg) {
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

// One 128 x 128 output tile (tile index `swz` of problem p; bz = batch * split + contraction slice).
__device__ __forceinline__ void tn_tile(const TnP& p, const int swz, const int bz, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int b1 = swz / p.tiles_n2, b2 = swz - b1 * p.tiles_n2;
  const int n1_0 = b1 * 128, n2_0 = b2 * 128;
  const int zb = bz / p.split, zs = bz - zb * p.split;
  const int m_begin = zs * p.mslice, m_end = min(p.M, m_begin + p.mslice);
  const int nt = (m_end - m_begin + TBK - 1) / TBK;
  const char* Ab = reinterpret_cast<const char*>(p.A + (long long)zb * p.sA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (long long)zb * p.sB);

  // ---- staging.  A stage = 32 pieces of 1 KiB (4 rows x 256 B): pieces 0..15 the A tile, 16..31 the B tile; wave w fills
  //      pieces 4w .. 4w+3 (waves 0-3: A, waves 4-7: B).  Lane -> row 4 q + (lane >> 4), LDS chunk lane & 15.
  const bool isA = wid < 4;
  const int ld = isA ? p.lda : p.ldb, ncols = isA ? p.N1 : p.N2, c0 = isA ? n1_0 : n2_0;
  const int grp = isA ? p.gA : p.gB;
  const long long sgrp = isA ? p.sgA : p.sgB;
  const char* base = isA ? Ab : Bb;
  int rloc[4];
  long long coff[4];                          // byte offset of this lane's 16-B source chunk inside a row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qp = (wid & 3) * 4 + i;
    const int r = 4 * qp + (lane >> 4);
    const int f = ((r & 3) << 2) | ((r >> 2) & 3);
    int col = c0 + (((lane & 15) ^ f) << 3);
    col = min(col, ncols - 8);                // columns beyond the matrix: any valid chunk (those outputs are never stored)
    if (isA && p.cgA > 0) col = (col / p.cgA) * p.cgsA + col % p.cgA;
    rloc[i] = r;
    coff[i] = (long long)col * 2;
  }
  auto stage = [&](int buf, int t) {
    const int mt = m_begin + t * TBK;                                      // wave-uniform
    const long long tb = grp > 0 ? (long long)(mt / grp) * sgrp + (long long)(mt % grp) * ld : (long long)mt * ld;
    const int lim = m_end - 1 - mt;                                        // last valid row of this K-tile (>= 0)
    char* dst = smem + buf * STAGE_BYTES + (isA ? 0 : TILE_BYTES) + (wid & 3) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(base + (tb + (long long)min(rloc[i], lim) * ld) * 2 + coff[i], dst + i * 1024);
  };

  // ---- transposed fragment reads: lane (kq = lane >> 4, i = lane & 15 = 4 qq + pp) supplies row 8 kq + qq (+ 4 e), columns
  //      cb + 4 pp .. + 3 of the 16-column block cb; byte = 256 row + 16 ((cb / 8 + (pp >> 1)) ^ f) + 8 (pp & 1),
  //      f = (qq << 2) | ((2 kq + e) & 3)   (32 ks rows further for k-step ks: + 8192 ks, f unchanged)
  const int kq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  int fa[4][2], fb[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 8 * kq + qq + 4 * e, f = (qq << 2) | ((2 * kq + e) & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i][e] = 256 * row + ((((wr * 64 + i * 16) >> 3) + (pp >> 1)) ^ f) * 16 + 8 * (pp & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) fb[i][e] = TILE_BYTES + 256 * row + ((((wc * 32 + i * 16) >> 3) + (pp >> 1)) ^ f) * 16 + 8 * (pp & 1);
  }

  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nt > 0) stage(0, 0);
  int buf = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // K-tile t visible to every wave; every wave has finished reading the other buffer
    asm volatile("" ::: "memory");
    if (t + 1 < nt) stage(buf ^ 1, t + 1);
    const char* sb = smem + buf * STAGE_BYTES;
    const int valid = m_end - (m_begin + t * TBK);          // rows of this K-tile inside the contraction range (wave-uniform)
    bf16x8 fm[2][4], fn[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = tr_read(sb + ks * 8192 + fa[i][0]), hi = tr_read(sb + ks * 8192 + fa[i][1]);
        fm[ks][i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x4 lo = tr_read(sb + ks * 8192 + fb[i][0]), hi = tr_read(sb + ks * 8192 + fb[i][1]);
        fn[ks][i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
    if (valid < TBK) {                     // contraction tail: rows >= valid contribute nothing (A side holds clamped, finite rows)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (ks * 32 + 8 * kq + e >= valid) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fn[ks][i][e] = 0;
          }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[ks][ni], fm[ks][mi], acc[ni][mi], 0, 0, 0);
    buf ^= 1;
  }

  // ---- epilogue: lane owns, per (ni, mi): row n1 = 16 mi + (lane & 15), columns n2 = 16 ni + 4 (lane >> 4) + {0..3}
  const int w1 = n1_0 + wr * 64, w2 = n2_0 + wc * 32;
  const int lq = lane >> 4, lr = lane & 15;
  if (p.ws != nullptr) {                   // contraction slice zs: raw fp32 accumulators into its own plane
    float* plane = p.ws + ((long long)bz) * p.N1 * p.N2;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int r = w1 + mi * 16 + lr, c = w2 + ni * 16 + lq * 4;
        if (r < p.N1 && c + 3 < p.N2) *reinterpret_cast<f32x4*>(plane + (long long)r * p.N2 + c) = acc[ni][mi];
      }
    return;
  }
  bf16_t* Cb = p.C + (long long)zb * p.sC;
  const bf16_t* Rb = p.R ? p.R + (long long)zb * p.sR : nullptr;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int r = w1 + mi * 16 + lr, c = w2 + ni * 16 + lq * 4;
      if (r >= p.N1 || c + 3 >= p.N2) continue;            // (N2 % 8 == 0: a lane's four columns are all inside or all outside)
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][j] * p.alpha;
      if (Rb) {
        const uint2 rv = *reinterpret_cast<const uint2*>(Rb + (long long)r * p.ldr + c);
        v[0] = rbf(v[0]) + bf2f((bf16_t)(rv.x & 0xffff)); v[1] = rbf(v[1]) + bf2f((bf16_t)(rv.x >> 16));
        v[2] = rbf(v[2]) + bf2f((bf16_t)(rv.y & 0xffff)); v[3] = rbf(v[3]) + bf2f((bf16_t)(rv.y >> 16));
      }
      *reinterpret_cast<uint2*>(Cb + (long long)r * p.ldc + c) = uint2{pack2(v[0], v[1]), pack2(v[2], v[3])};
    }
}

// ================================================================================================ 256 x 256 tile (round 3)
// The same product on gemm256.hip's skeleton: 256 (n1) x 256 (n2) output tile, 8 waves as 2 x 4, wave tile 128 x 64 = four
// 64 x 32 quadrants, operands in LDS as eight 16-KiB half-tiles (2 K-tile buffers x {B0, A0, B1, A1}), each the [64 m][128
// columns] image of the 128-tile kernel above (A-half h = columns {wr' 128 + h 64 + [0, 64)}, B-half h = columns {wc' 64 + h 32 +
// [0, 32)}: a wave reads its columns of a half-tile exactly as it reads them of a 128-tile), the two-phase K loop with its
// segment-counted LDS-DMA schedule (gemm256.hip, file header: phase X reads B0, A0, B1 and multiplies Q00, Q01; phase Y reads
// A1 and multiplies Q11, Q10; every half-tile is issued four segments before the counted wait that retires it), one workgroup per
// CU and per tile.  A weight gradient contracts over thousands of rows (88 K-tiles at batch 16), so what pays is the K loop:
// half the staged bytes per FLOP of the 128-tile and no wait on the fill.  Plain, batched, row-grouped, column-grouped and
// accumulating problems; the contraction split stays on the 128-tile kernel.
constexpr int T2_HT = 16384;
constexpr int T2_LDS = 8 * T2_HT;

// LDS-DMA from inline asm (see gemm256.hip: a builtin global_load_lds makes hipcc drain vmcnt(0) around it; counted by hand below)
__device__ __forceinline__ void glds16s(const char* base, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

#define TN_BARRIER()                       \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

#define TN_MMA_QUADRANT(Q, FN, FM)                                                                             \
  do {                                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                         \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                       \
          Q[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FN[ks][ni], FM[ks][mi], Q[ni][mi], 0, 0, 0);     \
  } while (0)

__device__ __forceinline__ void tn256_tile(const TnP& p, const int swz, const int zb, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int b1 = swz / p.tiles_n2, b2 = swz - b1 * p.tiles_n2;
  const int n1_0 = b1 * 256, n2_0 = b2 * 256;
  const int nt = (p.M + TBK - 1) / TBK;
  const char* Ab = reinterpret_cast<const char*>(p.A + (long long)zb * p.sA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (long long)zb * p.sB);

  // ---- staging: wave w fills pieces 2w, 2w + 1 (rows 8w .. 8w + 7) of every half-tile; lane -> row + (lane >> 4), LDS chunk
  //      lane & 15 holds source chunk (lane & 15) ^ f(row) of the half-tile's 128 columns
  const int r0 = wid * 8 + (lane >> 4);
  unsigned ca[2][2], cb[2][2];                 // byte offset of this lane's source chunk inside a row: [half][piece]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = r0 + 4 * j;
    const int f = ((r & 3) << 2) | ((r >> 2) & 3);
    const int lc = ((lane & 15) ^ f) << 3;     // first of this lane's 8 columns inside the half-tile
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ga = n1_0 + (lc >> 6) * 128 + h * 64 + (lc & 63);
      ga = min(ga, p.N1 - 8);                  // columns beyond the matrix: any valid chunk (those outputs are never stored)
      if (p.cgA > 0) ga = (ga / p.cgA) * p.cgsA + ga % p.cgA;
      ca[h][j] = (unsigned)ga * 2u;
      const int gb = min(n2_0 + (lc >> 5) * 64 + h * 32 + (lc & 31), p.N2 - 8);
      cb[h][j] = (unsigned)gb * 2u;
    }
  }
  const unsigned wdst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + wid * 2048;
  auto row_base = [&](int mt, int grp, long long sgrp, int ld) -> long long {          // wave-uniform, elements
    return grp > 0 ? (long long)(mt / grp) * sgrp + (long long)(mt % grp) * ld : (long long)mt * ld;
  };
  auto stage_a = [&](int slot, int h, int t) {
    const int mt = t * TBK;
    const char* base = Ab + row_base(mt, p.gA, p.sgA, p.lda) * 2;
    const unsigned lim = (unsigned)(p.M - 1 - mt), ld2 = (unsigned)p.lda * 2u;          // contraction tail: clamped to the last valid row
    glds16s(base, min((unsigned)r0, lim) * ld2 + ca[h][0], wdst + slot * T2_HT);
    glds16s(base, min((unsigned)r0 + 4u, lim) * ld2 + ca[h][1], wdst + slot * T2_HT + 1024);
  };
  auto stage_b = [&](int slot, int h, int t) {
    const int mt = t * TBK;
    const char* base = Bb + row_base(mt, p.gB, p.sgB, p.ldb) * 2;
    const unsigned lim = (unsigned)(p.M - 1 - mt), ld2 = (unsigned)p.ldb * 2u;
    glds16s(base, min((unsigned)r0, lim) * ld2 + cb[h][0], wdst + slot * T2_HT);
    glds16s(base, min((unsigned)r0 + 4u, lim) * ld2 + cb[h][1], wdst + slot * T2_HT + 1024);
  };

  // ---- transposed fragment reads (as tn_tile): byte offsets inside a half-tile, second k-step 8192 further
  const int kq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  int fa[4][2], fb[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 8 * kq + qq + 4 * e, f = (qq << 2) | ((2 * kq + e) & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i][e] = 256 * row + ((((wr * 64 + i * 16) >> 3) + (pp >> 1)) ^ f) * 16 + 8 * (pp & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) fb[i][e] = 256 * row + ((((wc * 32 + i * 16) >> 3) + (pp >> 1)) ^ f) * 16 + 8 * (pp & 1);
  }
  auto read_a = [&](const char* sb, bf16x8 (&fm)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = tr_read(sb + ks * 8192 + fa[i][0]), hi = tr_read(sb + ks * 8192 + fa[i][1]);
        fm[ks][i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  };
  auto read_b = [&](const char* sb, bf16x8 (&fn)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x4 lo = tr_read(sb + ks * 8192 + fb[i][0]), hi = tr_read(sb + ks * 8192 + fb[i][1]);
        fn[ks][i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  };
  auto mask_tail = [&](bf16x8 (&fn)[2][2], int valid) {      // rows >= valid of the last K-tile contribute nothing
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ks * 32 + 8 * kq + e >= valid) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fn[ks][i][e] = 0;
        }
  };

  // ---- prologue: K-tile 0 (all four half-tiles) and B0, A0, B1 of K-tile 1
  stage_b(0, 0, 0); stage_a(1, 0, 0); stage_b(2, 1, 0); stage_a(3, 1, 0);
  if (nt > 1) {
    stage_b(4, 0, 1); stage_a(5, 0, 1); stage_b(6, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  TN_BARRIER();
  if (wr == 1) TN_BARRIER();         // stagger: the wr = 1 waves run one segment behind

  f32x4 acc[2][2][2][4];             // [mh][nh][ni][mi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][b][c][e] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 fm[2][4], fn0[2][2], fn1[2][2];
  int d = 0;
  for (int t = 0; t < nt; ++t) {
    const char* kb = smem + d * 4 * T2_HT;
    const int so = d * 4, sn = (d ^ 1) * 4;
    const int valid = p.M - t * TBK;                         // rows of this K-tile inside the contraction (wave-uniform)
    // ================= phase X: reads B0, A0, B1; issues A1 of K-tile t+1; retires A1 of K-tile t
    {
      read_b(kb + 0 * T2_HT, fn0);
      read_a(kb + 1 * T2_HT, fm);
      read_b(kb + 2 * T2_HT, fn1);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nt) {
        stage_a(sn + 3, 1, t + 1);
        if (t > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (t > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (valid < TBK) { mask_tail(fn0, valid); mask_tail(fn1, valid); }
      TN_BARRIER();
      __builtin_amdgcn_s_setprio(1);
      TN_MMA_QUADRANT(acc[0][0], fn0, fm);
      TN_MMA_QUADRANT(acc[0][1], fn1, fm);
      __builtin_amdgcn_s_setprio(0);
      TN_BARRIER();
    }
    // ================= phase Y: reads A1; issues B0, A0, B1 of K-tile t+2; retires B0, A0, B1 of K-tile t+1
    {
      read_a(kb + 3 * T2_HT, fm);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nt) {
        stage_b(so + 0, 0, t + 2); stage_a(so + 1, 0, t + 2); stage_b(so + 2, 1, t + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (t + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TN_BARRIER();
      __builtin_amdgcn_s_setprio(1);
      TN_MMA_QUADRANT(acc[1][1], fn1, fm);
      TN_MMA_QUADRANT(acc[1][0], fn0, fm);
      __builtin_amdgcn_s_setprio(0);
      TN_BARRIER();
    }
    d ^= 1;
  }
  if (wr == 0) TN_BARRIER();         // pairs with the last barrier of the wr = 1 waves

  // ---- epilogue: lane owns, per (mh, nh, ni, mi): row n1 = 16 mi + (lane & 15), columns n2 = 16 ni + 4 (lane >> 4) + {0..3}
  const int lq = lane >> 4, lr = lane & 15;
  bf16_t* Cb = p.C + (long long)zb * p.sC;
  const bf16_t* Rb = p.R ? p.R + (long long)zb * p.sR : nullptr;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int r = n1_0 + wr * 128 + mh * 64 + mi * 16 + lr, c = n2_0 + wc * 64 + nh * 32 + ni * 16 + lq * 4;
          if (r >= p.N1 || c + 3 >= p.N2) continue;        // (N2 % 8 == 0: a lane's four columns are all inside or all outside)
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mh][nh][ni][mi][j] * p.alpha;
          if (Rb) {
            const uint2 rv = *reinterpret_cast<const uint2*>(Rb + (long long)r * p.ldr + c);
            v[0] = rbf(v[0]) + bf2f((bf16_t)(rv.x & 0xffff)); v[1] = rbf(v[1]) + bf2f((bf16_t)(rv.x >> 16));
            v[2] = rbf(v[2]) + bf2f((bf16_t)(rv.y & 0xffff)); v[3] = rbf(v[3]) + bf2f((bf16_t)(rv.y >> 16));
          }
          *reinterpret_cast<uint2*>(Cb + (long long)r * p.ldc + c) = uint2{pack2(v[0], v[1]), pack2(v[2], v[3])};
        }
}

__global__ __launch_bounds__(512) void gemm_tn256_kernel(TnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tn256_tile(p, xcd_order(blockIdx.x, p.ntiles), blockIdx.z, smem);
}

__global__ __launch_bounds__(512) void gemm_tn_kernel(TnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tn_tile(p, xcd_order(blockIdx.x, p.ntiles), blockIdx.z, smem);
}

// GROUPED launch: up to TN_GROUP_MAX independent products (the weight gradients of several Linears / layers) as ONE tile list.
// A dW product alone is 49 ... 532 tiles on a chip with 512 workgroup slots - its tail round leaves up to half the CUs idle; the
// products of a whole backward piece together (3600 tiles for four LLM layers) run at the tile list's own granularity.  The
// problem table travels by value in the kernel arguments (no device-side state, capturable in a hipGraph).
constexpr int TN_GROUP_MAX = 48;
struct TnG {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  int M, N1, N2, lda, ldb, ldc, cgA, cgsA;
  float alpha; int tiles_n2;
};
struct TnGroup { TnG g[TN_GROUP_MAX]; int start[TN_GROUP_MAX + 1]; int count, total; };

__global__ __launch_bounds__(512) void gemm_tn_grouped_kernel(TnGroup G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_order(blockIdx.x, G.total);
  int pi = 0;
  while (pi + 1 < G.count && t >= G.start[pi + 1]) ++pi;          // wave-uniform scan of the (short) table
  const TnG& g = G.g[pi];
  TnP p;
  p.A = g.A; p.B = g.B; p.C = g.C; p.R = nullptr; p.ws = nullptr;
  p.M = g.M; p.N1 = g.N1; p.N2 = g.N2; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldr = 0;
  p.sA = p.sB = p.sC = p.sR = 0;
  p.alpha = g.alpha;
  p.gA = p.gB = 0; p.sgA = p.sgB = 0;
  p.cgA = g.cgA; p.cgsA = g.cgsA;
  p.tiles_n2 = g.tiles_n2; p.ntiles = 0; p.split = 1; p.mslice = g.M;
  tn_tile(p, t - G.start[pi], 0, smem);
}

__global__ __launch_bounds__(512) void gemm_tn256_grouped_kernel(TnGroup G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_order(blockIdx.x, G.total);
  int pi = 0;
  while (pi + 1 < G.count && t >= G.start[pi + 1]) ++pi;
  const TnG& g = G.g[pi];
  TnP p;
  p.A = g.A; p.B = g.B; p.C = g.C; p.R = nullptr; p.ws = nullptr;
  p.M = g.M; p.N1 = g.N1; p.N2 = g.N2; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldr = 0;
  p.sA = p.sB = p.sC = p.sR = 0;
  p.alpha = g.alpha;
  p.gA = p.gB = 0; p.sgA = p.sgB = 0;
  p.cgA = g.cgA; p.cgsA = g.cgsA;
  p.tiles_n2 = g.tiles_n2; p.ntiles = 0; p.split = 1; p.mslice = g.M;
  tn256_tile(p, t - G.start[pi], 0, smem);
}

// Which tile: the 256 x 256 kernel for products whose output holds whole big tiles and whose contraction is long enough to
// amortise its prologue - when the launch (a batch, or a whole group) brings at least half a round of such tiles: one 256-tile
// is ~130 us of K loop at 88 K-tiles, a launch of 20 of them leaves the chip to the tile's latency (tools/bench_tn.py: 5632 x
// 1152 x 896 alone 86 us on 128-tiles, 142 us on 256-tiles; the four-layer group 848 -> 705 us).  VLA_TN_TILE=128 / 256 forces
// one (read per launch for the A/B tools).
constexpr int TN256_MIN_TILES = 128;
int tn_forced() {
  const char* e = getenv("VLA_TN_TILE");
  return e != nullptr ? atoi(e) : 0;
}
bool tn_eligible_256(int M, int N1, int N2, int split) { return split <= 1 && M >= 1024 && N1 >= 192 && N2 >= 192; }
long long tn_tiles_256(int N1, int N2) { return (long long)((N1 + 255) / 256) * ((N2 + 255) / 256); }
bool tn_use_256(int M, int N1, int N2, int split, int batch) {
  if (split > 1) return false;
  const int f = tn_forced();
  if (f == 128) return false;
  if (f == 256) return true;
  return tn_eligible_256(M, N1, N2, split) && tn_tiles_256(N1, N2) * batch >= TN256_MIN_TILES;
}

// second pass of the contraction split: C = bf16(bf16(alpha * sum_s ws[b][s]) + R); 4 columns per thread
__global__ void tn_finalize_kernel(const float* __restrict__ ws, const bf16_t* __restrict__ R, bf16_t* __restrict__ C, int N1, int N2,
                                   int ldc, int ldr, long long sC, long long sR, float alpha, int split) {
  const long long plane = (long long)N1 * N2, total = plane / 4;
  const int b = blockIdx.y;
  const float* w = ws + (long long)b * split * plane;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i * 4 / N2), c = (int)(i * 4 - (long long)r * N2);
    f32x4 a = *reinterpret_cast<const f32x4*>(w + i * 4);
    for (int s = 1; s < split; ++s) a += *reinterpret_cast<const f32x4*>(w + s * plane + i * 4);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = a[j] * alpha;
    if (R) {
      const uint2 rv = *reinterpret_cast<const uint2*>(R + b * sR + (long long)r * ldr + c);
      v[0] = rbf(v[0]) + bf2f((bf16_t)(rv.x & 0xffff)); v[1] = rbf(v[1]) + bf2f((bf16_t)(rv.x >> 16));
      v[2] = rbf(v[2]) + bf2f((bf16_t)(rv.y & 0xffff)); v[3] = rbf(v[3]) + bf2f((bf16_t)(rv.y >> 16));
    }
    *reinterpret_cast<uint2*>(C + b * sC + (long long)r * ldc + c) = uint2{pack2(v[0], v[1]), pack2(v[2], v[3])};
  }
}

}  // namespace

extern "C" int vla_gemm_bf16_tn(void* stream, const vla_gemm_tn_desc* d) {
  VLA_REQUIRE(d && d->A && d->B && d->C, "gemm_tn: null operand");
  VLA_REQUIRE(d->M > 0 && d->N1 > 0 && d->N2 > 0 && d->batch > 0, "gemm_tn: empty problem");
  VLA_REQUIRE(d->N1 % 8 == 0 && d->N2 % 8 == 0, "gemm_tn: N1 and N2 must be multiples of 8 (16-B column chunks)");
  VLA_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "gemm_tn: lda / ldb must be multiples of 8 elements, ldc of 4");
  VLA_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0 && ((uintptr_t)d->C & 7) == 0, "gemm_tn: A / B must be 16-B, C 8-B aligned");
  VLA_REQUIRE(d->sA % 8 == 0 && d->sB % 8 == 0 && d->sC % 4 == 0, "gemm_tn: batch strides must keep the alignment");
  VLA_REQUIRE(d->a_group >= 0 && d->b_group >= 0 && d->a_group % 64 == 0 && d->b_group % 64 == 0 && d->a_group_stride % 8 == 0 &&
                  d->b_group_stride % 8 == 0,
              "gemm_tn: contraction-row groups must be multiples of 64 rows (a K-tile never straddles a group) with 16-B aligned strides");
  VLA_REQUIRE(d->a_col_group >= 0 && d->a_col_group % 8 == 0 && d->a_col_group_stride % 8 == 0 &&
                  (d->a_col_group == 0 || d->a_col_group_stride >= d->a_col_group),
              "gemm_tn: column groups on A must be multiples of 8 columns");
  if (d->R) VLA_REQUIRE(((uintptr_t)d->R & 7) == 0 && d->ldr % 4 == 0 && d->sR % 4 == 0, "gemm_tn: R must be 8-B aligned");
  const int split = d->split > 1 ? d->split : 1;
  if (split > 1)
    VLA_REQUIRE(d->ws && ((uintptr_t)d->ws & 15) == 0 && d->N2 % 4 == 0, "gemm_tn: split needs an fp32 workspace [batch, split, N1, N2]");
  TnP p;
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = (bf16_t*)d->C; p.R = (const bf16_t*)d->R;
  p.ws = split > 1 ? d->ws : nullptr;
  p.M = d->M; p.N1 = d->N1; p.N2 = d->N2; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.sA = d->sA; p.sB = d->sB; p.sC = d->sC; p.sR = d->sR;
  p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  p.gA = d->a_group; p.gB = d->b_group; p.sgA = d->a_group_stride; p.sgB = d->b_group_stride;
  p.cgA = d->a_col_group; p.cgsA = d->a_col_group_stride;
  p.tiles_n2 = (d->N2 + 127) / 128;
  p.ntiles = ((d->N1 + 127) / 128) * p.tiles_n2;
  p.split = split;
  p.mslice = split > 1 ? ((d->M + split - 1) / split + TBK - 1) / TBK * TBK : d->M;       // K-tile aligned slices (groups stay intact)
  VLA_REQUIRE((long long)p.mslice * (split - 1) < d->M, "gemm_tn: split leaves an empty contraction slice (lower it)");
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
    attr_set = true;
  }
  if (tn_use_256(d->M, d->N1, d->N2, split, d->batch)) {
    p.tiles_n2 = (d->N2 + 255) / 256;
    p.ntiles = ((d->N1 + 255) / 256) * p.tiles_n2;
    hipLaunchKernelGGL(gemm_tn256_kernel, dim3(p.ntiles, 1, d->batch), dim3(512), T2_LDS, st, p);
    VLA_CHECK_LAUNCH("gemm_bf16_tn (256)");
    return VLA_OK;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.ntiles, 1, d->batch * split), dim3(512), LDS_BYTES, st, p);
  VLA_CHECK_LAUNCH("gemm_bf16_tn");
  if (split > 1) {
    const long long total = (long long)d->N1 * d->N2 / 4;
    const unsigned nblk = (unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(tn_finalize_kernel, dim3(nblk, d->batch), dim3(256), 0, st, (const float*)d->ws, (const bf16_t*)d->R, (bf16_t*)d->C,
                       d->N1, d->N2, d->ldc, d->ldr, d->sC, d->sR, p.alpha, split);
    VLA_CHECK_LAUNCH("gemm_tn_finalize");
  }
  return VLA_OK;
}

extern "C" int vla_gemm_bf16_tn_grouped(void* stream, const vla_gemm_tn_desc* descs, int count) {
  VLA_REQUIRE(descs && count > 0 && count <= TN_GROUP_MAX, "gemm_tn_grouped: 1 .. 48 problems per launch");
  TnGroup G[2];                      // [0]: 128-tile kernel, [1]: 256-tile kernel
  int total[2] = {0, 0}, cnt[2] = {0, 0};
  const int forced = tn_forced();
  long long big = 0;                 // 256-tiles of the group's eligible products: below half a round they all stay on the 128-tile kernel
  for (int i = 0; i < count; ++i)
    if (tn_eligible_256(descs[i].M, descs[i].N1, descs[i].N2, descs[i].split)) big += tn_tiles_256(descs[i].N1, descs[i].N2);
  const bool any256 = forced == 256 || (forced != 128 && big >= TN256_MIN_TILES);
  for (int i = 0; i < count; ++i) {
    const vla_gemm_tn_desc* d = descs + i;
    VLA_REQUIRE(d->A && d->B && d->C && d->M > 0 && d->N1 > 0 && d->N2 > 0, "gemm_tn_grouped: null operand / empty problem");
    VLA_REQUIRE(d->batch == 1 && d->split <= 1 && !d->R && d->a_group == 0 && d->b_group == 0,
                "gemm_tn_grouped: plain problems only (batch 1, no split / addend / row groups)");
    VLA_REQUIRE(d->N1 % 8 == 0 && d->N2 % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "gemm_tn_grouped: N1, N2, lda, ldb % 8, ldc % 4");
    VLA_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0 && ((uintptr_t)d->C & 7) == 0, "gemm_tn_grouped: alignment");
    VLA_REQUIRE(d->a_col_group >= 0 && d->a_col_group % 8 == 0 && d->a_col_group_stride % 8 == 0 &&
                    (d->a_col_group == 0 || d->a_col_group_stride >= d->a_col_group), "gemm_tn_grouped: bad column groups");
    const int k = (any256 && (forced == 256 || tn_eligible_256(d->M, d->N1, d->N2, d->split))) ? 1 : 0, tile = k ? 256 : 128;
    TnG& g = G[k].g[cnt[k]];
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B; g.C = (bf16_t*)d->C;
    g.M = d->M; g.N1 = d->N1; g.N2 = d->N2; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.cgA = d->a_col_group; g.cgsA = d->a_col_group_stride;
    g.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    g.tiles_n2 = (d->N2 + tile - 1) / tile;
    G[k].start[cnt[k]] = total[k];
    total[k] += ((d->N1 + tile - 1) / tile) * g.tiles_n2;
    ++cnt[k];
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_tn256_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
    attr_set = true;
  }
  for (int k = 1; k >= 0; --k) {       // the long 256-tile products first: the short ones fill their tail
    if (cnt[k] == 0) continue;
    G[k].start[cnt[k]] = total[k];
    G[k].count = cnt[k]; G[k].total = total[k];
    if (k == 1) hipLaunchKernelGGL(gemm_tn256_grouped_kernel, dim3(total[k]), dim3(512), T2_LDS, (hipStream_t)stream, G[k]);
    else hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total[k]), dim3(512), LDS_BYTES, (hipStream_t)stream, G[k]);
    VLA_CHECK_LAUNCH("gemm_bf16_tn_grouped");
  }
  return VLA_OK;
}
