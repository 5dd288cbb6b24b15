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
    attr_set = true;
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
  TnGroup G;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    const vla_gemm_tn_desc* d = descs + i;
    VLA_REQUIRE(d->A && d->B && d->C && d->M > 0 && d->N1 > 0 && d->N2 > 0, "gemm_tn_grouped: null operand / empty problem");
    VLA_REQUIRE(d->batch == 1 && d->split <= 1 && !d->R && d->a_group == 0 && d->b_group == 0,
                "gemm_tn_grouped: plain problems only (batch 1, no split / addend / row groups)");
    VLA_REQUIRE(d->N1 % 8 == 0 && d->N2 % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "gemm_tn_grouped: N1, N2, lda, ldb % 8, ldc % 4");
    VLA_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0 && ((uintptr_t)d->C & 7) == 0, "gemm_tn_grouped: alignment");
    VLA_REQUIRE(d->a_col_group >= 0 && d->a_col_group % 8 == 0 && d->a_col_group_stride % 8 == 0 &&
                    (d->a_col_group == 0 || d->a_col_group_stride >= d->a_col_group), "gemm_tn_grouped: bad column groups");
    TnG& g = G.g[i];
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B; g.C = (bf16_t*)d->C;
    g.M = d->M; g.N1 = d->N1; g.N2 = d->N2; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.cgA = d->a_col_group; g.cgsA = d->a_col_group_stride;
    g.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    g.tiles_n2 = (d->N2 + 127) / 128;
    G.start[i] = total;
    total += ((d->N1 + 127) / 128) * g.tiles_n2;
  }
  G.start[count] = total;
  G.count = count; G.total = total;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total), dim3(512), LDS_BYTES, (hipStream_t)stream, G);
  VLA_CHECK_LAUNCH("gemm_bf16_tn_grouped");
  return VLA_OK;
}
