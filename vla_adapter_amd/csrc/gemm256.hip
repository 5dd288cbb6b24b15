// bf16 NT GEMM, 256 x 256 x 64 tile, for the large products of the step (LLM gate/up, down, ViT qkv / fc1 / fc2, the
// head's task K/V, the full-sequence backward):  C[M,N] = epilogue(A[M,K] . B[N,K]^T), fp32 accumulation on
// v_mfma_f32_16x16x32_bf16.  Same contract and epilogue rounding points as gemm.hip (which keeps the 128-row tiles for
// small-M / small-N problems and the fused RoPE epilogues).
//
// Why a second kernel: the 128 x 128 tile moves 32 KB of operands per 2.1 MFLOP (64 FLOP per staged byte) and sits at the
// CU's global->LDS fill rate; this tile stages half the bytes per FLOP and keeps its LDS-DMA in flight
// across barriers instead of relying on a second resident workgroup.
//
// Structure (cdna_hip_programming.md section 5, "The 256^2 8-phase template"):
//   * 8 waves as 2 (M) x 4 (N); wave tile 128 x 64 = four 64 x 32 quadrants Q(mh, nh); one quadrant x K = 64 is a PHASE
//     (16 MFMAs).  Quadrant order per K-tile: Q00, Q01, Q11, Q10 - the A fragments are read twice per K-tile (8 reads each),
//     the B fragments twice (4 reads each), B0 stays in registers for Q10: 24 ds_read_b128 per 64 MFMAs.
//   * operands live in LDS as eight 16-KiB HALF-TILES (2 K-tile buffers x {B0, A0, B1, A1}); A-half h holds rows
//     {wr*128 + h*64 + [0,64)} of the tile, B-half h columns {wc*64 + h*32 + [0,32)}: every wave needs exactly one A-half
//     and one B-half per quadrant.  Half-tiles are filled by global_load_lds_dwordx4 (2 per wave per half-tile) with the
//     16-B-chunk XOR swizzle on the SOURCE address and on the fragment read.
//   * phase p issues half-tile p + 7 (seven half-tiles = almost two K-tiles ahead); ONE counted wait per K-tile
//     (s_waitcnt vmcnt(6) in the Q10 phase: three half-tiles stay in flight), never vmcnt(0) before the last two K-tiles.
//   * every phase is two segments separated by raw s_barriers: {fragment reads + DMA issue} | {16 MFMAs}.  The wr = 1 waves
//     run ONE SEGMENT behind the wr = 0 waves (one extra barrier at the start), so on every SIMD one wave multiplies while
//     its partner (waves w and w + 4 share a SIMD) reads LDS and issues DMA.
//   * hazards, by barrier count (both wave groups take part in every barrier):
//       RAW  a half-tile issued in phase p is retired by the counted wait of the next Q10 phase P >= p + 3 (in both groups, the
//            later one a segment later) and read from phase P + 1 on;
//       WAR  slot of half-tile h is refilled in phase h + 1: B0 is last read in phase h (its 4 reads are retired by the
//            lgkmcnt(8) BEFORE that phase's first barrier), A0/B1/A1 are refilled >= 2 phases after their last read.
//   * epilogue through the (now free) operand LDS: per wave two 64 x 64 passes, 16-B stores of whole row segments.
#include <type_traits>
#include "gemm_params.h"
#include "../../include/vla_native.h"

namespace {

constexpr int BK = 64;
constexpr int HT = 16384;            // bytes per half-tile (128 rows x 64 k x 2 B)
constexpr int LDS_BYTES = 8 * HT;    // 128 KiB: one workgroup per CU
constexpr int EPI_STRIDE = 64 * 2 + 16;   // staged epilogue row: 64 bf16 + 16 B pad

#define VLA_BARRIER()                      \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // XCD-aware bijective remap + group-M order (same scheme as gemm.hip)
  const int nwg = p.ntiles, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int GM = p.gm > 0 ? p.gm : 4;
  const int tiles_m = p.ntiles / p.tiles_n, per_group = GM * p.tiles_n;
  const int grp = swz / per_group, rem = swz - grp * per_group;
  const int gmr = min(GM, tiles_m - grp * GM);
  const int bm = grp * GM + rem % gmr, bn = rem / gmr;
  const int m0 = bm * 256, n0 = bn * 256;
  const int z = blockIdx.z;
  const bf16_t* Ab = p.A + (long long)z * p.sA;
  const bf16_t* Bb = p.B + (long long)z * p.sB;

  // ---- staging sources: this wave fills rows [16 wid, 16 wid + 16) of every half-tile (2 pieces of 8 rows x 128 B)
  const int kc = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;   // LDS chunk lane&7 of row r holds global chunk (lane&7) ^ (r&7)
  const int lrow = lane >> 3;
  const bf16_t* pa[2][2];
  const bf16_t* pb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ra = min(m0 + wr * 128 + h * 64 + (wid & 3) * 16 + j * 8 + lrow, p.M - 1);
      pa[h][j] = Ab + (p.gA > 0 ? (long long)(ra / p.gA) * p.sgA + (long long)(ra % p.gA) * p.lda : (long long)ra * p.lda) + kc;
      const int rb = min(n0 + (wid >> 1) * 64 + h * 32 + (wid & 1) * 16 + j * 8 + lrow, p.N - 1);
      pb[h][j] = Bb + (long long)rb * p.ldb + kc;
    }
  char* const wdst = smem + wid * 2048;
  // half-tile kinds inside a K-tile buffer: 0 = B0, 1 = A0, 2 = B1, 3 = A1 (the order of first use)
  auto stage_a = [&](int slot, int h, int k0) {
    glds16(pa[h][0] + k0, wdst + slot * HT);
    glds16(pa[h][1] + k0, wdst + slot * HT + 1024);
  };
  auto stage_b = [&](int slot, int h, int k0) {
    glds16(pb[h][0] + k0, wdst + slot * HT);
    glds16(pb[h][1] + k0, wdst + slot * HT + 1024);
  };

  // ---- bias slice of this wave's 64 columns (4 per lane and n tile), fetched ahead of everything else
  const int lq = lane >> 4, lr = lane & 15;
  const int wn0 = n0 + wc * 64;
  uint2 braw[4];
  bool bok[4];
  const bf16_t* bias = (EPI != 2 && p.bias) ? p.bias + (long long)z * p.sBias : nullptr;
  {
    const bool bvec = bias && (((size_t)bias & 7) == 0);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int n = wn0 + t4 * 16 + lq * 4;
      bok[t4] = bvec && n + 3 < p.N;
      braw[t4] = bok[t4] ? *reinterpret_cast<const uint2*>(bias + n) : uint2{0, 0};
    }
  }

#ifdef G256_ABL
#if G256_ABL == 3
  const int nt = 1;                 // diagnostic build: prologue + one K-tile + epilogue only
#else
  const int nt = p.K / BK;
#endif
#else
  const int nt = p.K / BK;
#endif
  // ---- prologue: K-tile 0 (half-tiles 0..3) and the first three half-tiles of K-tile 1
  stage_b(0, 0, 0); stage_a(1, 0, 0); stage_b(2, 1, 0); stage_a(3, 1, 0);
  if (nt > 1) {
    stage_b(4, 0, BK); stage_a(5, 0, BK); stage_b(6, 1, BK);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  VLA_BARRIER();
  if (wr == 1) VLA_BARRIER();      // stagger: the wr = 1 waves run one segment behind

  f32x4 acc[2][2][2][4];           // [mh][nh][ni][mi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row * 128 + ((4 s + lq) ^ (row & 7)) * 16, row & 7 == lane & 7
  const int fo0 = lr * 128 + (((0 + lq) ^ (lane & 7)) << 4);
  const int fo1 = lr * 128 + (((4 + lq) ^ (lane & 7)) << 4);
  const int aoff = wr * 64 * 128, boff = wc * 32 * 128;

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  int d = 0;
  for (int t = 0; t < nt; ++t) {
    const char* kb = smem + d * 4 * HT;
    const int so = d * 4, sn = (d ^ 1) * 4;
    // ================= phase Q00: reads B0 (4, first) + A0 (8); issues A1 of K-tile t+1
    {
      const char* sb = kb + 0 * HT + boff;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        fb0[ni][0] = *reinterpret_cast<const bf16x8*>(sb + ni * 2048 + fo0);
        fb0[ni][1] = *reinterpret_cast<const bf16x8*>(sb + ni * 2048 + fo1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const char* sa = kb + 1 * HT + aoff;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        fa[mi][0] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo0);
        fa[mi][1] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nt) stage_a(sn + 3, 1, (t + 1) * BK);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the B0 reads are done: its slot is refilled next phase
      VLA_BARRIER();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[0][0][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[ni][s], fa[mi][s], acc[0][0][ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VLA_BARRIER();
    }
    // ================= phase Q01: reads B1 (4); issues B0 of K-tile t+2
    {
      const char* sb = kb + 2 * HT + boff;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        fb1[ni][0] = *reinterpret_cast<const bf16x8*>(sb + ni * 2048 + fo0);
        fb1[ni][1] = *reinterpret_cast<const bf16x8*>(sb + ni * 2048 + fo1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nt) stage_b(so + 0, 0, (t + 2) * BK);
      VLA_BARRIER();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[0][1][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[ni][s], fa[mi][s], acc[0][1][ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VLA_BARRIER();
    }
    // ================= phase Q11: reads A1 (8); issues A0 of K-tile t+2
    {
      const char* sa = kb + 3 * HT + aoff;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        fa[mi][0] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo0);
        fa[mi][1] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nt) stage_a(so + 1, 0, (t + 2) * BK);
      VLA_BARRIER();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[1][1][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[ni][s], fa[mi][s], acc[1][1][ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VLA_BARRIER();
    }
    // ================= phase Q10: no reads (B0 kept in registers); issues B1 of K-tile t+2; the K-tile's counted wait
    {
      if (t + 2 < nt) {
        stage_b(so + 2, 1, (t + 2) * BK);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // K-tile t+1 has landed; three half-tiles of t+2 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      VLA_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[1][0][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[ni][s], fa[mi][s], acc[1][0][ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      VLA_BARRIER();
    }
    d ^= 1;
  }
  if (wr == 0) VLA_BARRIER();      // pairs with the last barrier of the wr = 1 waves: every operand read is finished
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#if defined(G256_ABL) && G256_ABL == 1
  {                                 // diagnostic build: no epilogue at all (accumulators kept alive)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(acc[a][b][c][e]));
    return;
  }
#endif
  // ---------------- epilogue: two 64 x 64 passes per wave through its private staging region ----------------
  char* reg = smem + wid * (64 * EPI_STRIDE);
  float bv[4][4];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    if (bok[t4]) {
      bv[t4][0] = bf2f((bf16_t)(braw[t4].x & 0xffff)); bv[t4][1] = bf2f((bf16_t)(braw[t4].x >> 16));
      bv[t4][2] = bf2f((bf16_t)(braw[t4].y & 0xffff)); bv[t4][3] = bf2f((bf16_t)(braw[t4].y >> 16));
    } else {
      const int n = wn0 + t4 * 16 + lq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[t4][j] = (bias && n + j < p.N) ? bf2f(bias[n + j]) : 0.f;
    }
  }

#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
    const int wm0 = m0 + wr * 128 + mh * 64;
    // view of this half as [t4 = 2 nh + ni][mi]
    auto A4 = [&](int t4, int mi) -> f32x4& { return acc[mh][t4 >> 1][t4 & 1][mi]; };

    if (EPI == 2) {
      // SwiGLU backward fused into dH = dY . W_down (accumulator = dH of this 64 x 64 patch): read the matching interleaved
      // pre-activations GU[m, 2N], emit dGU in the same layout; dH itself is never stored.
      const bf16_t* GU = p.R + (long long)z * p.sR;
      bf16_t* Cb = p.C + (long long)z * p.sC;
      constexpr int ROWB = 2 * 64 * 2 + 16;
      char* reg2 = smem + wid * (32 * ROWB);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
          const int mi = 2 * half + mq;
          const int m = min(wm0 + mi * 16 + lr, p.M - 1);
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            const int hc = wn0 + t4 * 16 + lq * 4;
            const long long go = (p.gR > 0 ? (long long)(m / p.gR) * p.sgR + (long long)(m % p.gR) * p.ldr : (long long)m * p.ldr) +
                                 (hc >> 4) * 32 + (hc & 15);
            const bool ok = hc + 3 < p.N;
            const uint2 gv = ok ? *reinterpret_cast<const uint2*>(GU + go) : uint2{0, 0};
            const uint2 uv = ok ? *reinterpret_cast<const uint2*>(GU + go + 16) : uint2{0, 0};
            const float gg[4] = {bf2f((bf16_t)(gv.x & 0xffff)), bf2f((bf16_t)(gv.x >> 16)), bf2f((bf16_t)(gv.y & 0xffff)), bf2f((bf16_t)(gv.y >> 16))};
            const float uu[4] = {bf2f((bf16_t)(uv.x & 0xffff)), bf2f((bf16_t)(uv.x >> 16)), bf2f((bf16_t)(uv.y & 0xffff)), bf2f((bf16_t)(uv.y >> 16))};
            float dg[4], du[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float dd = rbf(A4(t4, mi)[j] * p.alpha);
              const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-gg[j]));
              du[j] = dd * gg[j] * sg;
              dg[j] = dd * uu[j] * (sg * (1.0f + gg[j] * (1.0f - sg)));
            }
            char* rowp = reg2 + (mq * 16 + lr) * ROWB + (t4 * 32 + lq * 4) * 2;
            *reinterpret_cast<uint2*>(rowp) = uint2{pack2(dg[0], dg[1]), pack2(dg[2], dg[3])};
            *reinterpret_cast<uint2*>(rowp + 32) = uint2{pack2(du[0], du[1]), pack2(du[2], du[3])};
          }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {                  // 32 rows x 16 chunks of 16 B: 4 rows per pass
          const int row = it * 4 + (lane >> 4), ch = lane & 15;
          const int m = wm0 + half * 32 + row, n2 = 2 * wn0 + ch * 8;
          if (m < p.M && n2 + 8 <= 2 * p.N)
            *reinterpret_cast<uint4*>(Cb + (long long)m * p.ldc + n2) = *reinterpret_cast<const uint4*>(reg2 + row * ROWB + ch * 16);
        }
      }
      continue;
    }
    if (EPI == 1) {
      // SwiGLU forward: columns interleaved in 16s - even n tiles are gate, odd n tiles the matching up columns.  The product
      // h (64 rows x 32 columns per pass) is staged through LDS like C, so that it leaves as 16-B row segments (direct 8-B
      // stores touched 16 cache lines per instruction: 4x the store instructions on every tile's tail).
      bf16_t* C2 = p.C2 + (long long)z * p.sC2;
      constexpr int HSTR = 32 * 2 + 16;                       // staged h row: 32 bf16 + pad
      char* regh = smem + 8 * (64 * EPI_STRIDE) + wid * (64 * HSTR);
      const bool plain = bias == nullptr && p.alpha == 1.f;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          float h[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (Qwen2's gate / up projections carry no bias and alpha is 1: the wave-uniform `plain` spares two FMAs per pair)
            const float g = rbf(plain ? A4(2 * pr, mi)[j] : A4(2 * pr, mi)[j] * p.alpha + bv[2 * pr][j]);
            const float u = rbf(plain ? A4(2 * pr + 1, mi)[j] : A4(2 * pr + 1, mi)[j] * p.alpha + bv[2 * pr + 1][j]);
            A4(2 * pr, mi)[j] = g;
            A4(2 * pr + 1, mi)[j] = u;
            h[j] = rbf(g * __builtin_amdgcn_rcpf(1.0f + __expf(-g))) * u;
          }
          *reinterpret_cast<uint2*>(regh + (mi * 16 + lr) * HSTR + (pr * 16 + lq * 4) * 2) = uint2{pack2(h[0], h[1]), pack2(h[2], h[3])};
        }
      const bool hvec = ((p.ldc2 & 7) == 0) && (((size_t)C2 & 15) == 0);
#pragma unroll
      for (int it = 0; it < 4; ++it) {                        // 64 rows x 4 chunks of 16 B: 16 rows per pass
        const int row = it * 16 + (lane >> 2), ch = lane & 3;
        const int m = wm0 + row, hc = (wn0 >> 1) + ch * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(regh + row * HSTR + ch * 16);
        if (m >= p.M || hc >= (p.N >> 1)) continue;
        bf16_t* dst = C2 + (long long)m * p.ldc2 + hc;
        if (hvec && hc + 8 <= (p.N >> 1)) {
          *reinterpret_cast<uint4*>(dst) = v;
        } else {
          const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (hc + k < (p.N >> 1)) dst[k] = (bf16_t)((k & 1) ? (wv[k >> 1] >> 16) : (wv[k >> 1] & 0xffffu));
        }
      }
      if (p.C == nullptr) continue;
      // pre-activations kept for a live-row backward only (c_live): a 64-row block without a live row stores nothing
      if (p.c_live_mod > 0 && (wm0 % p.c_live_mod) + 63 < p.c_live_from) continue;
    } else {
      float alpha = p.alpha;
      if (p.bias_post) {                      // bf16(bf16(alpha acc) + bias): torch CPU Linear on a strided input
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) A4(t4, mi)[j] = rbf(A4(t4, mi)[j] * alpha);
        alpha = 1.f;
      }
      auto finish = [&](auto fn) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) A4(t4, mi)[j] = fn(A4(t4, mi)[j] * alpha + bv[t4][j]);
      };
      if (p.act == VLA_ACT_GELU) finish([](float v) { return gelu_erf(rbf(v)); });
      else if (p.act == VLA_ACT_RELU) finish([](float v) { return fmaxf(v, 0.f); });
      else if (p.act == VLA_ACT_GELU_TANH) finish([](float v) { return gelu_tanh(rbf(v)); });
      else finish([](float v) { return v; });
    }

    // stage the 64 x 64 half (bf16) through the wave's private LDS region, then 16-B stores of whole row segments
    bf16_t* Cb = p.C + (long long)z * p.sC;
    const bf16_t* Rb = p.R ? p.R + (long long)z * p.sR : nullptr;
    const bool vec_ok = ((p.ldc & 7) == 0) && (!Rb || (p.ldr & 7) == 0);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        uint2 o = {pack2(A4(t4, mi)[0], A4(t4, mi)[1]), pack2(A4(t4, mi)[2], A4(t4, mi)[3])};
        *reinterpret_cast<uint2*>(reg + (mi * 16 + lr) * EPI_STRIDE + (t4 * 16 + lq * 4) * 2) = o;
      }
    // Row addressing: the plain case (no row groups, no broadcast residual, no live-row filter) must not pay the four integer
    // divisions per output row of the general case - 128 divisions per wave sat on every tile's tail.
    const bool plain_rows = (p.res_mod | p.gR | p.gC | p.c_live_mod) == 0;
    auto store_rows = [&](auto fast_t) {
      constexpr bool FAST = decltype(fast_t)::value;
#pragma unroll
      for (int it = 0; it < 8; ++it) {                       // 64 rows x 8 chunks of 16 B: 8 rows per pass
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int m = wm0 + row, n = wn0 + ch * 8;
        uint4 v = *reinterpret_cast<const uint4*>(reg + row * EPI_STRIDE + ch * 16);
        if (m >= p.M || n >= p.N) continue;
        long long roff, crow;
        if constexpr (FAST) {
          roff = (long long)m * p.ldr;
          crow = (long long)m * p.ldc;
        } else {
          if (p.c_live_mod > 0 && (m % p.c_live_mod) < p.c_live_from) continue;
          roff = p.res_mod > 0 ? (long long)(m % p.res_mod) * p.ldr
                 : p.gR > 0 ? (long long)(m / p.gR) * p.sgR + (long long)(m % p.gR) * p.ldr : (long long)m * p.ldr;
          crow = p.gC > 0 ? (long long)(m / p.gC) * p.sgC + (long long)(m % p.gC) * p.ldc : (long long)m * p.ldc;
        }
        if (vec_ok && n + 8 <= p.N) {
          if (Rb) {
            const uint4 rv = *reinterpret_cast<const uint4*>(Rb + roff + n);
            const unsigned a[4] = {v.x, v.y, v.z, v.w};
            const unsigned b[4] = {rv.x, rv.y, rv.z, rv.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              o[k] = pack2(bf2f((bf16_t)(a[k] & 0xffff)) + bf2f((bf16_t)(b[k] & 0xffff)),
                           bf2f((bf16_t)(a[k] >> 16)) + bf2f((bf16_t)(b[k] >> 16)));
            v = uint4{o[0], o[1], o[2], o[3]};
          }
          *reinterpret_cast<uint4*>(Cb + crow + n) = v;
        } else {
          const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (n + k < p.N) {
              float f = bf2f((bf16_t)((k & 1) ? (wv[k >> 1] >> 16) : (wv[k >> 1] & 0xffffu)));
              if (Rb) f += bf2f(Rb[roff + n + k]);
              Cb[crow + n + k] = f2bf(f);
            }
          }
        }
      }
    };
    if (plain_rows) store_rows(std::true_type{});
    else store_rows(std::false_type{});
  }
}

template <int EPI>
int launch256(const GemmP& p0, int batch, hipStream_t st) {
  GemmP p = p0;
  p.tiles_n = (p.N + 255) / 256;
  p.ntiles = ((p.M + 255) / 256) * p.tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<EPI>), dim3(p.ntiles, 1, batch), dim3(512), LDS_BYTES, st, p);
  return 0;
}

}  // namespace

int vla_gemm256_launch(const GemmP& p, int epi, int batch, hipStream_t st) {
  if (epi == 1) return launch256<1>(p, batch, st);
  if (epi == 2) return launch256<2>(p, batch, st);
  return launch256<0>(p, batch, st);
}
