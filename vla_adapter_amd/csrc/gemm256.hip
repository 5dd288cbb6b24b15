// bf16 NT GEMM, 256 x 256 x 64 tile, for the large products of the step (LLM gate/up, down, ViT qkv / fc1 / fc2, the
// head's task K/V, the full-sequence backward):  C[M,N] = epilogue(A[M,K] . B[N,K]^T), fp32 accumulation on
// v_mfma_f32_16x16x32_bf16.  Same contract and epilogue rounding points as gemm.hip (which keeps the 128-row tiles for
// small-M / small-N problems, the SwiGLU-backward epilogue and the interleaved RoPE epilogue).
//
// Why a second kernel: the 128 x 128 tile moves 32 KB of operands per 2.1 MFLOP (64 FLOP per staged byte) and sits at the
// CU's global->LDS fill rate; this tile stages half the bytes per FLOP and keeps its LDS-DMA in flight
// across barriers instead of relying on a second resident workgroup.
//
// Structure (after cdna_hip_programming.md section 5, "The 256^2 8-phase template", re-derived in round 3 as TWO phases per K-tile):
//   * 8 waves as 2 (M) x 4 (N); wave tile 128 x 64 = four 64 x 32 quadrants Q(mh, nh), 16 MFMAs each per K-tile.
//   * operands live in LDS as eight 16-KiB HALF-TILES (2 K-tile buffers x {B0, A0, B1, A1}); A-half h holds rows
//     {wr*128 + h*64 + [0,64)} of the tile, B-half h columns {wc*64 + h*32 + [0,32)}.  Half-tiles are filled by
//     global_load_lds_dwordx4 (2 per wave per half-tile) with the 16-B-chunk XOR swizzle on the SOURCE address and on the
//     fragment read.
//   * a K-tile is two PHASES, each two SEGMENTS separated by raw s_barriers: {fragment reads + DMA issue + counted wait} |
//     {32 MFMAs}.  Phase X reads B0, A0, B1 (16 ds_read_b128) and multiplies Q00, Q01; phase Y reads A1 (8) and multiplies Q11,
//     Q10 (B0 / B1 stay in registers).  The wr = 1 waves run ONE SEGMENT behind the wr = 0 waves (one extra barrier at the
//     start), so on every SIMD one wave multiplies while its partner (waves w and w + 4 share a SIMD) reads LDS and issues DMA.
//     The template's four phases of 16 MFMAs (eight barriers per K-tile) measured 5 - 12 % slower on every shape of the step
//     (tools/bench_gemm256.py, round 3): in-kernel stamps put this K loop at 2 090 - 2 130 cycles per K-tile against 2 048 for
//     its 128 MFMAs per SIMD - what is left between it and the vendor kernel is the clock the chip holds under it, not cycles.
//   * hazards, by segment count (segments of the wr = 0 waves: 4t X-read, 4t+1 X-mma, 4t+2 Y-read, 4t+3 Y-mma; wr = 1 one later;
//     every read segment ends with lgkmcnt(0) BEFORE its closing barrier):
//       WAR  B0/A0/B1 of K-tile t: last read in segment 4t+1, refilled (K-tile t+2) in Y-read(t) = 4t+2 / 4t+3;
//            A1 of K-tile t: last read in 4t+3, refilled (K-tile t+2) in X-read(t+1) = 4t+4 / 4t+5;
//       RAW  B0/A0/B1 of K-tile t+1 (issued in Y-read(t-1) / at the tile top) are retired by the counted wait of Y-read(t)
//            (segments 4t+2, 4t+3; first read in 4t+4); A1 of K-tile t+1 (issued in X-read(t)) by the wait of X-read(t+1)
//            (4t+4, 4t+5; first read in 4t+6).  Issue -> wait distance: four segments for every half-tile; never vmcnt(0)
//            before the last two K-tiles.
//   * epilogue through the (now free) operand LDS: per wave two 64 x 64 passes, 16-B stores of whole row segments; a pass requests
//     its eight staged segments before the first store waits for one (eight serial LDS round trips per pass were the epilogue's
//     largest single item on the stamped build: tools/diag/g256_stamps.py).  Residual segments are requested only by the
//     instantiation that adds one (template flag RES).
//   * PERSISTENT over tiles: at most one workgroup per CU walks the tile list.  K-tile 0 of the NEXT tile is issued (LDS-DMA
//     into the K-tile buffer the epilogue does not stage through) before the epilogue of the current one, so the pipeline
//     fill of every tile but a workgroup's first hides behind an epilogue; the epilogue's stores drain behind the next fill.
//     The walk is static (workgroup b takes the tiles of the virtual workgroups b, b + G, b + 2G ...): handing tiles out by
//     atomic tickets fetched a tile ahead measured 0.5 - 1 % SLOWER on the whole step (26.03-26.14 vs 25.82-25.86 ms, one
//     workgroup per tile 25.90-26.02) and needed device-side state; it is gone.
#include <type_traits>
#include "gemm_params.h"
#include "../../include/vla_native.h"

namespace {

constexpr int BK = 64;
constexpr int HT = 16384;            // bytes per half-tile (128 rows x 64 k x 2 B)
constexpr int LDS_BYTES = 8 * HT;    // 128 KiB of operands: one workgroup per CU
constexpr int STG = 8192;            // epilogue staging per wave: 64 rows x 128 B, 16-B chunk c of row r at c ^ ((r >> 1) & 7)

// LDS-DMA issued from inline asm: 16 B per lane from (wave-uniform base + per-lane 32-bit byte offset) to LDS address `dst`
// (+ lane * 16).  Hidden from hipcc on purpose: beside a builtin global_load_lds it drains vmcnt(0) before every ordinary
// load, every ds_write that might alias the DMA target and every reuse of a loaded register - i.e. all through an epilogue
// that runs under the next tile's K-tile 0.  Its completion is counted by hand (the s_waitcnt statements below); M0 is
// saved and restored in the same statement (it is compiler-reserved).
__device__ __forceinline__ void glds16s(const char* base, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// One quadrant x one K-tile, written out IN PLACE in each phase (a lambda here let the compiler sink MFMAs below the phase's
// closing barrier - the machine scheduler's region no longer ended at the sched_barrier - and cost 0.9 ms on the step):
// 16 bf16 MFMAs (two 32-deep k-steps) or 8 fp8 MFMAs (one 128-deep step on the concatenated 32-byte fragments).
#define VLA_MMA_QUADRANT(Q, FB, FA)                                                                                                   \
  do {                                                                                                                                \
    if constexpr (F8) {                                                                                                               \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                                                \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                              \
          Q[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(FB[ni][0], FB[ni][1]), cat8(FA[mi][0], FA[mi][1]), Q[ni][mi], \
                                                                       0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                            \
    } else {                                                                                                                          \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                                   \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                                              \
          _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                            \
            Q[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[ni][s], FA[mi][s], Q[ni][mi], 0, 0, 0);                            \
    }                                                                                                                                 \
  } while (0)

typedef int v8i_f8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8i_f8 cat8(bf16x8 lo, bf16x8 hi) {          // two 16-B fragment reads -> the 32-byte fp8 operand
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x4 l = __builtin_bit_cast(f32x4, lo), h = __builtin_bit_cast(f32x4, hi);
  return __builtin_bit_cast(v8i_f8, f32x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]});
}

#define VLA_BARRIER()                      \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

// F8: OCP e4m3 operands with per-row fp32 scales (gemm.hip's fp8 form, same conventions): a K-tile stays 128 B per row = 128
// elements = ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile instead of two bf16 MFMAs - identical staging, LDS image and
// barrier structure, half the K-tiles per product.  The scales are applied to the accumulators at the start of the epilogue.
// EXT (round 4): K extension - the contraction continues over a second bf16 operand pair (A2 [M, K2], B2 [N, K2]: gemm.hip's EXT, the
// LoRA-wrapped Linear as one product); the K-tiles behind K / 64 are staged from (A2, B2) through four more per-lane offsets each, the
// loop, its counted waits and the epilogues are unchanged.  Separate instantiations: the plain kernels keep their register count.
template <int EPI, bool F8 = false, bool RES = true, bool R2 = false, bool EXT = false>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EB = F8 ? 1 : 2;      // bytes per operand element
  constexpr bool HAS_RES = EPI == 0 && RES;   // residual segments are requested (always all sixteen: see below) only by launches that add one
  constexpr bool PRE = EPI != 2;      // next tile's K-tile 0 in flight during the epilogue (SwiGLU backward stages wider rows)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // ---- the tile list: `total` tiles (all batches) in eight contiguous XCD lists (bijective split, as gemm.hip); virtual
  //      workgroup v owns entry v >> 3 of list v & 7, and workgroup b of a grid of G walks v = b, b + G, b + 2G ...
  const int G = gridDim.x, bid = blockIdx.x;
  const int total = p.ntiles * p.batch;
  const int q8 = total >> 3, r8 = total & 7;
  const int xcd = bid & 7;
  auto rstart = [&](int x) { return x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8; };
  int vt = bid;
  // ---- tile identity + staging sources: this wave fills rows [16 wid, 16 wid + 16) of every half-tile (2 pieces of 8 rows)
  // group-M width of the tile order (A row panels an XCD keeps L2-resident while B tiles stream): 6, as for the 128-row tiles -
  // on the one-pipeline step 25.84-25.87 ms against 25.93-26.09 for 4 and 25.95-26.02 for 8 (same box, three alternating runs)
  constexpr int GM = 6;
  int m0, n0, z;
  // sources as wave-uniform bases + 32-bit per-lane byte offsets (half the address registers of eight pointers)
  const char* Ab; const char* Bb;
  unsigned oa[2][2], ob[2][2];
  unsigned oa2[EXT ? 2 : 1][2], ob2[EXT ? 2 : 1][2];      // EXT: the same rows of A2 / B2 (their own row strides)
  auto setup = [&](int swz) {
    int sl = lane;
    asm volatile("" : "+v"(sl));       // per-tile address arithmetic stays here (hoisted out of the tile loop it costs registers
    const int kc = ((sl & 7) ^ ((sl >> 3) & 7)) * 16;  // (bytes) across the K loop); LDS chunk lane&7 of row r holds global chunk (lane&7)^(r&7)
    const int lrow = sl >> 3;
    int gA = p.gA;
    asm volatile("" : "+s"(gA));       // (the same for the reciprocal of a divisor: recomputed per tile, not carried in VGPRs)
    int ntl = p.ntiles;
    asm volatile("" : "+s"(ntl));
    z = swz / ntl;
    const int tl = swz - z * ntl;
    // group-M order (an XCD-blocked order - every XCD on a compact block of the tile grid - cut the fabric fetch of fc2 by 25 %
    // and changed no launch duration by more than 1 %: profiles/r03_gemm256_traffic_order_ab.json; tools/diag/gemm256_pruned_paths.patch)
    int br = ntl / p.tiles_n, bc = p.tiles_n;
    asm volatile("" : "+s"(br), "+s"(bc));
    const int pgb = GM * bc;
    const int grp = tl / pgb, rem = tl - grp * pgb;
    const int gmr = min(GM, br - grp * GM);
    m0 = (grp * GM + rem % gmr) * 256;
    n0 = (rem / gmr) * 256;
    Ab = reinterpret_cast<const char*>(p.A) + (long long)z * p.sA * EB;
    Bb = reinterpret_cast<const char*>(p.B) + (long long)z * p.sB * EB;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ra = min(m0 + wr * 128 + h * 64 + (wid & 3) * 16 + j * 8 + lrow, p.M - 1);
        oa[h][j] = (unsigned)((gA > 0 ? (long long)(ra / gA) * p.sgA + (long long)(ra % gA) * p.lda : (long long)ra * p.lda) * EB + kc);
        const int rb = min(n0 + (wid >> 1) * 64 + h * 32 + (wid & 1) * 16 + j * 8 + lrow, p.N - 1);
        ob[h][j] = (unsigned)((long long)rb * p.ldb * EB + kc);
        if constexpr (EXT) {
          oa2[h][j] = (unsigned)((long long)ra * p.lda2 * 2 + kc);
          ob2[h][j] = (unsigned)((long long)rb * p.ldb2 * 2 + kc);
        }
      }
  };
  const unsigned wdst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + wid * 2048;
  // half-tile kinds inside a K-tile buffer: 0 = B0, 1 = A0, 2 = B1, 3 = A1 (the order of first use)
  auto stage_a = [&](int slot, int h, int k0) {
    if (EXT && k0 >= p.K) {          // (wave-uniform: K-tiles of the extension)
      const char* b2 = reinterpret_cast<const char*>(p.A2) + (k0 - p.K) * 2;
      glds16s(b2, oa2[EXT ? h : 0][0], wdst + slot * HT);
      glds16s(b2, oa2[EXT ? h : 0][1], wdst + slot * HT + 1024);
      return;
    }
    glds16s(Ab + k0 * 2, oa[h][0], wdst + slot * HT);
    glds16s(Ab + k0 * 2, oa[h][1], wdst + slot * HT + 1024);
  };
  auto stage_b = [&](int slot, int h, int k0) {
    if (EXT && k0 >= p.K) {
      const char* b2 = reinterpret_cast<const char*>(p.B2) + (k0 - p.K) * 2;
      glds16s(b2, ob2[EXT ? h : 0][0], wdst + slot * HT);
      glds16s(b2, ob2[EXT ? h : 0][1], wdst + slot * HT + 1024);
      return;
    }
    glds16s(Bb + k0 * 2, ob[h][0], wdst + slot * HT);
    glds16s(Bb + k0 * 2, ob[h][1], wdst + slot * HT + 1024);
  };
  auto stage_k0 = [&](int buf) {
    stage_b(buf * 4 + 0, 0, 0); stage_a(buf * 4 + 1, 0, 0); stage_b(buf * 4 + 2, 1, 0); stage_a(buf * 4 + 3, 1, 0);
  };
  // bias slice of this wave's 64 columns (4 per lane and n tile): requested at the top of the tile BEHIND K-tile 1, always as
  // exactly four loads (the top's counted wait leaves them in flight; its count is hand-written)
  uint2 braw[4];
  auto bias_vec = [&]() {
    const bf16_t* bias = (EPI != 2 && p.bias) ? p.bias + (long long)z * p.sBias : nullptr;
    return bias != nullptr && (((size_t)bias & 7) == 0) && p.N >= 4;
  };
  auto fetch_bias = [&](bool bvec) {
    int sl = lane;
    asm volatile("" : "+v"(sl));
    const int lq = sl >> 4;
    // always four loads, in one block (a conditional load would reach `braw` through a copy, and the copy waits for it): without
    // a usable bias they read the head of B and are ignored
    // (as an opaque byte distance from B, so that the address stays a global one and no copy of the loads is specialised per case)
    long long dist = bvec ? (long long)((uintptr_t)(p.bias + (long long)z * p.sBias) - (uintptr_t)p.B) : 0;
    int lim = bvec ? p.N : 0;
    asm volatile("" : "+s"(dist), "+s"(lim));
    const bf16_t* src = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(p.B) + dist);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int n = n0 + wc * 64 + t4 * 16 + lq * 4;
      braw[t4] = *reinterpret_cast<const uint2*>(src + (n + 3 < lim ? n : 0));
    }
  };

  const int nt = p.K / (F8 ? 2 * BK : BK) + (EXT ? p.K2 / BK : 0);      // K-tiles of 128 B per row (EXT: those of the extension behind)

  const int aoff = wr * 64 * 128, boff = wc * 32 * 128;

  int cur = rstart(xcd) + (bid >> 3);
  // Free de-phasing: when the last round of the walk is partial, the workgroups that walk one tile fewer would finish a tile early;
  // they start late by a fraction of a tile instead, so that their epilogues' store bursts fall between those of the others.
  if (p.stagger > 0 && (total - 1 - bid) / G + 1 < (total + G - 1) / G) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);       // ~1024 cycles each
  }
  setup(cur);
  stage_k0(0);
  int d = 0;

  for (;;) {
    // ================= top of a tile: its K-tile 0 is in flight in buffer d (behind it, the previous epilogue's stores)
    vt += G;                           // the walk: virtual workgroup ids bid, bid + G, ... (same tile map as one workgroup per tile)
    const int nxt = vt < total ? rstart(vt & 7) + (vt >> 3) : -1;
    // Nothing but the walk position and the lane index is carried across an epilogue: the sources of this tile are derived again
    // (its K-tile 0 was issued from the same values before the previous epilogue), the bias slice is requested now.
    asm volatile("" : "+s"(cur));
    setup(cur);
    const bool bvec = bias_vec();
    if (nt > 1) {
      const int sn = (d ^ 1) * 4;
      stage_b(sn + 0, 0, BK); stage_a(sn + 1, 0, BK); stage_b(sn + 2, 1, BK);
    }
    fetch_bias(bvec);
    if (nt > 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // K-tile 0 is in; K-tile 1's three half-tiles and the bias stay in flight
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    VLA_BARRIER();
    // fragment read offsets: row * 128 + ((4 s + lq) ^ (row & 7)) * 16, row & 7 == lane & 7
    int fl = lane;
    asm volatile("" : "+v"(fl));
    // (fp8: the lane's 32 contiguous bytes of the row = chunks 2 lq and 2 lq + 1)
    const int fo0 = (fl & 15) * 128 + ((((F8 ? 2 * (fl >> 4) : 0 + (fl >> 4))) ^ (fl & 7)) << 4);
    const int fo1 = (fl & 15) * 128 + ((((F8 ? 2 * (fl >> 4) + 1 : 4 + (fl >> 4))) ^ (fl & 7)) << 4);
    if (wr == 1) VLA_BARRIER();      // stagger: the wr = 1 waves run one segment behind

    f32x4 acc[2][2][2][4];           // [mh][nh][ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][c][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    for (int t = 0; t < nt; ++t) {
      const char* kb = smem + d * 4 * HT;
      const int so = d * 4, sn = (d ^ 1) * 4;
      // ================= phase X: reads B0 (4), A0 (8), B1 (4); issues A1 of K-tile t+1; retires A1 of K-tile t
      {
        const char* sb = kb + 0 * HT + boff;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          fb0[ni][0] = *reinterpret_cast<const bf16x8*>(sb + ni * 2048 + fo0);
          fb0[ni][1] = *reinterpret_cast<const bf16x8*>(sb + ni * 2048 + fo1);
        }
        const char* sa = kb + 1 * HT + aoff;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          fa[mi][0] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo0);
          fa[mi][1] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo1);
        }
        const char* sb1 = kb + 2 * HT + boff;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          fb1[ni][0] = *reinterpret_cast<const bf16x8*>(sb1 + ni * 2048 + fo0);
          fb1[ni][1] = *reinterpret_cast<const bf16x8*>(sb1 + ni * 2048 + fo1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) {
          stage_a(sn + 3, 1, (t + 1) * BK);
          if (t > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // A1 of K-tile t is in; B0/A0/B1 + A1 of t+1 in flight
        } else if (t > 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VLA_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        VLA_MMA_QUADRANT(acc[0][0], fb0, fa);
        VLA_MMA_QUADRANT(acc[0][1], fb1, fa);
        __builtin_amdgcn_s_setprio(0);
        VLA_BARRIER();
      }
      // ================= phase Y: reads A1 (8); issues B0, A0, B1 of K-tile t+2; retires B0, A0, B1 of K-tile t+1
      {
        const char* sa = kb + 3 * HT + aoff;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          fa[mi][0] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo0);
          fa[mi][1] = *reinterpret_cast<const bf16x8*>(sa + mi * 2048 + fo1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) {
          stage_b(so + 0, 0, (t + 2) * BK); stage_a(so + 1, 0, (t + 2) * BK); stage_b(so + 2, 1, (t + 2) * BK);
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // (t = 0: the four bias loads sit between K-tile 1 and these - retired too)
        } else if (t + 1 < nt) {
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // A1 of the last K-tile stays in flight
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VLA_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        VLA_MMA_QUADRANT(acc[1][1], fb1, fa);
        VLA_MMA_QUADRANT(acc[1][0], fb0, fa);
        __builtin_amdgcn_s_setprio(0);
        VLA_BARRIER();
      }
      d ^= 1;
    }
    if (wr == 0) VLA_BARRIER();      // pairs with the last barrier of the wr = 1 waves: every operand read is finished
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---------------- epilogue of tile (em0, en0, ez): two 64 x 64 passes per wave through its private staging region,
    //                  in the K-tile buffer that the next tile's K-tile 0 (issued first) does not occupy
    const int em0 = m0, en0 = n0, ez = z;
    int el = lane;
    asm volatile("" : "+v"(el));       // (see setup: nothing of the epilogue's addressing may live across the K loop)
    const int lq = el >> 4, lr = el & 15;
    int gR = p.gR, gC = p.gC, res_mod = p.res_mod, c_live_mod = p.c_live_mod;
    asm volatile("" : "+s"(gR), "+s"(gC), "+s"(res_mod), "+s"(c_live_mod));
    const bf16_t* bias = (EPI != 2 && p.bias) ? p.bias + (long long)ez * p.sBias : nullptr;
    const int wn0 = en0 + wc * 64;
    if constexpr (F8) {                  // dequantisation: acc[m][n] *= scaleA[m] * scaleB[n], before anything else
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        float sbv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sbv[j] = p.scaleB[min(wn0 + t4 * 16 + lq * 4 + j, p.N - 1)];
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            const float sav = p.scaleA[min(em0 + wr * 128 + mh * 64 + mi * 16 + lr, p.M - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mh][t4 >> 1][t4 & 1][mi][j] *= sav * sbv[j];
          }
      }
    }
    float bv[4][4];
    int ebvec = __builtin_amdgcn_readfirstlane(bias_vec() ? 1 : 0);
    asm volatile("" : "+s"(ebvec));      // (opaque: the four loads of the top are read on every path, so that nothing stays pending)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int n = wn0 + t4 * 16 + lq * 4;
      const bool ok = ebvec != 0 && n + 3 < p.N;
      bv[t4][0] = ok ? bf2f((bf16_t)(braw[t4].x & 0xffff)) : 0.f; bv[t4][1] = ok ? bf2f((bf16_t)(braw[t4].x >> 16)) : 0.f;
      bv[t4][2] = ok ? bf2f((bf16_t)(braw[t4].y & 0xffff)) : 0.f; bv[t4][3] = ok ? bf2f((bf16_t)(braw[t4].y >> 16)) : 0.f;
      if (bias && !ok) {                   // unaligned bias / the ragged last columns: element loads
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < p.N) bv[t4][j] = bf2f(bias[n + j]);
      }
    }
    // the two 64 x 64 halves of this wave: outside C / entirely inside / on its edge
    bf16_t* Cb = p.C + (long long)ez * p.sC;
    const bf16_t* Rb = (HAS_RES && p.R) ? p.R + (long long)ez * p.sR : nullptr;
    const bool vec_ok = ((p.ldc & 7) == 0) && (!Rb || (p.ldr & 7) == 0);
    const bool plain_rows = (res_mod | gR | gC | c_live_mod) == 0;
    // a live-row filter alone (SwiGLU forward of the adapter-only step) keeps the fast path: the filter is a per-lane predicate on
    // the stores (stamped build: the general path's eight serial read -> modulo -> store rounds per live half were 4.6k cycles of
    // phase B and as much again at the closing barrier, of a 50k-cycle gate/up tile)
    const bool live_rows = EPI == 1 && (res_mod | gR | gC) == 0 && c_live_mod >= 64;
    bool skip[2], inside[2], fastm[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const int wm0 = em0 + wr * 128 + mh * 64;
      skip[mh] = wm0 >= p.M || wn0 >= p.N;          // nothing of this half exists (N = 3.5 tiles: half the waves of the last column)
      inside[mh] = wm0 + 64 <= p.M && wn0 + 64 <= p.N;
      fastm[mh] = EPI != 2 && (plain_rows || live_rows) && vec_ok && inside[mh] && !skip[mh];
    }
    // Residual segments (plain epilogue): ALWAYS sixteen loads per lane, in two straight-line groups, read on every path - a
    // conditional load reaches its registers through a copy that waits for it, and a load that some path never reads leaves the
    // compiler a pending register to protect with vmcnt(0) wherever it reuses it.  A half that takes the general path, or a
    // GEMM without residual, reads the head of B instead and ignores it.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rv[2][8];
    int fres0 = 0, fres1 = 0;
    const char* rsrc = nullptr;
    if constexpr (HAS_RES) {
      long long rdist = Rb ? (long long)((uintptr_t)Rb - (uintptr_t)p.B) : 0;
      fres0 = __builtin_amdgcn_readfirstlane((fastm[0] && Rb) ? 1 : 0);
      fres1 = __builtin_amdgcn_readfirstlane((fastm[1] && Rb) ? 1 : 0);
      asm volatile("" : "+s"(rdist), "+s"(fres0), "+s"(fres1));
      rsrc = reinterpret_cast<const char*>(p.B) + rdist;
    }
    auto load_res = [&](int mh, int it0, int it1) {
#pragma unroll
      for (int it = it0; it < it1; ++it) {
        const long long off = ((long long)(em0 + wr * 128 + mh * 64 + it * 8 + (el >> 3)) * p.ldr + wn0 + (el & 7) * 8) * 2;
        rv[mh][it] = *reinterpret_cast<const u32x4*>(rsrc + ((mh == 0 ? fres0 : fres1) ? off : 0));
      }
    };
    auto drop_res = [&](int mh) {          // a path that does not add the residual still reads the registers (see above)
      asm volatile("" ::"v"(rv[mh][0]), "v"(rv[mh][1]), "v"(rv[mh][2]), "v"(rv[mh][3]), "v"(rv[mh][4]), "v"(rv[mh][5]), "v"(rv[mh][6]),
                   "v"(rv[mh][7]));
    };
    // next tile: its K-tile 0 goes out now
    if (PRE && nxt >= 0) { setup(nxt); stage_k0(d); }
    char* const reg = PRE ? smem + (d ^ 1) * 4 * HT + wid * STG : smem + wid * STG;
    if constexpr (HAS_RES) load_res(0, 0, 4); // the first segments: in flight under the activation math (16 registers: all 32
                                               // of the half do not fit beside 128 accumulators)

    if constexpr (EPI == 2) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int wm0 = em0 + wr * 128 + mh * 64;
        auto A4 = [&](int t4, int mi) -> f32x4& { return acc[mh][t4 >> 1][t4 & 1][mi]; };
        // SwiGLU backward fused into dH = dY . W_down (accumulator = dH of this 64 x 64 patch): read the matching interleaved
          // pre-activations GU[m, 2N], emit dGU in the same layout; dH itself is never stored.
          const bf16_t* GU = p.R + (long long)ez * p.sR;
          bf16_t* Cb = p.C + (long long)ez * p.sC;
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
                const long long go = (gR > 0 ? (long long)(m / gR) * p.sgR + (long long)(m % gR) * p.ldr : (long long)m * p.ldr) +
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
              const int row = it * 4 + (el >> 4), ch = el & 15;
              const int m = wm0 + half * 32 + row, n2 = 2 * wn0 + ch * 8;
              if (m < p.M && n2 + 8 <= 2 * p.N)
                *reinterpret_cast<uint4*>(Cb + (long long)m * p.ldc + n2) = *reinterpret_cast<const uint4*>(reg2 + row * ROWB + ch * 16);
            }
          }
      }
    } else {
      auto flo = [](unsigned v) { return __builtin_bit_cast(float, v << 16); };
      auto fhi = [](unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); };

      // ---- phase A: alpha / bias / activation on all 128 values of the lane, rounded and packed.  The fp32 accumulators end
      //      here: what the stores below carry is half the registers, and the residual segments fit beside it.
      uint2 pk[2][4][4];          // [mh][t4 = 2 nh + ni][mi]: columns t4*16 + lq*4 .. +3 of row mi*16 + lr
      uint2 hp[2][2][4];          // SwiGLU forward: the products h
      if constexpr (EPI == 1) {
        // columns interleaved in 16s - even n tiles are gate, odd n tiles the matching up columns
        // (Qwen2's gate / up projections carry no bias and alpha is 1: the wave-uniform `plain` spares two FMAs per pair)
        const bool plain = bias == nullptr && p.alpha == 1.f;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              f32x4 ga = acc[mh][pr][0][mi], ua = acc[mh][pr][1][mi];
              if (!plain) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { ga[j] = ga[j] * p.alpha + bv[2 * pr][j]; ua[j] = ua[j] * p.alpha + bv[2 * pr + 1][j]; }
              }
              const uint2 gp = {pack2(ga[0], ga[1]), pack2(ga[2], ga[3])}, up = {pack2(ua[0], ua[1]), pack2(ua[2], ua[3])};
              const float g[4] = {flo(gp.x), fhi(gp.x), flo(gp.y), fhi(gp.y)}, u[4] = {flo(up.x), fhi(up.x), flo(up.y), fhi(up.y)};
              float h[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) h[j] = rbf(g[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-g[j]))) * u[j];
              pk[mh][2 * pr][mi] = gp;
              pk[mh][2 * pr + 1][mi] = up;
              hp[mh][pr][mi] = uint2{pack2(h[0], h[1]), pack2(h[2], h[3])};
            }
      } else {
        auto phase_a = [&](auto fn, auto post_t) {
          constexpr bool POST = decltype(post_t)::value;     // bf16(bf16(alpha acc) + bias): torch CPU Linear on a strided input
          const float alpha = p.alpha;
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
              for (int mi = 0; mi < 4; ++mi) {
                const f32x4 a = acc[mh][t4 >> 1][t4 & 1][mi];
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  x[j] = fn(POST ? rbf(a[j] * alpha) + bv[t4][j] : a[j] * alpha + bv[t4][j]);
                }
                pk[mh][t4][mi] = uint2{pack2(x[0], x[1]), pack2(x[2], x[3])};
              }
        };
        // HF rotate_half RoPE on the projected q / k columns (rope_mode 1, head dim 64 == this wave's 64 columns: d <-> d + 32 are
        // the n tiles t4 and t4 + 2 of the same lane), on the bf16-rounded projection, every product rounded - the arithmetic of
        // gemm.hip's fused epilogue, value for value
        auto phase_a_rope = [&]() {
          const float alpha = p.alpha;
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              const int pos = min(em0 + wr * 128 + mh * 64 + mi * 16 + lr, p.M - 1) % p.rope_T;
#pragma unroll
              for (int ni = 0; ni < 2; ++ni) {
                const int d = ni * 16 + lq * 4;
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.rope_cos + (long long)pos * 32 + d);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(p.rope_sin + (long long)pos * 32 + d);
                const f32x4 xa = acc[mh][0][ni][mi], xb = acc[mh][1][ni][mi];       // n tiles ni and ni + 2
                float ya[4], yb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float a = rbf(xa[j] * alpha + bv[ni][j]), b = rbf(xb[j] * alpha + bv[ni + 2][j]);
                  ya[j] = rbf(a * c[j]) + rbf(-b * sn[j]);
                  yb[j] = rbf(b * c[j]) + rbf(a * sn[j]);
                }
                pk[mh][ni][mi] = uint2{pack2(ya[0], ya[1]), pack2(ya[2], ya[3])};
                pk[mh][ni + 2][mi] = uint2{pack2(yb[0], yb[1]), pack2(yb[2], yb[3])};
              }
              __builtin_amdgcn_sched_barrier(0);      // (one row block's table segments at a time)
            }
        };
        // interleaved RoPE of the action head (action_heads.py:125-146: pairs (2i, 2i+1), tables of cat([f, f]); rope_mode 2): the
        // rotation partners are neighbours inside a lane's four columns - gemm.hip's fused epilogue value for value, on the
        // bf16-rounded projection, every product rounded.  (The head's task-token K|V projection, 8192 x 1792: fused it used to leave
        // this kernel for the 128-row one - 76 us in situ against 39 + the stand-alone pass's 10-20.)
        auto phase_a_rope2 = [&](auto post_t) {
          constexpr bool POST = decltype(post_t)::value;
          const float alpha = p.alpha;
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              const int pos = min(em0 + wr * 128 + mh * 64 + mi * 16 + lr, p.M - 1) % p.rope_T;
#pragma unroll
              for (int t4 = 0; t4 < 4; ++t4) {
                const int d = (wn0 + t4 * 16 + lq * 4) % p.rope_dh;
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.rope_cos + (long long)pos * p.rope_dh + d);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(p.rope_sin + (long long)pos * p.rope_dh + d);
                const f32x4 a = acc[mh][t4 >> 1][t4 & 1][mi];
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  x[j] = rbf(POST ? rbf(a[j] * alpha) + bv[t4][j] : a[j] * alpha + bv[t4][j]);
                }
                const float y0 = rbf(x[0] * c[0]) + rbf(-x[1] * sn[0]), y1 = rbf(x[1] * c[1]) + rbf(x[0] * sn[1]);
                const float y2 = rbf(x[2] * c[2]) + rbf(-x[3] * sn[2]), y3 = rbf(x[3] * c[3]) + rbf(x[2] * sn[3]);
                pk[mh][t4][mi] = uint2{pack2(y0, y1), pack2(y2, y3)};
                if (t4 & 1) __builtin_amdgcn_sched_barrier(0);      // (two n tiles' table segments at a time)
              }
            }
        };
        if (p.rope_mode == 1 && wn0 < p.rope_cols) phase_a_rope();
        else if (R2 && p.rope_mode == 2 && wn0 < p.rope_cols) {      // (its own instantiation: the table segments cost registers)
          if (p.bias_post) phase_a_rope2(std::true_type{});
          else phase_a_rope2(std::false_type{});
        } else if (p.bias_post) phase_a([](float v) { return v; }, std::true_type{});       // (plain epilogue only: checked by the host)
        else if (p.act == VLA_ACT_GELU) phase_a([](float v) { return gelu_erf(rbf(v)); }, std::false_type{});
        else if (p.act == VLA_ACT_RELU) phase_a([](float v) { return fmaxf(v, 0.f); }, std::false_type{});
        else if (p.act == VLA_ACT_GELU_TANH) phase_a([](float v) { return gelu_tanh(rbf(v)); }, std::false_type{});
        else phase_a([](float v) { return v; }, std::false_type{});
      }
      // Phase A ends HERE, for the compiler too: every packed value is pinned, and the lane index that phase B computes its
      // addresses from is only defined afterwards (left alone, the residual loads are hoisted above the packing and the
      // second half's packing sunk below the first half's stores: 128 accumulators + 64 residual registers, and spills).
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
          asm volatile("" : "+v"(pk[mh][t4][0].x), "+v"(pk[mh][t4][0].y), "+v"(pk[mh][t4][1].x), "+v"(pk[mh][t4][1].y),
                            "+v"(pk[mh][t4][2].x), "+v"(pk[mh][t4][2].y), "+v"(pk[mh][t4][3].x), "+v"(pk[mh][t4][3].y));
      asm volatile("" : "+v"(el));
      // ---- phase B: per 64 x 64 half, through the wave's staging region, 16-B stores of whole row segments.  The common case
      //      - the half entirely inside C, 16-B aligned rows, plain row addressing - runs without a branch and with all its
      //      residual segments requested up front (in the general path every conditional load is followed by its own
      //      vmcnt(0): 16 serial round trips per tile).
      if constexpr (HAS_RES) { load_res(0, 4, 8); load_res(1, 0, 8); }     // the rest: behind the first stores
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int wm0 = em0 + wr * 128 + mh * 64;
        if (skip[mh]) {
          if constexpr (HAS_RES) drop_res(mh);
          continue;
        }
        if constexpr (EPI == 1) {
          // the product h (64 rows x 32 columns per half) is staged through LDS like C, so that it leaves as 16-B row segments
          bf16_t* C2 = p.C2 + (long long)ez * p.sC2;
          constexpr int HSTR = 32 * 2 + 16;                       // staged h row: 32 bf16 + pad (64 rows: 5 KiB of the region)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
              *reinterpret_cast<uint2*>(reg + (mi * 16 + lr) * HSTR + (pr * 16 + lq * 4) * 2) = hp[mh][pr][mi];
          const bool hvec = ((p.ldc2 & 7) == 0) && (((size_t)C2 & 15) == 0);
          uint4 hv[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) hv[it] = *reinterpret_cast<const uint4*>(reg + (it * 16 + (el >> 2)) * HSTR + (el & 3) * 16);
          __builtin_amdgcn_sched_barrier(0);                      // (the four staged segments in flight before the first store)
#pragma unroll
          for (int it = 0; it < 4; ++it) {                        // 64 rows x 4 chunks of 16 B: 16 rows per pass
            const int row = it * 16 + (el >> 2), ch = el & 3;
            const int m = wm0 + row, hc = (wn0 >> 1) + ch * 8;
            const uint4 v = hv[it];
            if (hvec && inside[mh]) {                             // (wave-uniform) whole half inside: no per-lane test
              *reinterpret_cast<uint4*>(C2 + (long long)m * p.ldc2 + hc) = v;
              continue;
            }
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
          if (c_live_mod > 0 && (wm0 % c_live_mod) + 63 < p.c_live_from) continue;
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            const int row = mi * 16 + lr;
            *reinterpret_cast<uint2*>(reg + row * 128 + (((t4 * 2 + (lq >> 1)) ^ ((row >> 1) & 7)) << 4) + (lq & 1) * 8) = pk[mh][t4][mi];
          }
        if (fastm[mh]) {
          // all eight staged segments are requested before the first store waits for one (one LDS round trip per half instead
          // of eight serial ones: 270 cycles each on the stamped build - the epilogue's stores are not what it waits for)
          uint4 vv[8];
#pragma unroll
          for (int it = 0; it < 8; ++it) {                       // 64 rows x 8 chunks of 16 B: 8 rows per pass
            const int row = it * 8 + (el >> 3), ch = el & 7;
            vv[it] = *reinterpret_cast<const uint4*>(reg + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
          }
          __builtin_amdgcn_sched_barrier(0);
          auto store_fast = [&](auto live_t) {
          constexpr bool LIVE = decltype(live_t)::value;         // live-row filter: a per-lane predicate on the stores
          int lpos = 0;                                          // position of the half's first row in its row group
          if constexpr (LIVE) lpos = __builtin_amdgcn_readfirstlane(wm0 % c_live_mod);
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (el >> 3), ch = el & 7;
            if constexpr (LIVE) {                                // (64 rows wrap at most once: c_live_mod >= 64)
              int pos = lpos + row;
              if (pos >= c_live_mod) pos -= c_live_mod;
              if (pos < p.c_live_from) continue;
            }
            uint4 v = vv[it];
            if constexpr (HAS_RES) {
              const u32x4 r = rv[mh][it];
              const unsigned a[4] = {v.x, v.y, v.z, v.w};
              unsigned o[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) o[k] = (mh == 0 ? fres0 : fres1) ? pack2(flo(a[k]) + flo(r[k]), fhi(a[k]) + fhi(r[k])) : a[k];
              v = uint4{o[0], o[1], o[2], o[3]};
            }
            *reinterpret_cast<uint4*>(Cb + (long long)(wm0 + row) * p.ldc + wn0 + ch * 8) = v;
          }
          };
          if (EPI == 1 && c_live_mod > 0) store_fast(std::true_type{});
          else store_fast(std::false_type{});
          continue;
        }
        // Row addressing: the plain case (no row groups, no broadcast residual, no live-row filter) must not pay the four integer
        // divisions per output row of the general case - 128 divisions per wave sat on every tile's tail.
        auto store_rows = [&](auto fast_t) {
          constexpr bool FAST = decltype(fast_t)::value;
#pragma nounroll
          for (int it = 0; it < 8; ++it) {                       // 64 rows x 8 chunks of 16 B: 8 rows per pass (rare path: kept rolled)
            const int row = it * 8 + (el >> 3), ch = el & 7;
            const int m = wm0 + row, n = wn0 + ch * 8;
            uint4 v = *reinterpret_cast<const uint4*>(reg + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
            if (m >= p.M || n >= p.N) continue;
            long long roff, crow;
            if constexpr (FAST) {
              roff = (long long)m * p.ldr;
              crow = (long long)m * p.ldc;
            } else {
              if (c_live_mod > 0 && (m % c_live_mod) < p.c_live_from) continue;
              roff = res_mod > 0 ? (long long)(m % res_mod) * p.ldr
                     : gR > 0 ? (long long)(m / gR) * p.sgR + (long long)(m % gR) * p.ldr : (long long)m * p.ldr;
              crow = gC > 0 ? (long long)(m / gC) * p.sgC + (long long)(m % gC) * p.ldc : (long long)m * p.ldc;
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
        if constexpr (HAS_RES) drop_res(mh);
        if (plain_rows) store_rows(std::true_type{});
        else store_rows(std::false_type{});
      }
    }

    if (nxt < 0) break;
    VLA_BARRIER();                   // every wave is done with its staging region: K-tile 1 may land there
    if (!PRE) { setup(nxt); stage_k0(d); }
    cur = nxt;
  }
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

template <int EPI, bool F8 = false, bool RES = true, bool R2 = false, bool EXT = false>
int launch256(const GemmP& p0, int batch, hipStream_t st) {
  GemmP p = p0;
  p.tiles_n = (p.N + 255) / 256;
  p.ntiles = ((p.M + 255) / 256) * p.tiles_n;
  p.batch = batch;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI, F8, RES, R2, EXT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  // one workgroup per CU walks the tiles (VLA_GEMM256_GRID overrides the workgroup count: 0 = one workgroup per tile)
  const char* ge = getenv("VLA_GEMM256_GRID");      // (read per launch: the A/B tool flips it in one process)
  const long long total = (long long)p.ntiles * batch;
  // Free de-phasing of a partial last round (kernel: start of the walk).  Every workgroup has the same work per tile, so all 256
  // epilogues burst their stores into the fabric in the same few microseconds and wait for it; the workgroups that walk one tile
  // fewer have a tile's time to spare, and starting them late by half a tile puts their bursts between those of the others.
  // Same box: gate/up 203 -> 196 us, d->dh 97 -> 95, exact-round launches unchanged, step -0.15 ... -0.2 ms.  (A delay for EVERY
  // group of workgroups - not only those with slack - shortens the epilogues by what the delay costs: measured, no gain.)
  // VLA_GEMM256_STAGGER = percent of the estimated tile time (default 50, 0 = off; read per launch for the A/B tools).
  const char* se = getenv("VLA_GEMM256_STAGGER");
  long long grid = ge ? atoll(ge) : num_cus();
  if (grid <= 0 || grid > total) grid = total;
  p.stagger = 0;
  if (total > grid && total % grid != 0) {
    const long long tile_cycles = (long long)((p.K + (EXT ? p.K2 : 0)) / BK) * 2128 + 13000;       // K loop + what surrounds it (stamped: DESIGN section 4)
    p.stagger = (int)(tile_cycles * (se != nullptr ? atoi(se) : 50) / 100 / 1024);
  }
  hipLaunchKernelGGL((gemm256_kernel<EPI, F8, RES, R2, EXT>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st, p);
  return 0;
}

}  // namespace

int vla_num_cus() { return num_cus(); }

int vla_gemm256_launch(const GemmP& p, int epi, int batch, hipStream_t st) {
  if (p.K2 > 0) {                    // K extension (host: bf16, batch 1, plain / residual / rotate_half or SwiGLU-forward epilogue)
    if (epi == 1) return launch256<1, false, true, false, true>(p, 1, st);
    return p.R ? launch256<0, false, true, false, true>(p, 1, st) : launch256<0, false, false, false, true>(p, 1, st);
  }
  if (p.scaleA != nullptr) return epi == 1 ? launch256<1, true>(p, batch, st) : launch256<0, true>(p, batch, st);     // fp8 operands
  if (p.rope_mode == 2) return launch256<0, false, false, true>(p, batch, st);     // (host: plain epilogue, no residual)
  if (epi == 1) return launch256<1>(p, batch, st);
  if (epi == 2) return launch256<2>(p, batch, st);
  return p.R ? launch256<0, false, true>(p, batch, st) : launch256<0, false, false>(p, batch, st);
}
