// bf16 NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)   (both operands K-contiguous, the
// nn.Linear layout; fp32 accumulation on v_mfma_f32_16x16x32_bf16).
//
// Replaces every cuBLAS call the reference reaches through nn.Linear: timm ViT qkv/proj/fc1/fc2,
// PrismaticProjector (modeling_prismatic.py:261-273), Qwen2 q/k/v/o/gate/up/down, the action head's Linears
// (action_heads.py:337-410), and - with pre-transposed operands - their dX / dW products.
//
// Structure (cdna_hip_programming.md section 5):
//   * block tile BM x BN x 64 with (BM/64) x 2 waves, each wave a 64 x (BN/2) sub-tile of 16x16 MFMA tiles;
//   * A/B tiles stream HBM->LDS with global_load_lds_dwordx4 (1 KiB per wave-instruction) into a ring of STAGES
//     buffers; the LDS image is linear with the 16-B-chunk XOR swizzle applied on the SOURCE address and on the
//     fragment read (rule 21);
//   * ONE raw s_barrier per K-tile; with STAGES=3 the loads run TWO K-tiles ahead and the loop waits with a COUNTED
//     s_waitcnt vmcnt(pieces-per-stage) - never vmcnt(0) inside the loop ("Pipelining across barriers"): a tile's
//     pieces are retired by every wave's own counted wait, the barrier then publishes them to the other waves and
//     proves every wave has finished reading the buffer that the next stage overwrites;
//   * operands are swapped at the MFMA (A-operand := B rows) so each lane owns 4 consecutive n of one row m: the
//     epilogue stages the wave's tile through LDS and stores whole row segments with 16-B lanes;
//   * XCD-aware bijective tile order (T1).
// Three instantiations: 256x128 (8 waves, 3 stages, 144 KiB LDS, 1 block/CU) for the large GEMMs, 128x128 (4 waves,
// 2 stages, 2 blocks/CU) and 128x64 (4 waves, 2 stages, 48 KiB, 3 blocks/CU) - the host picks per problem by a
// wave-quantisation model (a 616-tile problem on 512 slots wastes 40 % of the machine with 128x128 tiles).
//
// Epilogue rounding points follow the reference under bf16 autocast: bf16(acc+bias) -> act -> bf16 -> (+residual) -> bf16.
#include <atomic>
#include "common.h"
#include "gemm_params.h"
#include "../../include/vla_native.h"

namespace {

constexpr int BK = 64;
constexpr int EPI_PAD = 16;  // bytes of padding per staged epilogue row (keeps 16-B alignment, spreads banks)


__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case VLA_ACT_GELU: return rbf(gelu_erf(v));
    case VLA_ACT_RELU: return fmaxf(v, 0.f);
    case VLA_ACT_GELU_TANH: return rbf(gelu_tanh(v));
    default: return v;
  }
}

typedef int v8i_f8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8i_f8 cat8(bf16x8 lo, bf16x8 hi) {          // two 16-B fragment reads -> the 32-byte fp8 operand
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x4 l = __builtin_bit_cast(f32x4, lo), h = __builtin_bit_cast(f32x4, hi);
  return __builtin_bit_cast(v8i_f8, f32x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]});
}

template <int BM, int BN, int STAGES, int WN = 2>
struct Cfg {
  static constexpr int NW = (BM / 64) * WN;              // waves: (BM/64) along M x WN along N
  static constexpr int NTHREADS = NW * 64;
  static constexpr int WTN = BN / WN;                    // wave tile: 64 x WTN
  static constexpr int NT = WTN / 16;                    // 16-wide n tiles per wave (4 or 2)
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int PIECES = STAGE_BYTES / 1024;      // 1 KiB LDS-DMA pieces per stage
  static constexpr int PPW = PIECES / NW;                // pieces per wave per stage
  static constexpr int EPI_STRIDE = WTN * 2 + EPI_PAD;   // bytes per staged output row
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static constexpr int NHALF = NW * 64 * EPI_STRIDE <= LDS_BYTES ? 1 : 2;   // output staged in 1 or 2 row-halves per wave
  static constexpr int EPI_ROWS = 64 / NHALF;
  static constexpr int EPI_BYTES = NW * EPI_ROWS * EPI_STRIDE;
  static_assert(EPI_BYTES <= LDS_BYTES, "epilogue staging must fit the operand ring");
  static_assert(PIECES % NW == 0, "pieces must divide evenly over the waves");
};

// ROPE is a compile-time switch (0 none / 1 rotate_half / 2 interleaved): the epilogue's extra registers and table
// loads must not leak into the plain kernel that every other GEMM of the step runs.
// F8: operands are OCP e4m3 bytes with per-row fp32 scales (p.scaleA[m], p.scaleB[n]); a K-tile is still 128 B per row = 128
// elements = ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile (unit block scales): the same bytes moved, the same LDS
// reads and the same MFMA cycles per K-tile as the bf16 form, at twice the FLOPs (lane l holds row l % 16, k = 32 (l / 16) +
// byte: tools/probe_f8.py).  The scales are applied to the fp32 accumulator before anything else of the epilogue.
// EXT: K extension - the contraction continues over a second operand pair, C = epilogue(A . B^T + A2 . B2^T) with A2 [M, K2]
// (lda2) and B2 [N, K2] (ldb2): K-tiles [0, K / 64) stream from (A, B), the rest from (A2, B2), one accumulator, one rounding.
// This is how a LoRA-wrapped Linear runs (vla-scripts/finetune.py:832-844): y = x W^T + (2 x A^T) B^T with the low-rank
// branch inside the base GEMM's fp32 accumulator - the base product keeps its bias / RoPE / SwiGLU epilogue and no
// read-modify-write pass over y exists (backward alike: dx = dy W + dt A).
template <int BM, int BN, int STAGES, int ROPE, int WN = 2, bool F8 = false, bool EXT = false>
__global__ __launch_bounds__((BM / 64) * 64 * WN) __attribute__((amdgpu_waves_per_eu((F8 && EXT) ? 4 : 1)))   // (fp8 + extension: two
void gemm_nt_kernel(GemmP p) {                                   //  workgroups per CU like every other form: 128 registers, no spill)
  using C = Cfg<BM, BN, STAGES, WN>;
  constexpr int EB = F8 ? 1 : 2;                           // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid - wr * WN;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
  // contiguous run of tiles so neighbouring tiles (same A row-panel) hit the same L2.
  const int nwg = p.ntiles, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // group-M order inside the XCD's run: GM row-panels x all column tiles, row index fastest, so the ~32-64 tiles an
  // XCD has in flight form a GM x (32/GM) patch whose A panels stay in that XCD's 4 MiB L2 while B tiles stream once
  // (row-major order re-streamed every B tile from beyond L2 for every row panel: 44 x 17 MB for the gate/up GEMM).
  // 6 for the 128-row tiles: in the two/three-stream step the XCD's L2 is shared with the other streams' kernels and the
  // smaller patch wins (same-box sweep: 8 -> 31.55 ms/step, 6 -> 30.85, 5 -> 30.8, 4 -> 31.0, 3 -> 30.85, 12 -> 32.3;
  // isolated launches are indifferent between 4 and 8).
  constexpr int GM = 6;                                    // A panels kept L2-resident per XCD
  const int tiles_m = p.ntiles / p.tiles_n, per_group = GM * p.tiles_n;
  const int grp = swz / per_group, rem = swz - grp * per_group;
  const int gm = min(GM, tiles_m - grp * GM);
  const int bm = grp * GM + rem % gm, bn = rem / gm;
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;
  const char* Ab = reinterpret_cast<const char*>(p.A) + (long long)z * p.sA * EB;
  const char* Bb = reinterpret_cast<const char*>(p.B) + (long long)z * p.sB * EB;

  // ---- staging: piece pc (1 KiB = 8 LDS rows) of a stage; pieces [0, BM/8) are A rows, the rest B rows.
  //      lane -> row 8pc + (lane>>3); LDS chunk lane&7 holds global chunk (lane&7) ^ (row&7)
  const int kc = ((lane & 7) ^ ((lane >> 3) & 7)) * 16;      // bytes
  const char* pp[C::PPW];
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    const int pc = wid * C::PPW + i;                      // wave-uniform
    if (pc < BM / 8) {
      const int ra = min(m0 + pc * 8 + (lane >> 3), p.M - 1);
      pp[i] = Ab + (p.gA > 0 ? (long long)(ra / p.gA) * p.sgA + (long long)(ra % p.gA) * p.lda : (long long)ra * p.lda) * EB + kc;
    } else {
      const int rb = min(n0 + (pc - BM / 8) * 8 + (lane >> 3), p.N - 1);
      pp[i] = Bb + (long long)rb * p.ldb * EB + kc;
    }
  }
  const char* pp2[EXT ? C::PPW : 1];
  if constexpr (EXT) {
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) {
      const int pc = wid * C::PPW + i;
      if (pc < BM / 8) pp2[i] = reinterpret_cast<const char*>(p.A2) + (long long)min(m0 + pc * 8 + (lane >> 3), p.M - 1) * p.lda2 * 2 + kc;
      else pp2[i] = reinterpret_cast<const char*>(p.B2) + (long long)min(n0 + (pc - BM / 8) * 8 + (lane >> 3), p.N - 1) * p.ldb2 * 2 + kc;
    }
  }
  // (split-K: slice z owns p.K of the contraction - the last slice what is left of p.Ktot when the slices are uneven)
  const int nt1 = (p.Ktot > 0 ? min(p.K, p.Ktot - z * p.K) : p.K) / (F8 ? 2 * BK : BK);        // K-tiles of 128 B per row from (A, B)
  auto stage = [&](int buf, int t) {               // K-tile t: 128 B further along every row per tile
    char* base = smem + buf * C::STAGE_BYTES + wid * C::PPW * 1024;
    if (EXT && t >= nt1) {
#pragma unroll
      for (int i = 0; i < C::PPW; ++i) glds16(pp2[i] + (t - nt1) * 128, base + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < C::PPW; ++i) glds16(pp[i] + t * 128, base + i * 1024);
    }
  };

  f32x4 acc[C::NT][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < C::NT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside an operand tile, per k-step s: row*128 + ((4s + (lane>>4)) ^ (lane&7))*16
  const int frow = lane & 15;
  int foff[2], foff2[2];                      // foff2: the bf16 K-extension tiles of an fp8 product (F8 && EXT)
#pragma unroll
  for (int s = 0; s < 2; ++s) {               // (fp8: the lane's 32 contiguous bytes = chunks 2q and 2q + 1, q = lane >> 4)
    foff[s] = frow * 128 + (((F8 ? 2 * (lane >> 4) + s : 4 * s + (lane >> 4)) ^ (lane & 7)) << 4);
    foff2[s] = frow * 128 + (((4 * s + (lane >> 4)) ^ (lane & 7)) << 4);
  }

  // epilogue coordinates: lane owns, for tile (ni, mi): m = 16mi + (lane&15), n = 16ni + 4(lane>>4) + {0..3}.
  // The bias slice is fetched HERE (one 8-B load per n-tile) so its latency hides under the main loop; fetched in the
  // epilogue it cost a dependent global load per element right on every tile's tail.
  const int wm0 = m0 + wr * 64, wn0 = n0 + wc * C::WTN;
  const int lq = lane >> 4, lr = lane & 15;
  // Tile-local first column of this wave's 16-wide n tile ni.  Plain: WTN consecutive columns.  Fused rotate_half on the
  // 8-wave geometry (32 columns per wave, head dim 64): the wave owns 16 columns of EACH half of one head, so that the
  // rotation partners d <-> d + 32 are n tiles 0 and 1 of the same lane (no exchange between waves).
  // Head dim 128 (Qwen2.5-1.5B, round 4): the tile's 128 columns are ONE head, the halves are columns [0, 64) and [64, 128): wave wc owns
  // 16 columns of each (n tile 0: 16 wc .., n tile 1: 64 + 16 wc ..) - the same lane-local rotation, another column map.
  constexpr bool PERM = ROPE == 1 && C::NT == 2;
  const bool wide = PERM && p.rope_dh == 128;          // (wave-uniform)
  auto cbase = [&](int ni) { return PERM ? (wide ? ni * 64 + wc * 16 : (wc >> 1) * 64 + ni * 32 + (wc & 1) * 16) : wc * C::WTN + ni * 16; };
  const bf16_t* bias = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
  float bv[C::NT][4];
  {
    const bool bvec = bias && (((size_t)bias & 7) == 0);
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) {
      const int n = n0 + cbase(ni) + lq * 4;
      if (bvec && n + 3 < p.N) {
        const uint2 b2 = *reinterpret_cast<const uint2*>(bias + n);
        bv[ni][0] = bf2f((bf16_t)(b2.x & 0xffff)); bv[ni][1] = bf2f((bf16_t)(b2.x >> 16));
        bv[ni][2] = bf2f((bf16_t)(b2.y & 0xffff)); bv[ni][3] = bf2f((bf16_t)(b2.y >> 16));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[ni][j] = (bias && n + j < p.N) ? bf2f(bias[n + j]) : 0.f;
      }
    }
  }

  // (fp8 base product with a bf16 K extension - a LoRA-wrapped Linear on e4m3 base operands: the dequantisation scales are applied
  //  to the accumulator BETWEEN the two contractions, so the low-rank branch adds to the dequantised base product)
  const int nt = nt1 + (EXT ? p.K2 / BK : 0);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nt) stage(s, s);
  int buf = 0;
  static_assert(C::NT <= 4, "wave tiles are 64 x 32 or 64 x 64");
  static_assert(STAGES >= 2 && STAGES <= 6 && 4 * C::PPW < 64, "the counted waits below cover rings of 2 to 6 stages (vmcnt is a 6-bit counter)");
  // One K-tile (written as a macro so that the fp8 product with a bf16 extension can run it as TWO loops - e4m3 tiles, then bf16
  // tiles - each with one MFMA form and one set of fragment offsets: as one loop with a per-tile branch the kernel needed 200
  // registers and lost its second resident workgroup).  All fragment reads of the K-tile (both 32-deep k-steps) are issued up
  // front: the second k-step's LDS latency hides under the first k-step's MFMAs instead of stalling between them.
#define VLA_KTILE(F8T, FO)                                                                                                           \
  do {                                                                                                                               \
    /* retire this wave's pieces of tile t; up to STAGES-2 younger tiles stay in flight (in-order return: a counted wait) */       \
    if (STAGES >= 6 && t + 4 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * C::PPW) : "memory");                                  \
    else if (STAGES >= 5 && t + 3 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * C::PPW) : "memory");                             \
    else if (STAGES >= 4 && t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::PPW) : "memory");                             \
    else if (STAGES >= 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PPW) : "memory");                                 \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                            \
    __builtin_amdgcn_s_barrier(); /* tile t visible to every wave; every wave is done reading tile t-1's buffer */                    \
    asm volatile("" ::: "memory");                                                                                                   \
    if (t + STAGES - 1 < nt) {                                                                                                       \
      int nb = buf + STAGES - 1;                                                                                                     \
      if (nb >= STAGES) nb -= STAGES;                                                                                                \
      stage(nb, t + STAGES - 1);                                                                                                     \
    }                                                                                                                                \
    const char* sa = smem + buf * C::STAGE_BYTES + wr * 64 * 128;                                                                    \
    const char* sb = smem + buf * C::STAGE_BYTES + C::A_BYTES; /* B tile; n tile i of this wave starts at row cbase(i) */             \
    bf16x8 fm[2][4], fn[2][C::NT];                                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) fm[s][i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 128 + FO[s]);          \
      _Pragma("unroll") for (int i = 0; i < C::NT; ++i) fn[s][i] = *reinterpret_cast<const bf16x8*>(sb + cbase(i) * 128 + FO[s]);    \
    }                                                                                                                                \
    if constexpr (F8T) {                                                                                                             \
      _Pragma("unroll") for (int ni = 0; ni < C::NT; ++ni)                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                             \
          acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(fn[0][ni], fn[1][ni]), cat8(fm[0][mi], fm[1][mi]),     \
                                                                         acc[ni][mi], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);            \
    } else {                                                                                                                         \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                                  \
        _Pragma("unroll") for (int ni = 0; ni < C::NT; ++ni)                                                                         \
          _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                           \
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[s][ni], fm[s][mi], acc[ni][mi], 0, 0, 0);                       \
    }                                                                                                                                \
    if (++buf == STAGES) buf = 0;                                                                                                    \
  } while (0)
  if constexpr (F8 && EXT) {
    for (int t = 0; t < nt1; ++t) VLA_KTILE(true, foff);
    // dequantisation of the base product, then the bf16 extension adds to it
    {
      float fsa[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fsa[mi] = p.scaleA[min(wm0 + mi * 16 + lr, p.M - 1)];
#pragma unroll
      for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sbv = p.scaleB[min(n0 + cbase(ni) + lq * 4 + j, p.N - 1)];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[ni][mi][j] *= fsa[mi] * sbv;
        }
    }
    for (int t = nt1; t < nt; ++t) VLA_KTILE(false, foff2);
  } else {
    for (int t = 0; t < nt; ++t) VLA_KTILE(F8, foff);
  }
#undef VLA_KTILE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all waves done with the operand tiles before the staging regions are overwritten
  asm volatile("" ::: "memory");

  // ---------------- epilogue ----------------
  if constexpr (F8 && !EXT) {
    // dequantisation: acc[m][n] *= scaleA[m] * scaleB[n] (fp32, before alpha / bias / activation)
    float sa[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) sa[mi] = p.scaleA[min(wm0 + mi * 16 + lr, p.M - 1)];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni) {
      const int n = n0 + cbase(ni) + lq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sbv = p.scaleB[min(n + j, p.N - 1)];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi][j] *= sa[mi] * sbv;
      }
    }
  }
  if (ROPE == 0 && p.ws != nullptr) {
    // split-K slice z: park the raw accumulators in its own fp32 plane ws[z][M][N] (plain 16-B stores; fp32 atomics into one
    // plane were throughput-bound: 4 M atomics per GEMM); the planes are summed and bias / activation / residual / bf16
    // rounding applied once, in splitk_finalize_kernel.  (Few-tile long-K problems are otherwise one serial K loop.)
    float* plane = p.ws + (long long)z * p.M * p.N;
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = wm0 + mi * 16 + lr, n = wn0 + ni * 16 + lq * 4;
        if (m < p.M && n + 3 < p.N)
          *reinterpret_cast<f32x4*>(plane + (long long)m * p.N + n) = acc[ni][mi];
      }
    return;
  }
  char* reg = smem + wid * (C::EPI_ROWS * C::EPI_STRIDE);

  if (ROPE == 3) {
    // ---- SwiGLU backward fused into the dH = dY.W_down GEMM (VLA_ACT_SWIGLU_BWD): the accumulator holds dH for this
    // wave's 64 x 64 patch; read the matching interleaved pre-activations GU[m, 2I], emit dGU in the same layout.
    // dH is never written (saves a 110 MB write + read per layer) and the stand-alone pass (548 MB of traffic) is gone.
    const bf16_t* GU = p.R + (long long)z * p.sR;          // aux input rides in the residual slot, row stride ldr
    bf16_t* Cb = p.C + (long long)z * p.sC;
    constexpr int ROWB = 2 * C::WTN * 2 + 16;              // staged row: 2*WTN bf16 (+16 B pad)
    constexpr int CH2 = 2 * C::WTN / 8, RPP2 = 64 / CH2;   // 16-B chunks per staged row, rows per pass
    constexpr int NIT = 32 / RPP2;                         // passes per 32-row half
    char* reg2 = smem + wid * (32 * ROWB);
    if (wn0 >= p.N || wm0 >= p.M) return;                  // (N % 64 == 0: a wave's WTN h-columns are all inside or all outside)
    // The pre-activations of the wave's patch arrive the way its results leave: whole 16-B row segments, ALL requested before
    // the first is used (row clamped, no per-element condition), staged in the wave's LDS region and overwritten IN PLACE by dGU
    // (same interleaved layout).  The former per-lane 8-B gathers (16 rows x 32 B per load instruction, each behind its own
    // bounds branch and wait) made this epilogue as long as the K loop of the live-row backward's M = 2048 GEMM.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 gin[2][NIT];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = it * RPP2 + lane / CH2, ch = lane % CH2;
        const int m = min(wm0 + half * 32 + row, p.M - 1);
        const long long ro = p.gR > 0 ? (long long)(m / p.gR) * p.sgR + (long long)(m % p.gR) * p.ldr : (long long)m * p.ldr;
        gin[half][it] = *reinterpret_cast<const u32x4*>(GU + ro + 2 * wn0 + ch * 8);
      }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = it * RPP2 + lane / CH2, ch = lane % CH2;
        *reinterpret_cast<u32x4*>(reg2 + row * ROWB + ch * 16) = gin[half][it];
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int mi = 2 * half + mh;
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
          char* rowp = reg2 + (mh * 16 + lr) * ROWB + (ni * 32 + lq * 4) * 2;
          const uint2 gv = *reinterpret_cast<const uint2*>(rowp);
          const uint2 uv = *reinterpret_cast<const uint2*>(rowp + 32);
          const float gg[4] = {bf2f((bf16_t)(gv.x & 0xffff)), bf2f((bf16_t)(gv.x >> 16)), bf2f((bf16_t)(gv.y & 0xffff)), bf2f((bf16_t)(gv.y >> 16))};
          const float uu[4] = {bf2f((bf16_t)(uv.x & 0xffff)), bf2f((bf16_t)(uv.x >> 16)), bf2f((bf16_t)(uv.y & 0xffff)), bf2f((bf16_t)(uv.y >> 16))};
          float dg[4], du[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = rbf(acc[ni][mi][j] * p.alpha);
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-gg[j]));
            du[j] = d * gg[j] * sg;
            dg[j] = d * uu[j] * (sg * (1.0f + gg[j] * (1.0f - sg)));
          }
          *reinterpret_cast<uint2*>(rowp) = uint2{pack2(dg[0], dg[1]), pack2(dg[2], dg[3])};
          *reinterpret_cast<uint2*>(rowp + 32) = uint2{pack2(du[0], du[1]), pack2(du[2], du[3])};
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = it * RPP2 + lane / CH2, ch = lane % CH2;
        const int m = wm0 + half * 32 + row, n2 = 2 * wn0 + ch * 8;
        if (m < p.M)
          *reinterpret_cast<uint4*>(Cb + (long long)m * p.ldc + n2) = *reinterpret_cast<const uint4*>(reg2 + row * ROWB + ch * 16);
      }
    }
    return;
  }
  if (p.act == VLA_ACT_SWIGLU) {
    // columns interleaved in 16s: even tiles are gate, odd tiles the matching up columns
    bf16_t* C2 = p.C2 + (long long)z * p.sC2;
#pragma unroll
    for (int pr = 0; pr < C::NT / 2; ++pr)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = wm0 + mi * 16 + lr;
        float h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = rbf(acc[2 * pr][mi][j] * p.alpha + bv[2 * pr][j]);
          const float u = rbf(acc[2 * pr + 1][mi][j] * p.alpha + bv[2 * pr + 1][j]);
          acc[2 * pr][mi][j] = g;
          acc[2 * pr + 1][mi][j] = u;
          h[j] = rbf(g * __builtin_amdgcn_rcpf(1.0f + __expf(-g))) * u;
        }
        const int hc = (wn0 >> 1) + pr * 16 + lq * 4;
        if (m < p.M && hc + 3 < (p.N >> 1)) {
          uint2 o = {pack2(h[0], h[1]), pack2(h[2], h[3])};
          *reinterpret_cast<uint2*>(C2 + (long long)m * p.ldc2 + hc) = o;
        }
      }
    if (p.C == nullptr) return;
  } else {
    // alpha, bias, activation on the bf16-rounded linear output (the reference's Linear emits bf16 before the activation
    // module; the activation's own bf16 rounding is the pack into the staging tile below).  The activation is resolved
    // OUTSIDE the element loops: a per-element switch cost 5 us on the ViT fc1 GEMM even for ReLU.
    float alpha = p.alpha;
    if (p.bias_post) {                        // bf16(bf16(alpha acc) + bias): torch CPU Linear on a strided input
#pragma unroll
      for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ni][mi][j] = rbf(acc[ni][mi][j] * alpha);
      alpha = 1.f;
    }
    auto finish = [&](auto fn) {
#pragma unroll
      for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ni][mi][j] = fn(acc[ni][mi][j] * alpha + bv[ni][j]);
    };
    if (p.act == VLA_ACT_GELU) finish([](float v) { return gelu_erf(rbf(v)); });
    else if (p.act == VLA_ACT_RELU) finish([](float v) { return fmaxf(v, 0.f); });
    else if (p.act == VLA_ACT_GELU_TANH) finish([](float v) { return gelu_tanh(rbf(v)); });
    else finish([](float v) { return v; });
    // ---- fused rotary embedding on the projected q/k columns (saves a full read+write pass per projection)
    if (ROPE != 0 && n0 + cbase(0) < p.rope_cols) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int pos = min(wm0 + mi * 16 + lr, p.M - 1) % p.rope_T;
        if (ROPE == 2) {
          // action_heads.py:125-146: pairs (2i, 2i+1), cos/sin tables of cat([f, f]) (different frequency per lane of a pair)
#pragma unroll
          for (int ni = 0; ni < C::NT; ++ni) {
            const int d = (wn0 + ni * 16 + lq * 4) % p.rope_dh;
            const float4 c = *reinterpret_cast<const float4*>(p.rope_cos + (long long)pos * p.rope_dh + d);
            const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (long long)pos * p.rope_dh + d);
            const float x0 = rbf(acc[ni][mi][0]), x1 = rbf(acc[ni][mi][1]), x2 = rbf(acc[ni][mi][2]), x3 = rbf(acc[ni][mi][3]);
            acc[ni][mi][0] = rbf(x0 * c.x) + rbf(-x1 * sn.x);
            acc[ni][mi][1] = rbf(x1 * c.y) + rbf(x0 * sn.y);
            acc[ni][mi][2] = rbf(x2 * c.z) + rbf(-x3 * sn.z);
            acc[ni][mi][3] = rbf(x3 * c.w) + rbf(x2 * sn.w);
          }
        } else if (ROPE == 1 && C::NT == 4) {
          // HF rotate_half, head dim 64 == this wave's 64 columns: d <-> d+32 are tiles ni and ni+2 of the same lane
          const int half = p.rope_dh >> 1;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int d = ni * 16 + lq * 4;
            const float4 c = *reinterpret_cast<const float4*>(p.rope_cos + (long long)pos * half + d);
            const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (long long)pos * half + d);
            const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = rbf(acc[ni][mi][j]), b = rbf(acc[ni + 2][mi][j]);
              acc[ni][mi][j] = rbf(a * cc[j]) + rbf(-b * ss[j]);
              acc[ni + 2][mi][j] = rbf(b * cc[j]) + rbf(a * ss[j]);
            }
          }
        } else if (ROPE == 1 && C::NT == 2) {
          // same rotation on the permuted 8-wave layout: n tile 0 holds head columns d = 16 (wc & 1) + 4 lq + j, tile 1 d + 32
          const int half = p.rope_dh >> 1, d = (wide ? wc * 16 : (wc & 1) * 16) + lq * 4;
          const float4 c = *reinterpret_cast<const float4*>(p.rope_cos + (long long)pos * half + d);
          const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (long long)pos * half + d);
          const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = rbf(acc[0][mi][j]), b = rbf(acc[1][mi][j]);
            acc[0][mi][j] = rbf(a * cc[j]) + rbf(-b * ss[j]);
            acc[1][mi][j] = rbf(b * cc[j]) + rbf(a * ss[j]);
          }
        }
      }
    }
  }

  // stage the wave's 64 x WTN tile (bf16) through its private LDS region (NHALF passes), then store 16 B per lane
  bf16_t* Cb = p.C + (long long)z * p.sC;
  const bf16_t* Rb = p.R ? p.R + (long long)z * p.sR : nullptr;
  const bool vec_ok = ((p.ldc & 7) == 0) && (!Rb || (p.ldr & 7) == 0);
  constexpr int CH = C::WTN / 8;        // 16-B chunks per staged row
  constexpr int RPP = 64 / CH;          // rows per pass
  constexpr int MIH = 4 / C::NHALF;     // 16-row m tiles per half
#pragma unroll
  for (int half = 0; half < C::NHALF; ++half) {
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
      for (int mh = 0; mh < MIH; ++mh) {
        const int mi = half * MIH + mh;
        uint2 o = {pack2(acc[ni][mi][0], acc[ni][mi][1]), pack2(acc[ni][mi][2], acc[ni][mi][3])};
        *reinterpret_cast<uint2*>(reg + (mh * 16 + lr) * C::EPI_STRIDE + (ni * 16 + lq * 4) * 2) = o;
      }
    // The common case - the wave's rows and columns all inside C, 16-B aligned rows, plain row addressing - without a branch
    // per row segment, the residual segments all requested before the first is added (as in gemm256.hip: in the general loop
    // below every conditional residual load is followed by its own vmcnt(0)).
    {
      constexpr int NIT = C::EPI_ROWS / RPP;
      const int wmh = wm0 + half * C::EPI_ROWS;
      const bool plain_rows = (p.res_mod | p.gR | p.gC | p.c_live_mod) == 0;
      if (plain_rows && vec_ok && wmh + C::EPI_ROWS <= p.M && n0 + cbase(C::NT - 1) + 16 <= p.N) {
        const int row0 = lane / CH, ch = lane % CH;
        const int n = n0 + cbase(ch >> 1) + (ch & 1) * 8;
        if (Rb) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 rv[NIT];
#pragma unroll
          for (int it = 0; it < NIT; ++it) rv[it] = *reinterpret_cast<const u32x4*>(Rb + (long long)(wmh + it * RPP + row0) * p.ldr + n);
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int row = it * RPP + row0;
            const uint4 v = *reinterpret_cast<const uint4*>(reg + row * C::EPI_STRIDE + ch * 16);
            const unsigned a[4] = {v.x, v.y, v.z, v.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              o[k] = pack2(bf2f((bf16_t)(a[k] & 0xffff)) + bf2f((bf16_t)(rv[it][k] & 0xffff)), bf2f((bf16_t)(a[k] >> 16)) + bf2f((bf16_t)(rv[it][k] >> 16)));
            *reinterpret_cast<uint4*>(Cb + (long long)(wmh + row) * p.ldc + n) = uint4{o[0], o[1], o[2], o[3]};
          }
        } else {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int row = it * RPP + row0;
            *reinterpret_cast<uint4*>(Cb + (long long)(wmh + row) * p.ldc + n) = *reinterpret_cast<const uint4*>(reg + row * C::EPI_STRIDE + ch * 16);
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int it = 0; it < C::EPI_ROWS / RPP; ++it) {
      const int row = it * RPP + lane / CH, ch = lane % CH;
      const int m = wm0 + half * C::EPI_ROWS + row, n = n0 + cbase(ch >> 1) + (ch & 1) * 8;
      uint4 v = *reinterpret_cast<const uint4*>(reg + row * C::EPI_STRIDE + ch * 16);
      if (m >= p.M || n >= p.N) continue;
      if (p.c_live_mod > 0 && (m % p.c_live_mod) < p.c_live_from) continue;   // row never read again (live-row backward)
      const long long roff = p.res_mod > 0 ? (long long)(m % p.res_mod) * p.ldr
                             : p.gR > 0 ? (long long)(m / p.gR) * p.sgR + (long long)(m % p.gR) * p.ldr : (long long)m * p.ldr;
      const long long crow = p.gC > 0 ? (long long)(m / p.gC) * p.sgC + (long long)(m % p.gC) * p.ldc : (long long)m * p.ldc;
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
  }
}

// Pick the tile per problem.  Calibrated on MI355X (tools/bench_kernels.py, round 1):
//  * 256x256 (8 waves of 64x128, 1 block/CU) only pays on huge squares (8192^3: 1201 vs 1093 TF/s); 256x128x3-stage
//    is kept as a forced option (VLA_GEMM_TILE=1);
//  * 128x128 with EIGHT waves (2x4 waves of 64x32, 2 blocks/CU = 16 waves/CU): the K-loop is latency/barrier bound
//    (rocprofv3 PMC: SQ_WAIT_ANY 30-40 %, MFMA busy ~41 % with 4 waves), so more resident waves win over operand
//    reuse per wave: +5..+40 % over the 4-wave 64x64 geometry on every hot shape;
//  * 128x64 (4 waves, 3 blocks/CU) only for problems smaller than one round of tiles (the M=256 head GEMMs).
struct TileChoice { int bm, bn; };
// When the 256 x 256 8-phase kernel (one workgroup per CU) is chosen automatically: problems with at least a chip-full of
// its tiles' worth of work in both dimensions.  VLA_GEMM_TILE=6 forces it, VLA_NO_GEMM256 disables it.
inline bool use_256(int M, int N, int K, int batch, int act) {
  static const bool off = getenv("VLA_NO_GEMM256") != nullptr;
  if (off) return false;
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch;
  // the SwiGLU-backward epilogue streams GU in and dGU out (4 N bytes per row against 2 K of operands: as long as a short K
  // loop, HBM-bound on the full sequence): the 128-row kernel's second resident workgroup overlaps it with the other one's
  // K loop (live rows, M = 2048: 38 vs 44 us; full sequence, M = 11264: 166 vs 191 us)
  if (act == VLA_ACT_SWIGLU_BWD) return false;
  // a GELU epilogue (10 us of VALU per round of tiles) with a thin tail round (ViT fc1: 544 tiles = two rounds + 32) is the one
  // large shape the 128-row kernel still wins isolated (96 vs 106-114 us: its second resident workgroup computes under the
  // first one's epilogue); on the step the two routings tie (25.25-25.34 vs 25.29-25.32 ms)
  const int ncu = vla_num_cus();
  if ((act == VLA_ACT_GELU || act == VLA_ACT_GELU_TANH) && tiles > ncu && (tiles % ncu) != 0 && (tiles % ncu) * 4 < ncu) return false;
  // Rounds model (round 3, tools/bench_tiles_b16.py): a 256 x 256 tile is four 128 x 128 tiles of work on one CU; the 128-row kernel
  // keeps two workgroups per CU and reaches ~0.87 of the 256-row kernel's per-CU rate.  Cost in units of "one 128 x 128 tile at
  // the 256-row kernel's rate":  rounds256 x 4  against  rounds128 x 2 / 0.87.  Reproduces every measured winner of the batch-32
  // step (gate/up, down, o, q|k|v, ViT qkv / proj / fc2, task K/V -> 256) and of the batch-16 shapes of the LoRA / full steps, where
  // the round-2 threshold (>= 96 tiles) sent two shapes the wrong way: LLM q|k|v 5632 x 1152 (110 tiles = 0.43 round: 25.2 vs
  // 17.3 us) and ViT fc1 / dX fc2 4096 x 4352 (272 tiles = 1.06 rounds: 58.3 vs 47.8 us).
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  const double est256 = (double)((tiles + ncu - 1) / ncu) * 4.0, est128 = (double)((t128 + 2 * ncu - 1) / (2 * ncu)) * (2.0 / 0.87);
  return M >= 1024 && N >= 768 && K >= 256 && est256 < est128;
}
inline TileChoice choose_tile(int M, int N, int K, int force, int rope_mode, int batch = 1, int split = 1, int act = 0) {
  // rotate_half RoPE is fused in both kernels (bit-identical).  The LLM's q|k|v projection goes to the 256-row kernel when it
  // fills most of the chip with its tiles (whole-batch launch, M = 11264: 220 tiles, 26.64-26.69 vs 26.84-26.85 ms on the step,
  // same box); the half-batch launches of the two-pipeline forward (110 tiles) measured 26.10-26.20 vs 26.07-26.10 and stay
  // on the 128-row kernel.  VLA_NO_ROPE256 switches it off.
  static const bool no_rope256 = getenv("VLA_NO_ROPE256") != nullptr;
  if (split == 1) {     // (both RoPE conventions are fused in both kernels)
    if (force == 6) return {256, 257};   // 256 x 256 two-phase kernel (gemm256.hip)
    const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch;
    if (force == 0 && (rope_mode == 0 || (!no_rope256 && t256 >= 192)) && use_256(M, N, K, batch, act)) return {256, 257};
  }
  if (rope_mode == 1) return {128, 128};   // rotate_half: 8 waves, each owning 16 columns of both halves of one head
  if (force == 2) return {128, 128};
  if (force == 3) return {128, 64};
  // In situ (whole training step, same-box A/B) the 8-wave 128x128 geometry beats the narrow tile on every shape of
  // the step, including the M=256 head GEMMs that overlap the LLM on the side stream (48.5 vs 49.5 vs 52.0 ms/step
  // for always-square / mixed / always-narrow); the narrow tile stays available through VLA_GEMM_TILE=3.
  // Exception: long-K problems that fill less than half the chip with square tiles (the live-row gate/up dX GEMM:
  // M 2048, N 896, K 9728 -> 112 tiles): the narrow tile doubles the number of K loops in flight.
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (tiles <= 128 && K >= 4096 && N % 64 == 0) return TileChoice{128, 64};
  return TileChoice{128, 128};
}

// split-K second pass: C = bf16(bf16(act(bf16(sum_z ws[z] * alpha + bias))) + R); 4 columns per thread
__global__ void splitk_finalize_kernel(const float* __restrict__ ws, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R,
                                       bf16_t* __restrict__ C, int M, int N, int ldc, int ldr, int act, float alpha, int split, int vec) {
  const long long total = (long long)M * N / 4, plane = (long long)M * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i * 4 / N), n = (int)(i * 4 - (long long)m * N);
    // every plane's segment (and the bias / residual segments) is requested before the first one is used: summed in a rolled
    // loop, each plane paid its own dependent round trip (13 us in situ for 29 MB on the live-row backward's chain)
    f32x4 pv[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) pv[z] = z < split ? *reinterpret_cast<const f32x4*>(ws + z * plane + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // (vec: bias / R / C rows are 8-B aligned - host-checked; otherwise element accesses)
    uint2 braw = {0, 0}, rraw = {0, 0};
    if (vec) {
      if (bias) braw = *reinterpret_cast<const uint2*>(bias + n);
      if (R) rraw = *reinterpret_cast<const uint2*>(R + (long long)m * ldr + n);
    } else {
      if (bias) braw = uint2{(unsigned)bias[n] | ((unsigned)bias[n + 1] << 16), (unsigned)bias[n + 2] | ((unsigned)bias[n + 3] << 16)};
      if (R) {
        const bf16_t* rp = R + (long long)m * ldr + n;
        rraw = uint2{(unsigned)rp[0] | ((unsigned)rp[1] << 16), (unsigned)rp[2] | ((unsigned)rp[3] << 16)};
      }
    }
    f32x4 a = pv[0];
#pragma unroll
    for (int z = 1; z < 8; ++z) a += pv[z];                   // (same order as the rolled sum; absent planes add +0)
    for (int z = 8; z < split; ++z) a += *reinterpret_cast<const f32x4*>(ws + z * plane + i * 4);
    const float bb[4] = {bf2f((bf16_t)(braw.x & 0xffff)), bf2f((bf16_t)(braw.x >> 16)), bf2f((bf16_t)(braw.y & 0xffff)), bf2f((bf16_t)(braw.y >> 16))};
    const float rr[4] = {bf2f((bf16_t)(rraw.x & 0xffff)), bf2f((bf16_t)(rraw.x >> 16)), bf2f((bf16_t)(rraw.y & 0xffff)), bf2f((bf16_t)(rraw.y >> 16))};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = a[j] * alpha + (bias ? bb[j] : 0.f);
      if (act != VLA_ACT_NONE) v = apply_act(rbf(v), act);
      if (R) v = rbf(v) + rr[j];
      o[j] = v;
    }
    if (vec) {
      *reinterpret_cast<uint2*>(C + (long long)m * ldc + n) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) C[(long long)m * ldc + n + j] = f2bf(o[j]);
    }
  }
}

template <int BM, int BN, int STAGES, int ROPE, int WN = 2, bool F8 = false, bool EXT = false>
int launch(const GemmP& p0, int M, int N, int batch, hipStream_t st) {
  using C = Cfg<BM, BN, STAGES, WN>;
  GemmP p = p0;
  p.tiles_n = (N + BN - 1) / BN;
  p.ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, STAGES, ROPE, WN, F8, EXT>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, STAGES, ROPE, WN, F8, EXT>), dim3(p.ntiles, 1, batch), dim3(C::NTHREADS), C::LDS_BYTES, st, p);
  return 0;
}

}  // namespace

// gemm256_kernel keeps its per-lane operand addresses as 32-bit BYTE offsets from a wave-uniform base (the 128-row kernel uses
// 64-bit pointers): every operand row a 256-row tile can touch - tiles round M and N up to 256, row groups included - must
// start below 4 GiB minus one row.  Pure host arithmetic (no device query): exported so that the routing is unit-testable.
extern "C" int vla_gemm256_extent_ok(const vla_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  const unsigned long long EB = d->fp8 ? 1 : 2, lim = 1ull << 32;
  const unsigned long long ra = (unsigned long long)((d->M + 255) / 256) * 256 - 1, rb = (unsigned long long)((d->N + 255) / 256) * 256 - 1;
  const unsigned long long offA = d->a_group > 0 ? (ra / d->a_group) * (unsigned long long)d->a_group_stride + (ra % d->a_group) * (unsigned long long)d->lda
                                                 : ra * (unsigned long long)d->lda;
  const unsigned long long offB = rb * (unsigned long long)d->ldb;
  return (offA + d->K) * EB < lim && (offB + d->K) * EB < lim;
}

// Caller's hint (process-wide, read at launch = at graph capture): the following products run on an otherwise idle chip and are bound
// by latency, not throughput (batch-1 predict_action).  Selects kernels only - results are bit-identical either way.
static std::atomic<int> g_latency_hint{0};
extern "C" int vla_gemm_latency_hint(int on) {
  return on < 0 ? g_latency_hint.load(std::memory_order_relaxed) : g_latency_hint.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

// The tile choice of vla_gemm_bf16_nt for a descriptor (shared with the predicate below).
static TileChoice route(const vla_gemm_desc* d) {
  const int split = d->split_k > 1 ? d->split_k : 1;
  const char* e = getenv("VLA_GEMM_TILE");     // 0/unset auto, 2: 128x128, 3: 128x64, 6: 256x256  (test / benchmarking aid)
  TileChoice tc = choose_tile(d->M, d->N, d->K / split, e ? atoi(e) : 0, d->rope_mode, split > 1 ? split : d->batch, split, d->act);
  if (tc.bm == 256 && tc.bn == 257 && !vla_gemm256_extent_ok(d)) tc = TileChoice{128, 128};   // operands of 4 GiB and more: 64-bit-pointer kernel
  // interleaved RoPE on the 256-row kernel: the plain epilogue without residual (the head's K|V projections) - anything else keeps gemm.hip's
  if (tc.bm == 256 && tc.bn == 257 && d->rope_mode == 2 && (d->R || d->act != VLA_ACT_NONE || d->fp8)) tc = TileChoice{128, 128};
  // rotate_half at head dim 128: the 128-row kernel's column map (one head per 128-column tile); the 256-row kernel's is built on 64-wide heads
  if (tc.bm == 256 && tc.bn == 257 && d->rope_mode == 1 && d->rope_dh != 64) tc = TileChoice{128, 128};
  return tc;
}

// K extension on the 256 x 256 kernel (round 4; bf16 operands): the routing of the plain product over K + K2, provided the extension's rows
// are addressable like the main operands' (32-bit per-lane byte offsets, no row groups on A) and the epilogue is one the kernel has
// (plain / activation / residual / rotate_half at head dim 64 / SwiGLU forward).
static bool ext_on_256(const vla_gemm_desc* d) {
  if (d->K2 <= 0 || d->fp8 || d->batch != 1 || d->split_k > 1 || d->act == VLA_ACT_SWIGLU_BWD || d->rope_mode == 2 || d->a_group != 0 ||
      (d->rope_mode == 1 && d->rope_dh != 64) || !vla_gemm256_extent_ok(d))
    return false;
  const unsigned long long ra = (unsigned long long)((d->M + 255) / 256) * 256, rb = (unsigned long long)((d->N + 255) / 256) * 256;
  if ((ra * d->lda2 + d->K2) * 2 >= (1ull << 32) || (rb * d->ldb2 + d->K2) * 2 >= (1ull << 32)) return false;
  const char* e = getenv("VLA_GEMM_TILE");
  const int force = e ? atoi(e) : 0;
  if (force == 6) return true;
  if (force != 0) return false;
  const long long t256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
  static const bool no_rope256 = getenv("VLA_NO_ROPE256") != nullptr;
  if (d->rope_mode == 1 && (no_rope256 || t256 < 192)) return false;
  return use_256(d->M, d->N, d->K + d->K2, 1, d->act);
}

extern "C" int vla_gemm_uses_256(const vla_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->fp8 || d->split_k > 1) return 0;
  if (d->K2 > 0) return ext_on_256(d) ? 1 : 0;
  const TileChoice tc = route(d);
  return tc.bm == 256 && tc.bn == 257 ? 1 : 0;
}

// second pass of a split-K product: the `split` fp32 planes of d->ws summed in order, then alpha / bias / activation / residual / rounding
static int splitk_finalize(const vla_gemm_desc* d, int split, hipStream_t st) {
  const long long total = (long long)d->M * d->N / 4;
  const unsigned nblk = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(nblk), dim3(256), 0, st, (const float*)d->ws, (const bf16_t*)d->bias,
                     (const bf16_t*)d->R, (bf16_t*)d->C, d->M, d->N, d->ldc, d->ldr, d->act, d->alpha == 0.f ? 1.f : d->alpha, split,
                     (d->ldc % 4 == 0 && ((uintptr_t)d->C & 7) == 0 && (!d->R || (d->ldr % 4 == 0 && ((uintptr_t)d->R & 7) == 0)) &&
                      (!d->bias || ((uintptr_t)d->bias & 7) == 0)) ? 1 : 0);
  VLA_CHECK_LAUNCH("gemm_splitk_finalize");
  return VLA_OK;
}

extern "C" int vla_gemm_bf16_nt(void* stream, const vla_gemm_desc* d) {
  VLA_REQUIRE(d && d->A && d->B, "gemm: null operand");
  VLA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0, "gemm: empty problem");
  VLA_REQUIRE(d->K % BK == 0, "gemm: K must be a multiple of 64 (pad the operands)");
  VLA_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements (16-B rows)");
  if (d->fp8)
    VLA_REQUIRE(d->fp8 == 1 && d->a_scale && d->b_scale && d->K % 128 == 0 && d->lda % 16 == 0 && d->ldb % 16 == 0 && d->batch == 1 &&
                    d->split_k <= 1 && d->rope_mode != 2 && d->a_group == 0 && (d->act != VLA_ACT_SWIGLU_BWD || d->K2 > 0),
                "gemm: fp8 needs scales, K % 128 == 0, 16-B rows, batch 1, no split-K / interleaved RoPE / row groups on A (SwiGLU backward: with a K extension only)");
  VLA_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "gemm: A/B must be 16-B aligned");
  VLA_REQUIRE(d->sA % 8 == 0 && d->sB % 8 == 0, "gemm: batch strides of A/B must keep 16-B alignment");
  if (d->act == VLA_ACT_SWIGLU_BWD) {
    VLA_REQUIRE(d->C && d->R && !d->bias && d->rope_mode == 0 && d->N % 64 == 0 && d->ldc % 8 == 0 && d->ldr % 4 == 0 &&
                    ((uintptr_t)d->R & 7) == 0 && d->c_group == 0 && d->res_mod == 0,
                "gemm: swiglu_bwd needs C = dGU [M, 2N] (ldc%8), R = GU [M, 2N] (interleaved), N%64 == 0");
  } else if (d->act == VLA_ACT_SWIGLU) {
    VLA_REQUIRE(d->C2 && d->N % 32 == 0 && d->ldc2 % 4 == 0 && ((uintptr_t)d->C2 & 7) == 0 && d->sC2 % 4 == 0,
                "gemm: swiglu needs C2, N%32==0, ldc2%4==0");
    VLA_REQUIRE(!d->R, "gemm: swiglu epilogue takes no residual");
  } else {
    VLA_REQUIRE(d->C, "gemm: null C");
  }
  if (d->C) VLA_REQUIRE(((uintptr_t)d->C & 15) == 0 || (d->ldc % 8) != 0, "gemm: C must be 16-B aligned");
  if (d->C && d->ldc % 8 == 0) VLA_REQUIRE(d->sC % 8 == 0, "gemm: sC must keep 16-B alignment");
  if (d->R && d->ldc % 8 == 0 && d->ldr % 8 == 0 && d->act != VLA_ACT_SWIGLU_BWD)
    VLA_REQUIRE(((uintptr_t)d->R & 15) == 0 && d->sR % 8 == 0, "gemm: R must be 16-B aligned");
  VLA_REQUIRE(d->a_group >= 0 && d->c_group >= 0 && d->a_group_stride % 8 == 0 &&
                  (d->c_group == 0 || d->ldc % 8 != 0 || d->c_group_stride % 8 == 0),
              "gemm: row-group strides must keep 16-B alignment");
  GemmP p;
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = (bf16_t*)d->C;
  p.bias = (const bf16_t*)d->bias; p.R = (const bf16_t*)d->R; p.C2 = (bf16_t*)d->C2;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.ldc2 = d->ldc2; p.res_mod = d->res_mod; p.act = d->act;
  p.sA = d->sA; p.sB = d->sB; p.sC = d->sC; p.sR = d->sR; p.sC2 = d->sC2; p.sBias = d->sBias;
  p.tiles_n = p.ntiles = 0; p.batch = 1; p.stagger = 0;
  p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  p.gA = d->a_group; p.sgA = d->a_group_stride; p.gC = d->c_group; p.sgC = d->c_group_stride;
  p.gR = d->r_group; p.sgR = d->r_group_stride;
  p.c_live_mod = d->c_live_mod; p.c_live_from = d->c_live_from;
  int split = d->split_k > 1 ? d->split_k : 1;     // (split-K below may lower it: uneven slices)
  p.bias_post = d->bias_post_round;
  VLA_REQUIRE(d->bias_post_round == 0 || (d->bias_post_round == 1 && d->bias && d->rope_mode != 1 && split == 1 && d->act == VLA_ACT_NONE),
              "gemm: bias_post_round needs a bias and a plain epilogue (no rotate_half rope / split-K / activation)");
  p.ws = nullptr; p.Ktot = 0;
  if (split > 1) {
    VLA_REQUIRE(d->ws && d->batch == 1 && d->rope_mode == 0 && d->c_group == 0 && d->r_group == 0 &&
                    d->res_mod == 0 && d->c_live_mod == 0 && d->C &&
                    (d->act == VLA_ACT_NONE || d->act == VLA_ACT_GELU || d->act == VLA_ACT_RELU || d->act == VLA_ACT_GELU_TANH),
                "gemm: split_k needs an fp32 workspace [split_k, M, N], batch 1 and a plain epilogue");
    VLA_REQUIRE(d->N % 4 == 0 && ((uintptr_t)d->ws & 15) == 0, "gemm: split_k needs N % 4 == 0 and a 16-B aligned workspace");
    // slices of ceil(K-tiles / split_k) K-tiles; when that does not divide, the last slice is shorter and fewer slices may be needed
    const int kt = d->K / BK, per = (kt + split - 1) / split;
    split = (kt + per - 1) / per;
    p.ws = d->ws;
    p.K = per * BK;                // every z slice owns `per` K-tiles of the contraction (the last one what is left of Ktot)
    p.Ktot = d->K;
    p.sA = p.sB = p.K;             // ... starting that many elements further along the rows of A and B
    p.bias = nullptr; p.R = nullptr;
  }
  VLA_REQUIRE(d->c_live_mod >= 0 && d->c_live_from >= 0 && (d->c_live_mod == 0 || d->c_live_from < d->c_live_mod),
              "gemm: c_live_from must lie inside c_live_mod");
  VLA_REQUIRE(d->r_group >= 0 && d->r_group_stride % 8 == 0 && (d->r_group == 0 || d->res_mod == 0),
              "gemm: r_group stride must keep 16-B alignment; r_group and res_mod are exclusive");
  p.rope_mode = d->rope_mode; p.rope_T = d->rope_T; p.rope_dh = d->rope_dh; p.rope_cols = d->rope_cols;
  p.rope_cos = d->rope_cos; p.rope_sin = d->rope_sin;
  p.scaleA = d->fp8 ? d->a_scale : nullptr; p.scaleB = d->fp8 ? d->b_scale : nullptr;
  p.A2 = (const bf16_t*)d->A2; p.B2 = (const bf16_t*)d->B2; p.K2 = d->K2; p.lda2 = d->lda2; p.ldb2 = d->ldb2;
  if (d->K2 != 0)
    VLA_REQUIRE(d->K2 > 0 && d->K2 % BK == 0 && d->A2 && d->B2 && d->lda2 % 8 == 0 && d->ldb2 % 8 == 0 && (((uintptr_t)d->A2 | (uintptr_t)d->B2) & 15) == 0 &&
                    d->batch == 1 && split == 1 && d->rope_mode != 2,
                "gemm: the K extension needs A2 / B2 (bf16, 16-B aligned rows, K2 % 64 == 0), batch 1, no split-K / interleaved RoPE");
  if (d->rope_mode != 0) {
    VLA_REQUIRE(d->rope_mode == 1 || d->rope_mode == 2, "gemm: rope_mode 0/1/2");
    VLA_REQUIRE(d->rope_cos && d->rope_sin && d->rope_T > 0 && d->rope_dh > 0 && d->rope_dh % 4 == 0 && d->rope_cols % 64 == 0 &&
                    (((uintptr_t)d->rope_cos | (uintptr_t)d->rope_sin) & 15) == 0 && d->act != VLA_ACT_SWIGLU,
                "gemm: bad rope arguments");
    if (d->rope_mode == 1) VLA_REQUIRE(d->rope_dh == 64 || (d->rope_dh == 128 && d->N % 128 == 0 && d->rope_cols % 128 == 0),
                                       "gemm: fused rotate_half RoPE needs head dim 64, or 128 with N and rope_cols multiples of 128");
  }
  const char* e = getenv("VLA_GEMM_TILE");
  {   // small-output products (the LoRA t / dt products; the action head's Linears on 8 x B rows; the batch-1 pass): gemm_skinny.hip
    const bool simple = d->batch == 1 && split == 1 && d->a_group == 0 && d->c_group == 0 && d->r_group == 0 && d->res_mod == 0 && d->c_live_mod == 0 &&
                        !d->fp8 && d->K2 == 0 && d->C && !(e && atoi(e) != 0);
    if (vla_gemm_skinny_try(p, simple, g_latency_hint.load(std::memory_order_relaxed) > 0, (hipStream_t)stream)) {
      VLA_CHECK_LAUNCH("gemm_bf16_nt(skinny)");
      return VLA_OK;
    }
  }
  const TileChoice tc = route(d);
  const bool fits256 = vla_gemm256_extent_ok(d) != 0;
  if (d->K2 > 0) {                 // K extension: the 128-row kernel, or - bf16, chip-filling shapes - the 256-row kernel's EXT instantiations
    hipStream_t sx = (hipStream_t)stream;
    if (d->fp8) {                  // e4m3 base operands, bf16 extension: the scales meet the accumulator between the two contractions
      if (d->act == VLA_ACT_SWIGLU_BWD) launch<128, 128, 2, 3, 4, true, true>(p, d->M, d->N, 1, sx);
      else if (d->rope_mode == 1) launch<128, 128, 2, 1, 4, true, true>(p, d->M, d->N, 1, sx);
      else launch<128, 128, 2, 0, 4, true, true>(p, d->M, d->N, 1, sx);
    } else if (ext_on_256(d)) vla_gemm256_launch(p, d->act == VLA_ACT_SWIGLU ? 1 : 0, 1, sx);     // (p.K2 > 0: the EXT instantiations)
    else if (d->act == VLA_ACT_SWIGLU_BWD) launch<128, 128, 2, 3, 4, false, true>(p, d->M, d->N, 1, sx);
    else if (d->rope_mode == 1) launch<128, 128, 2, 1, 4, false, true>(p, d->M, d->N, 1, sx);
    else launch<128, 128, 2, 0, 4, false, true>(p, d->M, d->N, 1, sx);
    VLA_CHECK_LAUNCH("gemm_bf16_nt(ext)");
    return VLA_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  if (d->fp8) {
    // e4m3 operands.  With the K loop halved, the 256 x 256 kernel's per-tile costs weigh twice as much: on the step's shapes the
    // 128-row kernel (two workgroups per CU, one's epilogue under the other's K loop) is as fast or faster (gate/up 162 vs 162 us,
    // down 62 vs 70, ViT fc1 63 vs 87), on big squares the 256-row kernel wins (8192^3: 2503 vs 2108 TF/s) - it takes those.
    const int force = e ? atoi(e) : 0;
    if ((force == 6 || (force == 0 && d->M >= 4096 && d->N >= 4096 && d->K >= 4096)) && d->c_group == 0 && d->r_group == 0 && fits256 &&
        !(d->rope_mode == 1 && d->rope_dh != 64))
      vla_gemm256_launch(p, d->act == VLA_ACT_SWIGLU ? 1 : 0, 1, st);
    else if (d->rope_mode == 1) launch<128, 128, 2, 1, 4, true>(p, d->M, d->N, 1, st);
    else launch<128, 128, 2, 0, 4, true>(p, d->M, d->N, 1, st);
    VLA_CHECK_LAUNCH("gemm_fp8_nt");
    return VLA_OK;
  }
  if (tc.bm == 256 && tc.bn == 257) {          // 256 x 256 staggered 8-phase kernel (gemm256.hip)
    const int epi = d->act == VLA_ACT_SWIGLU ? 1 : d->act == VLA_ACT_SWIGLU_BWD ? 2 : 0;
    // (a column peel of the tail round - the last column tiles on the 128-row kernel behind this launch - turned ViT fc1's 105 us into
    //  ~80 isolated and cost 0.1 ms on the step: the other streams already fill the tail round.  tools/diag/gemm256_pruned_paths.patch)
    vla_gemm256_launch(p, epi, d->batch, st);
  } else if (d->act == VLA_ACT_SWIGLU_BWD) launch<128, 128, 2, 3, 4>(p, d->M, d->N, d->batch, st);
  else {
    // Launches of at most one workgroup per CU (batch-1 inference: every product; the training step: the action head's x-chain) are
    // bound by the LATENCY of a K-tile, not by bandwidth: with the two-stage ring one K-tile is in flight per workgroup and every tile
    // costs a full trip to HBM (~1.2 us against 0.2 us of MFMAs: 17-22 us for ANY product of the batch-1 pass, K = 896-1152, whatever its
    // size).  They run on a four-stage ring (128 KiB of LDS: the CU is theirs anyway), three K-tiles in flight - or, when even 64-row tiles
    // leave CUs idle, on 64 x 128 tiles with a six-stage ring (144 KiB, five in flight, twice the workgroups: what such a launch can keep
    // in flight is workgroups x ring bytes) -, the same K order and MFMA sequence: bit-identical (test_gemm_deep_ring_bit_identical).
    // Only under vla_gemm_latency_hint(1): in the training step the same launches share the chip with two other streams' kernels, and a 128-KiB workgroup waits for a whole CU's LDS where the 64-KiB one
    // slips in beside another (step 24.13 -> 24.44 ms same box with the deep ring on every sub-round launch).  VLA_NO_DEEP_RING=1: two
    // stages everywhere (A/B aid).
    const int nb = split > 1 ? split : d->batch;
    const long long wgs = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128) * nb, wgs64 = (long long)((d->M + 63) / 64) * ((d->N + 127) / 128) * nb;
    const bool lat = g_latency_hint.load(std::memory_order_relaxed) > 0 && tc.bn == 128 && p.K / BK >= 4 && !getenv("VLA_NO_DEEP_RING");
    const bool deep64 = lat && wgs64 <= vla_num_cus();            // 64-row tiles, six stages (144 KiB): twice the workgroups, five K-tiles in flight
    const bool deep = lat && !deep64 && wgs <= vla_num_cus();     // 128-row tiles, four stages (128 KiB)
    if (d->rope_mode == 1) {               // 8 waves, rotation pairs inside a lane
      if (deep64) launch<64, 128, 6, 1, 4>(p, d->M, d->N, nb, st);
      else if (deep) launch<128, 128, 4, 1, 4>(p, d->M, d->N, nb, st);
      else launch<128, 128, 2, 1, 4>(p, d->M, d->N, nb, st);
    } else if (d->rope_mode == 2) {
      if (deep64) launch<64, 128, 6, 2, 4>(p, d->M, d->N, nb, st);
      else if (deep) launch<128, 128, 4, 2, 4>(p, d->M, d->N, nb, st);
      else if (tc.bn == 128) launch<128, 128, 2, 2, 4>(p, d->M, d->N, nb, st);
      else launch<128, 64, 2, 2>(p, d->M, d->N, nb, st);
    } else if (deep64) launch<64, 128, 6, 0, 4>(p, d->M, d->N, nb, st);
    else if (deep) launch<128, 128, 4, 0, 4>(p, d->M, d->N, nb, st);
    else if (tc.bn == 128) launch<128, 128, 2, 0, 4>(p, d->M, d->N, nb, st);   // 8 waves (2x4) of 64x32
    else launch<128, 64, 2, 0>(p, d->M, d->N, nb, st);
  }
  VLA_CHECK_LAUNCH("gemm_bf16_nt");
  if (split > 1) return splitk_finalize(d, split, st);
  return VLA_OK;
}
