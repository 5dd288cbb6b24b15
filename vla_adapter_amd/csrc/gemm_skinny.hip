// Tall-skinny NT product for gfx950:  C[M, N] = bf16(alpha * A[M, K] . B[N, K]^T)  with N = 64 / 128 / 192 and M in the thousands.
//
// Replaces, for the LoRA fine-tune (vla-scripts/finetune.py:832-844, peft Linear.forward / its autograd), the two low-rank
// products that sit on every wrapped Linear's forward / dX chain:  t = 2 x A_cat^T  (lora_A of all pairs that share the input)
// and  dt = 2 dy B_blk  (the transpose of lora_B).  On the 128 x 128 tiles of gemm.hip such a product is 32-44 workgroups, each
// pulling (128 + 128) K-rows through ONE CU's L1 (~50 GB/s per CU whatever the ring depth): 19-20 us per
// launch, 280 launches per batch-16 step.
//
// Here a workgroup owns a (16 MT) x (16 NT) output tile - 16 x 64 ... 64 x 96, picked per shape so that every CU gets one workgroup and
// pulls as few operand rows as possible - and its four waves split the contraction four ways (wave w: k in [w K/4, (w+1) K/4) of the
// workgroup's K range); they read their operand fragments straight from global memory into registers in the MFMA layout (lane l: row
// l % 16, eight consecutive k at 8 (l / 16): 16 B per lane, 64 B per row and k-step) - no LDS staging, no barrier in the loop, U k-steps
// of loads in flight per wave.  The four partial accumulators meet in LDS and are summed in wave order (deterministic), scaled, rounded
// once.
//
// Rounding: fp32 accumulation as four K-quarter partial sums added in order 0..3, then alpha, then ONE bf16 rounding - the arithmetic
// of vla_gemm_bf16_nt with split_k = 4 (bit-identical to it when K % 256 == 0: test_gemm_skinny).
#include "common.h"
#include "gemm_params.h"

namespace {

template <int MT, int NT, int U>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                          int M, int K, int lda, int ldb, int ldc, float alpha) {
  constexpr int TM = 16 * MT, TN = 16 * NT, LDP = TN + 4;          // partial rows padded by 16 B
  extern __shared__ __attribute__((aligned(16))) float part[];      // [4 waves][TM rows][LDP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const int kq = K >> 2, nks = kq >> 5;                             // this wave's K range: nks k-steps of 32
  const int k0 = wid * kq + lq * 8;
  const bf16_t* ap[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) ap[i] = A + (long long)min(m0 + 16 * i + lr, M - 1) * lda + k0;
  const bf16_t* bp = B + (long long)(n0 + lr) * ldb + k0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[U][MT], fb[U][NT];
  auto load = [&](int u, int ks) {
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[u][i] = *reinterpret_cast<const bf16x8*>(ap[i] + ks * 32);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[u][j] = *reinterpret_cast<const bf16x8*>(bp + (long long)j * 16 * ldb + ks * 32);
  };
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < nks) load(u, u);
  for (int ks = 0; ks < nks; ks += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ks + u < nks) {
        // operands swapped as in gemm.hip (first operand := B rows): lane owns row m = 16 i + lr and four consecutive n
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][j], fa[u][i], acc[i][j], 0, 0, 0);
        if (ks + u + U < nks) load(u, ks + u + U);
      }
    }
  }
  float* pw = part + wid * TM * LDP;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(pw + (16 * i + lr) * LDP + 16 * j + 4 * lq) = acc[i][j];
  __syncthreads();
  constexpr int CPR = TN / 8;                                        // 16-byte output chunks per row
  for (int c = tid; c < TM * CPR; c += 256) {
    const int row = c / CPR, col = (c - row * CPR) * 8;
    if (m0 + row >= M) continue;
    const float* s = part + row * LDP + col;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      a0 += *reinterpret_cast<const f32x4*>(s + w * TM * LDP);
      a1 += *reinterpret_cast<const f32x4*>(s + w * TM * LDP + 4);
    }
    const uint4 o = {pack2(a0[0] * alpha, a0[1] * alpha), pack2(a0[2] * alpha, a0[3] * alpha), pack2(a1[0] * alpha, a1[1] * alpha),
                     pack2(a1[2] * alpha, a1[3] * alpha)};
    *reinterpret_cast<uint4*>(C + (long long)(m0 + row) * ldc + n0 + col) = o;
  }
}

template <int MT, int NT, int U>
void launch_skinny(const GemmP& p, hipStream_t st) {
  constexpr int LDS = 4 * 16 * MT * (16 * NT + 4) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<MT, NT, U>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, U>), dim3((p.M + 16 * MT - 1) / (16 * MT), p.N / (16 * NT)), dim3(256), LDS, st, p.A, p.B, p.C, p.M, p.K,
                     p.lda, p.ldb, p.ldc, p.alpha);
}

}  // namespace

// Host-side predicate + launch (called from vla_gemm_bf16_nt after its argument checks): 1 = launched here, 0 = not this kernel's shape.
// Contractions of 2048 and more stay on the 128-row tiles with split-K (measured, tools/diag/bench_skinny.py: with K cut into slices here
// as well, 64-row tiles pull twice the operand rows per output and lose - dt of gate/up, K = 9728: 57.7 vs 32.1 us; t of down 22.8 vs 20.0).
int vla_gemm_skinny_try(const GemmP& p, int batch, int split, bool plain_epilogue, hipStream_t st) {
  if (getenv("VLA_NO_SKINNY") || !plain_epilogue || batch != 1 || split != 1) return 0;
  if (!(p.N == 64 || p.N == 128 || p.N == 192) || p.M < 1024 || p.K % 128 != 0 || p.K >= 2048 || p.ldc % 8 != 0 || ((uintptr_t)p.C & 15) != 0) return 0;
  // Tile: what a CU has to pull through its L1 is (tile rows + tile columns) x K operand rows per workgroup, times the workgroups it gets -
  // minimised over the instantiated tiles (every CU busy, as few rows each as possible)
  static const int TL[4][2] = {{1, 4}, {2, 4}, {4, 4}, {4, 6}};
  const int ncu = vla_num_cus();
  int best = -1;
  long long best_cost = 0;
  for (int t = 0; t < 4; ++t) {
    const int tm = 16 * TL[t][0], tn = 16 * TL[t][1];
    if (p.N % tn != 0) continue;
    const long long wgs = (long long)((p.M + tm - 1) / tm) * (p.N / tn), cost = (long long)(tm + tn) * ((wgs + ncu - 1) / ncu);
    if (best < 0 || cost < best_cost) best = t, best_cost = cost;
  }
  switch (best) {
    case 0: launch_skinny<1, 4, 4>(p, st); break;
    case 1: launch_skinny<2, 4, 4>(p, st); break;
    case 2: launch_skinny<4, 4, 3>(p, st); break;
    default: launch_skinny<4, 6, 2>(p, st); break;
  }
  return 1;
}
