// Small-output NT products for gfx950:  C[M, N] = epilogue(alpha * A[M, K] . B[N, K]^T)  whose OUTPUT is too small to give every CU a
// 128 x 128 tile: (a) tall-skinny, N = 64 / 128 / 192 with M in the thousands; (b) short, M <= 512 (any N % 16 == 0).
//
// (a) replaces, for the LoRA fine-tune (vla-scripts/finetune.py:832-844, peft Linear.forward / its autograd), the two low-rank
// products on every wrapped Linear's forward / dX chain:  t = 2 x A_cat^T  (lora_A of all pairs that share the input) and
// dt = 2 dy B_blk  (the transpose of lora_B); (b) the action head's per-block Linears on its 8 x B query rows (action_heads.py:337-410:
// q | k | v, o_proj, ffn and their dX products) and the batch-1 pass's products (modeling_prismatic.py:892-972).  On the 128 x 128
// tiles of gemm.hip such a product is 7-44 workgroups, each pulling (128 + 128) K-rows through ONE CU's L1 (~50 GB/s per CU whatever
// the ring depth): ~20 us per launch in the step, 120 (head) + 280 (LoRA) launches per step.
//
// Here a workgroup owns a (16 MT) x (16 NT) output tile - 16 x 16 ... 64 x 96, picked per shape so that every CU gets a workgroup and
// pulls as few operand rows as possible - and its four waves split the contraction four ways (wave w: k in [w K/4, (w+1) K/4)); they read
// their operand fragments straight from global memory into registers in the MFMA layout (lane l: row l % 16, eight consecutive k at
// 8 (l / 16): 16 B per lane, 64 B per row and k-step) - no LDS staging, no barrier in the loop, U k-steps of loads in flight per wave.
// The four partial accumulators meet in LDS and are summed in wave order (deterministic); the epilogue runs on 8-column row chunks.
//
// Rounding: fp32 accumulation as four K-quarter partial sums added in order 0..3, then gemm.hip's epilogue at gemm.hip's rounding points
// (alpha, bias, activation on the bf16-rounded value, rotate_half or interleaved RoPE, bf16, + residual, bf16): the arithmetic of vla_gemm_bf16_nt
// with split_k = 4 (bit-identical to it when K % 256 == 0: test_gemm_skinny / test_gemm_small_rows).
#include "common.h"
#include "gemm_params.h"
#include "../../include/vla_native.h"

namespace {

template <int MT, int NT, int U>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmP p) {
  constexpr int TM = 16 * MT, TN = 16 * NT, LDP = TN + 4;          // partial rows padded by 16 B
  extern __shared__ __attribute__((aligned(16))) float part[];      // [4 waves][TM rows][LDP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const int kq = p.K >> 2, nks = kq >> 5;                           // this wave's K range: nks k-steps of 32
  const int k0 = wid * kq + lq * 8;
  const bf16_t* ap[MT];
  const bf16_t* bp[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) ap[i] = p.A + (long long)min(m0 + 16 * i + lr, p.M - 1) * p.lda + k0;
#pragma unroll
  for (int j = 0; j < NT; ++j) bp[j] = p.B + (long long)min(n0 + 16 * j + lr, p.N - 1) * p.ldb + k0;

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
    for (int j = 0; j < NT; ++j) fb[u][j] = *reinterpret_cast<const bf16x8*>(bp[j] + ks * 32);
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
  const int act = p.act;
  for (int c = tid; c < TM * CPR; c += 256) {
    const int row = c / CPR, col = (c - row * CPR) * 8;
    const int m = m0 + row, n = n0 + col;
    if (m >= p.M || n >= p.N) continue;                              // (N % 8 == 0: a chunk is inside or outside)
    // the eight values of chunk (row, cc) after alpha / bias / activation (fp32, not yet rounded)
    auto chunk = [&](int cc, float* v) {
      const float* s = part + row * LDP + cc;
      f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        a0 += *reinterpret_cast<const f32x4*>(s + w * TM * LDP);
        a1 += *reinterpret_cast<const f32x4*>(s + w * TM * LDP + 4);
      }
      const float r[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        const uint4 b4 = *reinterpret_cast<const uint4*>(p.bias + n0 + cc);
        const unsigned bw[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) bb[2 * k] = bf2f((bf16_t)(bw[k] & 0xffff)), bb[2 * k + 1] = bf2f((bf16_t)(bw[k] >> 16));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float x = p.bias_post ? rbf(r[k] * p.alpha) + bb[k] : r[k] * p.alpha + bb[k];
        if (act == VLA_ACT_GELU) x = gelu_erf(rbf(x));
        else if (act == VLA_ACT_RELU) x = fmaxf(x, 0.f);
        else if (act == VLA_ACT_GELU_TANH) x = gelu_tanh(rbf(x));
        v[k] = x;
      }
    };
    float v[8];
    chunk(col, v);
    if (p.rope_mode == 1 && n < p.rope_cols) {                       // HF rotate_half, head dim 64: the partner d +- 32 is another chunk of this row (TN == 64)
      const int d0 = n & 63, lo = d0 < 32;
      float q[8];
      chunk(lo ? col + 32 : col - 32, q);
      const int pos = m % p.rope_T, d = d0 & 31;
      const float* cp = p.rope_cos + (long long)pos * 32 + d;
      const float* sp = p.rope_sin + (long long)pos * 32 + d;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
      const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float own = rbf(v[k]), oth = rbf(q[k]);                  // first half: a c - b s;  second half: b c + a s  (a = first, b = second half)
        v[k] = lo ? rbf(own * cc[k]) + rbf(-oth * ss[k]) : rbf(own * cc[k]) + rbf(oth * ss[k]);
      }
    }
    if (p.rope_mode == 2 && n < p.rope_cols) {                       // action_heads.py:125-146: pairs (2i, 2i+1), tables of cat([f, f])
      const int pos = m % p.rope_T, d = n % p.rope_dh;
      const float* cp = p.rope_cos + (long long)pos * p.rope_dh + d;
      const float* sp = p.rope_sin + (long long)pos * p.rope_dh + d;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
      const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const float x0 = rbf(v[k]), x1 = rbf(v[k + 1]);
        v[k] = rbf(x0 * cc[k]) + rbf(-x1 * ss[k]);
        v[k + 1] = rbf(x1 * cc[k + 1]) + rbf(x0 * ss[k + 1]);
      }
    }
    unsigned o[4] = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    if (p.R) {
      const uint4 r4 = *reinterpret_cast<const uint4*>(p.R + (long long)m * p.ldr + n);
      const unsigned rw[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = pack2(bf2f((bf16_t)(o[k] & 0xffff)) + bf2f((bf16_t)(rw[k] & 0xffff)), bf2f((bf16_t)(o[k] >> 16)) + bf2f((bf16_t)(rw[k] >> 16)));
    }
    *reinterpret_cast<uint4*>(p.C + (long long)m * p.ldc + n) = uint4{o[0], o[1], o[2], o[3]};
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
  hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, U>), dim3((p.M + 16 * MT - 1) / (16 * MT), (p.N + 16 * NT - 1) / (16 * NT)), dim3(256), LDS, st, p);
}

}  // namespace

// Host-side predicate + launch (called from vla_gemm_bf16_nt after its argument checks, `p` completely filled): 1 = launched here,
// 0 = not this kernel's shape / epilogue.  latency_hint: vla_gemm_latency_hint is set.  simple_addressing: batch 1, no split-K, no row groups / res_mod / c_live, no K extension, bf16.
// Contractions of 2048 and more stay on the 128-row tiles with split-K (measured, tools/diag/bench_skinny.py: with K cut into slices here
// as well, 64-row tiles pull twice the operand rows per output and lose - dt of gate/up, K = 9728: 57.7 vs 32.1 us; t of down 22.8 vs 20.0).
int vla_gemm_skinny_try(const GemmP& p, bool simple_addressing, bool latency_hint, hipStream_t st) {
  if (getenv("VLA_NO_SKINNY") || !simple_addressing) return 0;
  const bool tall = (p.N == 64 || p.N == 128 || p.N == 192) && p.M >= 1024;                         // (a)
  // (b) only under the caller's latency hint (the batch-1 pass): in a training step the head's products are off the critical path (measured:
  // no change of the step), and a product's bits would depend on its row count - the live-row backward equals the full one bit for bit
  // because every row is computed by the same instruction sequence whatever M is (tests/test_engine_gpu.py)
  const bool shortm = latency_hint && p.M <= 512 && p.N % 16 == 0 && p.K >= 512 && !getenv("VLA_NO_SMALL_ROWS");
  if (!(tall || shortm) || p.K % 128 != 0 || p.K >= 2048 || p.ldc % 8 != 0 || ((uintptr_t)p.C & 15) != 0) return 0;
  if (!(p.act == VLA_ACT_NONE || p.act == VLA_ACT_GELU || p.act == VLA_ACT_RELU || p.act == VLA_ACT_GELU_TANH)) return 0;
  if (p.bias && ((uintptr_t)p.bias & 15) != 0) return 0;
  if (p.R && (p.ldr % 8 != 0 || ((uintptr_t)p.R & 15) != 0)) return 0;
  if ((p.rope_mode == 1 && (p.rope_dh != 64 || p.rope_cols % 64 != 0 || tall)) || (p.rope_mode == 2 && (p.rope_dh % 8 != 0 || p.rope_cols % 8 != 0 || (((uintptr_t)p.rope_cos | (uintptr_t)p.rope_sin) & 15) != 0))) return 0;
  // Tile: what a CU has to pull through its L1 is (tile rows + tile columns) x K operand rows per workgroup, times the workgroups it gets -
  // minimised over the instantiated tiles (every CU busy, as few rows each as possible; ties: the larger tile)
  static const int TL[7][2] = {{1, 1}, {1, 2}, {2, 2}, {1, 4}, {2, 4}, {4, 4}, {4, 6}};
  const int ncu = vla_num_cus();
  int best = -1;
  long long best_cost = 0;
  for (int t = 0; t < 7; ++t) {
    const int tm = 16 * TL[t][0], tn = 16 * TL[t][1];
    if (tall && p.N % tn != 0) continue;
    if (p.rope_mode == 1 && tn != 64) continue;              // rotate_half: both halves of a head inside one tile row
    const long long wgs = (long long)((p.M + tm - 1) / tm) * ((p.N + tn - 1) / tn), cost = (long long)(tm + tn) * ((wgs + ncu - 1) / ncu);
    if (best < 0 || cost < best_cost || (cost == best_cost && t > best)) best = t, best_cost = cost;
  }
  // (b) only for tiles up to 32 x 64 (96 operand rows per workgroup): measured in the batch-1 pass (profiles/r04_prof_predict_summary.txt, us under
  // the profiler, this kernel vs gemm.hip's 64 x 128 six-stage ring): 16 x 16 tiles 5.1 vs 10.6 (the head's 8-row products), 32 x 32 6.6 vs 12,
  // 32 x 64 8.9-10.4 vs 12.7-12.9 (LLM o, ViT proj) - but 64 x 64 13.3 vs 12.3 (ViT q|k|v) and 64 x 96 17.3 vs 13.4 (ViT fc1): with 128+ operand
  // rows per workgroup the LDS-DMA ring keeps more bytes in flight than register fragments at one or two waves per SIMD do
  if (!tall && best_cost > 96) return 0;
  switch (best) {
    case 0: launch_skinny<1, 1, 8>(p, st); break;
    case 1: launch_skinny<1, 2, 6>(p, st); break;
    case 2: launch_skinny<2, 2, 6>(p, st); break;
    case 3: launch_skinny<1, 4, 4>(p, st); break;
    case 4: launch_skinny<2, 4, 4>(p, st); break;
    case 5: launch_skinny<4, 4, 3>(p, st); break;
    default: launch_skinny<4, 6, 2>(p, st); break;
  }
  return 1;
}
