// Action-head attention on the matrix cores (MLPResNetBlock_Pro.forward, action_heads.py:391-401): T<=32 action
// queries of one (sample, head) against three key/value segments [self | adapter | task]; third segment's scores
// scaled by tanh(gating_factor); softmax over all T+Ka+Kt keys.
//
// Forward: one WORKGROUP per (sample, head); its 4 waves take the 32-key tiles round-robin (wave w: tiles w, w+4, ...)
// on wave-private LDS tiles and meet once at the end (flash-style merge of (m, l, O) through LDS): the key loop of one
// (sample, head) is a serial chain of dependent global loads, and with one wave per (sample, head) a launch was 256 such
// chains of 11 tiles on a quarter of the CUs (37 us); split four ways it is 3 tiles (18 us).
// Backward: one WAVE per (sample, head) for dQ and per (sample, head, key tile) for dK/dV - 4 independent waves per
// workgroup, wave-private LDS tiles, no workgroup barrier (the same four-way split of the dQ loop measured 81 vs 85 us
// isolated but +60 us per layer inside the step: four times the workgroups queue for LDS behind the ViT's GEMMs).
// Same operand tricks as attention.hip: forward and dQ use the S^T[key x q] orientation (query on the lane, softmax
// statistics lane-local, P^T feeds O^T = V^T.P^T directly, V^T from ds_read_b64_tr_b16); dK/dV use S[q x key] (key on
// the lane) with the single 32-row query tile parked in LDS and K/V fragments loaded straight from global memory,
// so every key tile's dK/dV is complete after one pass (no accumulation across tiles, no atomics).
// Rounding points of the reference's bf16 module are kept: bf16(q.k) -> bf16(* tanh g) -> bf16(/ sqrt dh).
// The VALU kernels of head_attn.hip remain the fallback for head dims that are not a multiple of 16.
#include "attn_common.h"
#include "head_attn_params.h"

namespace {

constexpr int HEAD_KV_WAVES = 4;   // waves of a forward workgroup that share the key tiles of one (sample, head)

template <int D>
struct HG {
  static constexpr int DV = (D + 31) / 32 * 32;
  static constexpr int LD = DV + 8;
  static constexpr int KS = D / 16, DT = DV / 32;
  static constexpr int TILE = 32 * LD;                 // elements
  static constexpr int WAVE_BYTES = 2 * TILE * 2 + 256;
  static constexpr int CPR = D / 8, NCH = (32 * CPR + 63) / 64;
  static_assert(D % 16 == 0, "head dim must be a multiple of 16");
  static_assert((TILE * 2) % 16 == 0 && WAVE_BYTES % 16 == 0, "tile images are cleared and copied in 16-B pieces");
};

__device__ __forceinline__ const bf16_t* hrow(const bf16_t* s0, const bf16_t* s1, const bf16_t* s2, const HP& p, int b, int n,
                                              int hoff) {
  // branch-free segment select (the three-way if/else became ~40 exec-mask regions per key tile in the prefetch loops)
  const bool in0 = n < p.T, in1 = n < p.T + p.Ka;
  const bf16_t* base = in0 ? s0 : (in1 ? s1 : s2);
  const int len = in0 ? p.T : (in1 ? p.Ka : p.Kt);
  const int ld = in0 ? p.ld_self : (in1 ? p.ld_adp : p.ld_task);
  const int loc = in0 ? n : (in1 ? n - p.T : n - p.T - p.Ka);
  return base + ((long long)b * len + loc) * ld + hoff;
}

// 32 consecutive keys n0.. (clamped to N-1) of the segmented K AND V tensors -> registers -> wave-private LDS tiles.
// K and V of a segment share row count and row stride (checked by the host wrapper), so one element offset serves both.
// SELECT: branch-free segment select (backward: the three-way if/else became ~40 exec-mask regions per key tile, 117 ->
// 79 us for the launch) or plain branches (forward: 18 vs 23 us - its waves visit three tiles only).
template <int D, bool SELECT>
__device__ __forceinline__ void seg_prefetch_kv(u32x4* __restrict__ vk, u32x4* __restrict__ vv, const HP& p, int b, int hoff, int n0,
                                                int N, int lane) {
  using G = HG<D>;
#pragma unroll
  for (int i = 0; i < G::NCH; ++i) {
    const int c = min(lane + i * 64, 32 * G::CPR - 1);
    const int r = c / G::CPR, ch = c - r * G::CPR;
    const int n = min(n0 + r, N - 1);
    if constexpr (SELECT) {
      const bool in0 = n < p.T, in1 = n < p.T + p.Ka;
      const int len = in0 ? p.T : (in1 ? p.Ka : p.Kt);
      const int ld = in0 ? p.ld_self : (in1 ? p.ld_adp : p.ld_task);
      const int loc = in0 ? n : (in1 ? n - p.T : n - p.T - p.Ka);
      const long long off = ((long long)b * len + loc) * ld + hoff + ch * 8;
      vk[i] = *reinterpret_cast<const u32x4*>((in0 ? p.ks : (in1 ? p.ka : p.kt)) + off);
      vv[i] = *reinterpret_cast<const u32x4*>((in0 ? p.vs : (in1 ? p.va : p.vt)) + off);
    } else if (n < p.T) {
      const long long off = ((long long)b * p.T + n) * p.ld_self + hoff + ch * 8;
      vk[i] = *reinterpret_cast<const u32x4*>(p.ks + off);
      vv[i] = *reinterpret_cast<const u32x4*>(p.vs + off);
    } else if (n < p.T + p.Ka) {
      const long long off = ((long long)b * p.Ka + (n - p.T)) * p.ld_adp + hoff + ch * 8;
      vk[i] = *reinterpret_cast<const u32x4*>(p.ka + off);
      vv[i] = *reinterpret_cast<const u32x4*>(p.va + off);
    } else {
      const long long off = ((long long)b * p.Kt + (n - p.T - p.Ka)) * p.ld_task + hoff + ch * 8;
      vk[i] = *reinterpret_cast<const u32x4*>(p.kt + off);
      vv[i] = *reinterpret_cast<const u32x4*>(p.vt + off);
    }
  }
}
template <int D>
__device__ __forceinline__ void tile_put(const u32x4* __restrict__ v, bf16_t* dst, int lane) {
  using G = HG<D>;
#pragma unroll
  for (int i = 0; i < G::NCH; ++i) {
    const int c = lane + i * 64;
    if (c < 32 * G::CPR) {
      const int r = c / G::CPR, ch = c - r * G::CPR;
      *reinterpret_cast<u32x4*>(dst + r * G::LD + ch * 8) = v[i];
    }
  }
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// transpose of the action head's RoPE map on a 32-row accumulator tile set (lane = row, registers = d): pairs (2i, 2i+1)
template <int DT>
__device__ __forceinline__ void rope_inter_bwd_acc(f32x16 (&X)[DT], const float* ct, const float* st, int pos, int dh, int D, int h) {
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * t + 8 * g + 4 * h;
      if (d < D) {
        const float4 c = *reinterpret_cast<const float4*>(ct + (long long)pos * dh + d);
        const float4 s = *reinterpret_cast<const float4*>(st + (long long)pos * dh + d);
        const float a0 = X[t][4 * g], b0 = X[t][4 * g + 1], a1 = X[t][4 * g + 2], b1 = X[t][4 * g + 3];
        X[t][4 * g] = a0 * c.x + b0 * s.y;
        X[t][4 * g + 1] = b0 * c.y - a0 * s.x;
        X[t][4 * g + 2] = a1 * c.z + b1 * s.w;
        X[t][4 * g + 3] = b1 * c.w - a1 * s.z;
      }
    }
}

// score chain of the reference's bf16 module; returns the score in log2 domain (or -inf for padded keys)
__device__ __forceinline__ float score_chain(float dot, bool gated, float tg, float rs, bool valid) {
  float s = rbf(dot);
  if (gated) s = rbf(s * tg);
  s = rbf(s / rs);
  return valid ? s * 1.4426950408889634f : -INFINITY;
}

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void head_fwd_mfma(HP p) {
  using G = HG<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5;
  const int gid = blockIdx.x;                          // grid = B * H exactly
  const int b = gid / p.H, hd = gid - b * p.H, hoff = hd * D, N = p.T + p.Ka + p.Kt;
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem + w * G::WAVE_BYTES);
  bf16_t* sV = sK + G::TILE;
  lds_zero16(sK, 2 * G::TILE * 2, lane, 64);
  const int qi = lane & 31, qc = min(qi, p.T - 1);
  bf16x8 qf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(p.q + ((long long)b * p.T + qc) * p.ld_q + hoff + 16 * ks + 8 * h);
  const bf16_t graw = p.gate[0];
  f32x16 O[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) O[t] = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  u32x4 rk[G::NCH], rv[G::NCH];
  seg_prefetch_kv<D, false>(rk, rv, p, b, hoff, 32 * w, N, lane);
  __builtin_amdgcn_sched_barrier(0);          // the first tile's loads go out BEFORE anything waits for the gate (tanhf sat on its
  const float tg = rbf(tanhf(bf2f(graw))), rs = sqrtf((float)D);      // load's round trip in front of the prefetch: hf_stamps.py)
  for (int n0 = 32 * w; n0 < N; n0 += 32 * HEAD_KV_WAVES) {
    tile_put<D>(rk, sK, lane);
    tile_put<D>(rv, sV, lane);
    if (n0 + 32 * HEAD_KV_WAVES < N) {
      seg_prefetch_kv<D, false>(rk, rv, p, b, hoff, n0 + 32 * HEAD_KV_WAVES, N, lane);
    }
    wave_lds_sync();
    f32x16 S = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
      S = mfma32(*reinterpret_cast<const bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h), qf[ks], S);
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = n0 + acc_row(r, h);
      S[r] = score_chain(S[r], key >= p.T + p.Ka, tg, rs, key < N);
      mt = fmaxf(mt, S[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = fexp2(m_run - m_new);       // every tile a wave visits holds a valid key: m_new is finite
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = fexp2(S[r] - m_new);
      rsum += S[r];
    }
    rsum += __shfl_xor(rsum, 32, 64);
    l_run = l_run * alpha + rsum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack_acc(S, s);
#pragma unroll
      for (int t = 0; t < G::DT; ++t) O[t] = mfma32(tr_frag(sV, G::LD, s, 32 * t, lane), pf, O[t]);
    }
    wave_lds_sync();
  }
  // merge the waves' partial softmax states in wave 0 (a wave without tiles carries m = -inf, l = 0, O = 0)
  {
    float* myO = reinterpret_cast<float*>(smem + w * G::WAVE_BYTES);
    float* myml = reinterpret_cast<float*>(smem + w * G::WAVE_BYTES + 2 * G::TILE * 2);
    if (w > 0) {
#pragma unroll
      for (int t = 0; t < G::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) myO[(t * 16 + r) * 64 + lane] = O[t][r];
      if (lane < 32) { myml[lane] = m_run; myml[32 + lane] = l_run; }
    }
    __syncthreads();
    if (w > 0) return;
    float M = m_run;
#pragma unroll
    for (int ww = 1; ww < HEAD_KV_WAVES; ++ww)
      M = fmaxf(M, reinterpret_cast<const float*>(smem + ww * G::WAVE_BYTES + 2 * G::TILE * 2)[qi]);
    const float a0 = fexp2(m_run - M);
    l_run *= a0;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= a0;
#pragma nounroll                                              // (unrolled, the 3 x 64 partial values are all loaded up front: 244 registers, one wave per SIMD)
    for (int ww = 1; ww < HEAD_KV_WAVES; ++ww) {
      const float* oO = reinterpret_cast<const float*>(smem + ww * G::WAVE_BYTES);
      const float* oml = reinterpret_cast<const float*>(smem + ww * G::WAVE_BYTES + 2 * G::TILE * 2);
      const float a = fexp2(oml[qi] - M);
      l_run += oml[32 + qi] * a;
#pragma unroll
      for (int t = 0; t < G::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] += oO[(t * 16 + r) * 64 + lane] * a;
    }
    m_run = M;
  }
  if (qi < p.T) {
    const float inv = 1.f / l_run;
    bf16_t* op = p.out + ((long long)b * p.T + qi) * p.ld_out + hoff;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h;
        if (d < D) {
          uint2 o = {pack2(O[t][4 * g] * inv, O[t][4 * g + 1] * inv), pack2(O[t][4 * g + 2] * inv, O[t][4 * g + 3] * inv)};
          *reinterpret_cast<uint2*>(op + d) = o;
        }
      }
    if (h == 0) p.probs[(long long)gid * p.T * N + qi] = (m_run + log2f(l_run)) * 0.6931471805599453f;   // LSE slot
  }
}

// The same forward with the softmax weights rounded where ATen's bf16 softmax rounds them (HP::ref_softmax): the weights of a
// bf16 module leave softmax as bf16(exp(s - max) / sum) - AFTER normalisation - so max and sum over all T + Ka + Kt keys must be
// known before the first weight is formed.  Pass 1 walks the wave's key tiles for the statistics only (scores through the same
// rounded chain), the four waves meet once in LDS for the global (max, sum) per query; pass 2 walks the tiles again, forms
// P = bf16(exp(s - max) / sum) and accumulates O^T = V^T P^T; the waves' partial O are plain sums.  K is read twice (131 KB per
// (sample, head) at 585 keys: L2); used by the reference-run fixture tests and by anyone who wants the head to track a bf16 PyTorch
// run as closely as the GEMM summation order allows.  Natural-log domain for max / sum: s - max is then an exact bf16 difference.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void head_fwd_mfma_ref(HP p) {
  using G = HG<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5;
  const int gid = blockIdx.x;                          // grid = B * H exactly
  const int b = gid / p.H, hd = gid - b * p.H, hoff = hd * D, N = p.T + p.Ka + p.Kt;
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem + w * G::WAVE_BYTES);
  bf16_t* sV = sK + G::TILE;
  float* myml = reinterpret_cast<float*>(smem + w * G::WAVE_BYTES + 2 * G::TILE * 2);
  lds_zero16(sK, 2 * G::TILE * 2, lane, 64);
  const int qi = lane & 31, qc = min(qi, p.T - 1);
  bf16x8 qf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(p.q + ((long long)b * p.T + qc) * p.ld_q + hoff + 16 * ks + 8 * h);
  const float tg = rbf(tanhf(bf2f(p.gate[0]))), rs = sqrtf((float)D);
  constexpr float LN2 = 0.6931471805599453f, LOG2E = 1.4426950408889634f;
  u32x4 rk[G::NCH], rv[G::NCH];
  // ---- pass 1: max and sum of exp over this wave's tiles (natural units)
  float m_run = -INFINITY, l_run = 0.f;
  for (int n0 = 32 * w; n0 < N; n0 += 32 * HEAD_KV_WAVES) {
    seg_prefetch_kv<D, false>(rk, rv, p, b, hoff, n0, N, lane);
    tile_put<D>(rk, sK, lane);
    wave_lds_sync();
    f32x16 S = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
      S = mfma32(*reinterpret_cast<const bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h), qf[ks], S);
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = n0 + acc_row(r, h);
      S[r] = score_chain(S[r], key >= p.T + p.Ka, tg, rs, key < N) * LN2;      // (back to natural units; -inf stays -inf)
      mt = fmaxf(mt, S[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    float rsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) rsum += fexp2((S[r] - m_new) * LOG2E);
    rsum += __shfl_xor(rsum, 32, 64);
    l_run = l_run * fexp2((m_run - m_new) * LOG2E) + rsum;
    m_run = m_new;
    wave_lds_sync();
  }
  if (lane < 32) { myml[lane] = m_run; myml[32 + lane] = l_run; }
  __syncthreads();
  float M = -INFINITY, L = 0.f;
#pragma unroll
  for (int ww = 0; ww < HEAD_KV_WAVES; ++ww)
    M = fmaxf(M, reinterpret_cast<const float*>(smem + ww * G::WAVE_BYTES + 2 * G::TILE * 2)[qi]);
#pragma unroll
  for (int ww = 0; ww < HEAD_KV_WAVES; ++ww) {
    const float* oml = reinterpret_cast<const float*>(smem + ww * G::WAVE_BYTES + 2 * G::TILE * 2);
    const float mw = oml[qi];
    L += mw == -INFINITY ? 0.f : oml[32 + qi] * fexp2((mw - M) * LOG2E);       // (a wave without tiles carries m = -inf, l = 0)
  }
  // ---- pass 2: P = bf16(exp(s - M) / L), O^T += V^T P^T
  f32x16 O[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) O[t] = zero16();
  for (int n0 = 32 * w; n0 < N; n0 += 32 * HEAD_KV_WAVES) {
    seg_prefetch_kv<D, false>(rk, rv, p, b, hoff, n0, N, lane);
    tile_put<D>(rk, sK, lane);
    tile_put<D>(rv, sV, lane);
    wave_lds_sync();
    f32x16 S = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
      S = mfma32(*reinterpret_cast<const bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h), qf[ks], S);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = n0 + acc_row(r, h);
      const float s = score_chain(S[r], key >= p.T + p.Ka, tg, rs, key < N) * LN2;
      S[r] = fexp2((s - M) * LOG2E) / L;           // (pack_acc rounds to bf16: the weight ATen's softmax emits; padded keys: 0)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack_acc(S, s);
#pragma unroll
      for (int t = 0; t < G::DT; ++t) O[t] = mfma32(tr_frag(sV, G::LD, s, 32 * t, lane), pf, O[t]);
    }
    wave_lds_sync();
  }
  __syncthreads();                                  // every wave is done with its tiles: the regions now carry the partial O
  {
    float* myO = reinterpret_cast<float*>(smem + w * G::WAVE_BYTES);
    if (w > 0) {
#pragma unroll
      for (int t = 0; t < G::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) myO[(t * 16 + r) * 64 + lane] = O[t][r];
    }
    __syncthreads();
    if (w > 0) return;
#pragma nounroll
    for (int ww = 1; ww < HEAD_KV_WAVES; ++ww) {
      const float* oO = reinterpret_cast<const float*>(smem + ww * G::WAVE_BYTES);
#pragma unroll
      for (int t = 0; t < G::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] += oO[(t * 16 + r) * 64 + lane];
    }
  }
  if (qi < p.T) {
    bf16_t* op = p.out + ((long long)b * p.T + qi) * p.ld_out + hoff;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h;
        if (d < D) {
          uint2 o = {pack2(O[t][4 * g], O[t][4 * g + 1]), pack2(O[t][4 * g + 2], O[t][4 * g + 3])};
          *reinterpret_cast<uint2*>(op + d) = o;
        }
      }
    if (h == 0) p.probs[(long long)gid * p.T * N + qi] = M + logf(L);   // LSE slot (the backward rebuilds fp32 weights from it)
  }
}

// ------------------------------------------------------------------------------------------------ dQ (+ gate gradient)
template <int D>
__device__ __forceinline__ void head_dq_body(const HP& p, const int blk, char* smem) {
  using G = HG<D>;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5;
  const int gid = blk * 4 + w;
  if (gid >= p.B * p.H) return;
  const int b = gid / p.H, hd = gid - b * p.H, hoff = hd * D, N = p.T + p.Ka + p.Kt;
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem + w * G::WAVE_BYTES);
  bf16_t* sV = sK + G::TILE;
  lds_zero16(sK, 2 * G::TILE * 2, lane, 64);
  const int qi = lane & 31, qc = min(qi, p.T - 1);
  bf16x8 qf[G::KS], dof[G::KS];
  float delta = 0.f;
  {
    const long long ro = ((long long)b * p.T + qc);
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int d = hoff + 16 * ks + 8 * h;
      qf[ks] = *reinterpret_cast<const bf16x8*>(p.q + ro * p.ld_q + d);
      dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + ro * p.ld_out + d);
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.out + ro * p.ld_out + d);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += bf2f((bf16_t)ov[j]) * bf2f((bf16_t)dof[ks][j]);
    }
    delta += __shfl_xor(delta, 32, 64);
  }
  const float* slot = p.probs + (long long)gid * p.T * N;  // [0,T): LSE (forward)
  const float lse2 = slot[qc] * 1.4426950408889634f;
  const float g0 = bf2f(p.gate[0]);
  const float tg = rbf(tanhf(g0)), rs = sqrtf((float)D), irs = 1.f / rs;
  f32x16 dQ[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) dQ[t] = zero16();
  float gpart = 0.f;
  u32x4 rk[G::NCH], rv[G::NCH];
  seg_prefetch_kv<D, true>(rk, rv, p, b, hoff, 0, N, lane);
  for (int n0 = 0; n0 < N; n0 += 32) {
    tile_put<D>(rk, sK, lane);
    tile_put<D>(rv, sV, lane);
    if (n0 + 32 < N) {
      seg_prefetch_kv<D, true>(rk, rv, p, b, hoff, n0 + 32, N, lane);
    }
    wave_lds_sync();
    f32x16 S = zero16(), dP = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      S = mfma32(*reinterpret_cast<const bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h), qf[ks], S);
      dP = mfma32(*reinterpret_cast<const bf16x8*>(sV + (lane & 31) * G::LD + 16 * ks + 8 * h), dof[ks], dP);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = n0 + acc_row(r, h);
      const bool gated = key >= p.T + p.Ka, valid = key < N && qi < p.T;
      const float dot = rbf(S[r]);
      const float pr = valid ? fexp2(score_chain(S[r], gated, tg, rs, true) - lse2) : 0.f;
      const float ds = pr * (dP[r] - delta) * irs;         // d(score before the /sqrt(dh))
      if (gated) gpart += ds * dot;                        // d tanh(g)
      S[r] = gated ? ds * tg : ds;                         // d(q.k)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 df = pack_acc(S, s);
#pragma unroll
      for (int t = 0; t < G::DT; ++t) dQ[t] = mfma32(tr_frag(sK, G::LD, s, 32 * t, lane), df, dQ[t]);
    }
    wave_lds_sync();
  }
  if (p.rope_cos) rope_inter_bwd_acc<G::DT>(dQ, p.rope_cos, p.rope_sin, qc, D, D, h);
  if (qi < p.T) {
    bf16_t* op = p.dq + ((long long)b * p.T + qi) * p.ld_q + hoff;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h;
        if (d < D) {
          uint2 o = {pack2(dQ[t][4 * g], dQ[t][4 * g + 1]), pack2(dQ[t][4 * g + 2], dQ[t][4 * g + 3])};
          *reinterpret_cast<uint2*>(op + d) = o;
        }
      }
  }
  gpart = wave_sum(gpart);
  if (lane == 0 && p.dgate) {
    const float th = tanhf(g0);
    atomicAdd(p.dgate, gpart * (1.f - th * th));
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int D>
__device__ __forceinline__ void head_dkv_body(const HP& p, const int blk, char* smem) {
  using G = HG<D>;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5;
  const int N = p.T + p.Ka + p.Kt, ntile = (N + 31) / 32;
  const int wgid = blk * 4 + w;                        // one wave per (sample, head, 32-key tile): every tile's dK/dV is
  if (wgid >= p.B * p.H * ntile) return;               // independent, so the pass is embarrassingly parallel
  const int gid = wgid / ntile, n0 = (wgid - gid * ntile) * 32;
  const int b = gid / p.H, hd = gid - b * p.H, hoff = hd * D;
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem + w * G::WAVE_BYTES);
  bf16_t* sdO = sQ + G::TILE;
  float* sLse = reinterpret_cast<float*>(sdO + G::TILE);
  float* sDelta = sLse + 32;
  lds_zero16(sQ, 2 * G::TILE * 2, lane, 64);
  wave_lds_sync();
  // the single query tile (rows >= T stay zero) and its row constants
  for (int c = lane; c < p.T * G::CPR; c += 64) {
    const int r = c / G::CPR, ch = c - r * G::CPR;
    const long long ro = ((long long)b * p.T + r);
    *reinterpret_cast<u32x4*>(sQ + r * G::LD + ch * 8) = *reinterpret_cast<const u32x4*>(p.q + ro * p.ld_q + hoff + ch * 8);
    *reinterpret_cast<u32x4*>(sdO + r * G::LD + ch * 8) = *reinterpret_cast<const u32x4*>(p.dout + ro * p.ld_out + hoff + ch * 8);
  }
  const float* slot = p.probs + (long long)gid * p.T * N;
  if (lane < 32) {
    sLse[lane] = lane < p.T ? slot[lane] * 1.4426950408889634f : 0.f;
    sDelta[lane] = 0.f;
  }
  wave_lds_sync();
  // delta[q] = rowsum(dO * O), recomputed here (8 lanes per query row) so that this pass does not depend on the dQ pass:
  // both run as ONE launch (the dQ waves are few and long, the dK/dV waves many and short)
  for (int r0 = 0; r0 < p.T; r0 += 8) {
    const int r = r0 + (lane >> 3), part = lane & 7;
    float acc = 0.f;
    if (r < p.T) {
      const long long ro = ((long long)b * p.T + r);
      for (int ch = part; ch < G::CPR; ch += 8) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.out + ro * p.ld_out + hoff + ch * 8);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(sdO + r * G::LD + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += bf2f((bf16_t)ov[j]) * bf2f((bf16_t)dv[j]);
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (part == 0 && r < p.T) sDelta[r] = acc;
  }
  wave_lds_sync();
  const float tg = rbf(tanhf(bf2f(p.gate[0]))), rs = sqrtf((float)D), irs = 1.f / rs;
  {
    const int key = n0 + (lane & 31), kc = min(key, N - 1);
    const bool gated = key >= p.T + p.Ka;
    const bf16_t* kp = hrow(p.ks, p.ka, p.kt, p, b, kc, hoff);
    const bf16_t* vp = hrow(p.vs, p.va, p.vt, p, b, kc, hoff);
    f32x16 S = zero16(), dP = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + 16 * ks + 8 * h);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vp + 16 * ks + 8 * h);
      S = mfma32(*reinterpret_cast<const bf16x8*>(sQ + (lane & 31) * G::LD + 16 * ks + 8 * h), kf, S);     // S[q x key]
      dP = mfma32(*reinterpret_cast<const bf16x8*>(sdO + (lane & 31) * G::LD + 16 * ks + 8 * h), vf, dP);  // dO . V^T
    }
    f32x16 dS;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = acc_row(r, h);
      const bool valid = key < N && qr < p.T;
      const float pr = valid ? fexp2(score_chain(S[r], gated, tg, rs, true) - sLse[qr]) : 0.f;
      const float ds = pr * (dP[r] - sDelta[qr]) * irs;
      S[r] = pr;
      dS[r] = gated ? ds * tg : ds;
    }
    // One 32-column block of dV and dK at a time, stored before the next is formed: 2 x 16 live accumulator registers instead of
    // DT x 2 x 16 (128 at head dim 112) - the kernel needed 380 registers and spilled under its 256-register cap.
    const bf16x8 pf0 = pack_acc(S, 0), pf1 = pack_acc(S, 1), dsf0 = pack_acc(dS, 0), dsf1 = pack_acc(dS, 1);
    const int pos = kc < p.T ? kc : (kc < p.T + p.Ka ? kc - p.T : kc - p.T - p.Ka);   // positions restart per segment
    bf16_t* okp = const_cast<bf16_t*>(hrow(p.dks, p.dka, p.dkt, p, b, kc, hoff));
    bf16_t* ovp = const_cast<bf16_t*>(hrow(p.dvs, p.dva, p.dvt, p, b, kc, hoff));
#pragma unroll
    for (int t = 0; t < G::DT; ++t) {
      f32x16 dVt = zero16(), dKt = zero16();
      dVt = mfma32(tr_frag(sdO, G::LD, 0, 32 * t, lane), pf0, dVt);    // dV^T[d x key] = dO^T . P
      dVt = mfma32(tr_frag(sdO, G::LD, 1, 32 * t, lane), pf1, dVt);
      dKt = mfma32(tr_frag(sQ, G::LD, 0, 32 * t, lane), dsf0, dKt);    // dK^T[d x key] = Q^T . dDot
      dKt = mfma32(tr_frag(sQ, G::LD, 1, 32 * t, lane), dsf1, dKt);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h;
        if (d < D) {
          float a0 = dKt[4 * g], b0 = dKt[4 * g + 1], a1 = dKt[4 * g + 2], b1 = dKt[4 * g + 3];
          if (p.rope_cos) {                                              // inverse interleaved RoPE (rope_inter_bwd_acc, one block)
            const float4 c = *reinterpret_cast<const float4*>(p.rope_cos + (long long)pos * D + d);
            const float4 sn = *reinterpret_cast<const float4*>(p.rope_sin + (long long)pos * D + d);
            const float x0 = a0 * c.x + b0 * sn.y, y0 = b0 * c.y - a0 * sn.x, x1 = a1 * c.z + b1 * sn.w, y1 = b1 * c.w - a1 * sn.z;
            a0 = x0; b0 = y0; a1 = x1; b1 = y1;
          }
          if (key < N) {
            *reinterpret_cast<uint2*>(okp + d) = uint2{pack2(a0, b0), pack2(a1, b1)};
            *reinterpret_cast<uint2*>(ovp + d) = uint2{pack2(dVt[4 * g], dVt[4 * g + 1]), pack2(dVt[4 * g + 2], dVt[4 * g + 3])};
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, tile-uniform form
// (round 3) One WAVE per (sample, head, 32-key tile) does everything that tile contributes: dK and dV of its keys (complete),
// its share of dQ and of the gate gradient (fp32 partials in a workspace, summed in tile order by head_dq_reduce).  The long dQ
// waves of the combined kernel above (one per (sample, head): a serial loop over all key tiles with 64 accumulator + 112
// operand / prefetch registers) set that kernel's register allocation at 380 (256 with spills) and its LDS at 17.6 KB per
// wave; here every wave is the same short chain at <= 128 registers, and a workgroup = four consecutive key tiles of ONE
// (sample, head) shares the staged Q / dO rows (T + 1 rows each: row T is the zero row every padded query row reads), so
// four workgroups fit a CU.  Both score orientations are formed from the SAME fragments (S = Q.K^T with the key on the lane
// for dK / dV, S^T = K.Q^T with the query on the lane for dQ: the operands of one are the swapped operands of the other).
template <int D>
__global__ __launch_bounds__(256, 3) void head_bwd_tiles(HP p, float* __restrict__ ws_dq, float* __restrict__ ws_gate) {
  using G = HG<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  const int N = p.T + p.Ka + p.Kt, ntile = (N + 31) / 32, nq = (ntile + 3) / 4;
  const int gid = blockIdx.x / nq, tq = blockIdx.x - gid * nq;
  const int b = gid / p.H, hd = gid - b * p.H, hoff = hd * D;
  const int RQ = p.T + 1;                                   // staged query rows + the zero row
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem);
  bf16_t* sdO = sQ + RQ * G::LD;
  float* sLse = reinterpret_cast<float*>(sdO + RQ * G::LD);
  float* sDelta = sLse + 32;
  bf16_t* sK = reinterpret_cast<bf16_t*>(sDelta + 32) + w * G::TILE;
  const int tile = tq * 4 + w;
  const bool active = tile < ntile;
  const int n0 = tile * 32;
  const int key = n0 + (lane & 31), kc = min(key, N - 1);
  const bool gated = key >= p.T + p.Ka;
  const bf16_t graw = p.gate[0];                            // (requested with the rest of the prologue: used behind the barrier)
  // ---- ONE round of global latency: the tile's K and V rows as MFMA fragments (lane = key, 8 consecutive d) are requested first,
  //      then the workgroup's query rows (shared by its four key tiles), LSE and delta = rowsum(dO * O) from global as well
  bf16x8 kf[G::KS], vf[G::KS];
  {
    const bf16_t* kp = hrow(p.ks, p.ka, p.kt, p, b, kc, hoff);
    const bf16_t* vp = hrow(p.vs, p.va, p.vt, p, b, kc, hoff);
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks + 8 * h);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks + 8 * h);
    }
  }
  for (int c = tid; c < p.T * G::CPR; c += 256) {
    const int r = c / G::CPR, ch = c - r * G::CPR;
    const long long ro = ((long long)b * p.T + r);
    *reinterpret_cast<u32x4*>(sQ + r * G::LD + ch * 8) = *reinterpret_cast<const u32x4*>(p.q + ro * p.ld_q + hoff + ch * 8);
    *reinterpret_cast<u32x4*>(sdO + r * G::LD + ch * 8) = *reinterpret_cast<const u32x4*>(p.dout + ro * p.ld_out + hoff + ch * 8);
  }
  {                                                           // zero: pad columns [D, LD) of the T staged rows, and row T
    constexpr int PADC = (G::LD - D) / 8;
    for (int i = tid; i < 2 * p.T * PADC; i += 256) {
      const int which = i / (p.T * PADC), j = i - which * p.T * PADC, r = j / PADC, c = D + 8 * (j - r * PADC);
      *reinterpret_cast<u32x4*>((which ? sdO : sQ) + r * G::LD + c) = u32x4{0, 0, 0, 0};
    }
    for (int i = tid; i < 2 * G::LD / 8; i += 256) {
      const int which = i / (G::LD / 8), c = 8 * (i - which * (G::LD / 8));
      *reinterpret_cast<u32x4*>((which ? sdO : sQ) + p.T * G::LD + c) = u32x4{0, 0, 0, 0};
    }
  }
  const float* slot = p.probs + (long long)gid * p.T * N;
  if (tid < 32) sLse[tid] = tid < p.T ? slot[tid] * 1.4426950408889634f : 0.f;
  for (int r0 = w * 8; r0 < 32; r0 += 32) {                   // 8 lanes per query row (wave w: rows 8w .. 8w + 7)
    const int r = r0 + (lane >> 3), part = lane & 7;
    float acc = 0.f;
    if (r < p.T) {
      const long long ro = ((long long)b * p.T + r);
      for (int ch = part; ch < G::CPR; ch += 8) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.out + ro * p.ld_out + hoff + ch * 8);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(p.dout + ro * p.ld_out + hoff + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += bf2f((bf16_t)ov[j]) * bf2f((bf16_t)dv[j]);
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (part == 0) sDelta[r] = r < p.T ? acc : 0.f;
  }
  if (active) {
    if (G::DV > D) {                                         // pad columns [D, DV) of the K tile: read by the last transposed block
      for (int i = lane; i < 32 * (G::DV - D) / 8; i += 64) {
        const int r = i / ((G::DV - D) / 8), c = D + 8 * (i - r * ((G::DV - D) / 8));
        *reinterpret_cast<u32x4*>(sK + r * G::LD + c) = u32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) *reinterpret_cast<bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h) = kf[ks];
  }
  __syncthreads();
  if (!active) return;                                       // (no workgroup barrier below)
  const float tg = rbf(tanhf(bf2f(graw))), rs = sqrtf((float)D), irs = 1.f / rs;
  const bool two_steps = p.T > 16;                           // query rows 16 .. 31 exist: the second k-step of the q contraction
  const int qrow = min(lane & 31, p.T);                      // padded query rows read the zero row
  const int qi = lane & 31;
  // ---- both score orientations while the K / V fragments are live: S[q x key], dP[q x key] (key on the lane: dK, dV) and
  //      S^T[key x q], dP^T[key x q] (query on the lane: dQ) - the same fragments with the operands swapped
  bf16x8 pf0, pf1, dsf0, dsf1, df0, df1;
  float gpart = 0.f;
  {
    f32x16 S = zero16(), dP = zero16(), St = zero16(), dPt = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + qrow * G::LD + 16 * ks + 8 * h);
      const bf16x8 da = *reinterpret_cast<const bf16x8*>(sdO + qrow * G::LD + 16 * ks + 8 * h);
      S = mfma32(qa, kf[ks], S);
      dP = mfma32(da, vf[ks], dP);
      St = mfma32(kf[ks], qa, St);
      dPt = mfma32(vf[ks], da, dPt);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {                           // [q x key]: P and d(q.k)
      const int qr = acc_row(r, h);
      const bool valid = key < N && qr < p.T;
      const float pr = valid ? fexp2(score_chain(S[r], gated, tg, rs, true) - sLse[qr]) : 0.f;
      const float ds = pr * (dP[r] - sDelta[qr]) * irs;
      S[r] = pr;
      dP[r] = gated ? ds * tg : ds;
    }
    pf0 = pack_acc(S, 0); pf1 = pack_acc(S, 1); dsf0 = pack_acc(dP, 0); dsf1 = pack_acc(dP, 1);
    const float lse2 = sLse[qi], delta = sDelta[qi];
#pragma unroll
    for (int r = 0; r < 16; ++r) {                           // [key x q]: d(q.k) again, and the gate gradient
      const int kk = n0 + acc_row(r, h);
      const bool gk = kk >= p.T + p.Ka, valid = kk < N && qi < p.T;
      const float dot = rbf(St[r]);
      const float pr = valid ? fexp2(score_chain(St[r], gk, tg, rs, true) - lse2) : 0.f;
      const float ds = pr * (dPt[r] - delta) * irs;          // d(score before the /sqrt(dh))
      if (gk) gpart += ds * dot;                             // d tanh(g)
      St[r] = gk ? ds * tg : ds;                             // d(q.k)
    }
    df0 = pack_acc(St, 0); df1 = pack_acc(St, 1);
  }
  // ---- this tile's share of dQ (K read transposed from LDS) and of the gate gradient
  {
    float* wq = ws_dq + ((long long)(gid * ntile + tile) * p.T + qi) * D;
#pragma unroll
    for (int t = 0; t < G::DT; ++t) {
      f32x16 dQt = zero16();
      dQt = mfma32(tr_frag(sK, G::LD, 0, 32 * t, lane), df0, dQt);
      dQt = mfma32(tr_frag(sK, G::LD, 1, 32 * t, lane), df1, dQt);
      if (qi < p.T) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * t + 8 * g + 4 * h;
          if (d < D) *reinterpret_cast<float4*>(wq + d) = float4{dQt[4 * g], dQt[4 * g + 1], dQt[4 * g + 2], dQt[4 * g + 3]};
        }
      }
    }
    gpart = wave_sum(gpart);
    if (lane == 0) ws_gate[gid * ntile + tile] = gpart;
  }
  // ---- dV, then dK: one 32-column block at a time into the (now free) K tile as [key][d] bf16, then whole 16-B row chunks to global.
  //      The accumulators hold a key per LANE: stored directly, every lane wrote 8-B pieces of its own row - 64 scattered pieces per
  //      store instruction, 33 of this kernel's 70 thousand cycles (stamped).
  {
    const int pos = kc < p.T ? kc : (kc < p.T + p.Ka ? kc - p.T : kc - p.T - p.Ka);   // positions restart per segment
    // transposed fragments of the staged Q / dO rows: rows >= T come from the zero row (tr_frag with clamped rows)
    auto trq = [&](const bf16_t* tile_, int s_, int col0) {
      const int gi = (lane >> 4) & 1, i = lane & 15;
      const int r0 = 16 * s_ + 4 * h + (i >> 2);
      const bf16_t* p0 = tile_ + min(r0, p.T) * G::LD + col0 + 16 * gi + 4 * (i & 3);
      const bf16_t* p1 = tile_ + min(r0 + 8, p.T) * G::LD + col0 + 16 * gi + 4 * (i & 3);
      const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p0);
      const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p1);
      return bf16x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    };
    const float* rc = p.rope_cos ? p.rope_cos + (long long)pos * D : reinterpret_cast<const float*>(p.q);
    const float* rsn = p.rope_cos ? p.rope_sin + (long long)pos * D : reinterpret_cast<const float*>(p.q);
    bf16_t* stg = sK;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {                   // 0: dV, 1: dK
      wave_lds_sync();                                       // the tile's previous readers (dQ's transposed reads / pass 0's row reads) are done
#pragma unroll
      for (int t = 0; t < G::DT; ++t) {
        float4 rcv[4], rsv[4];
        if (pass == 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {                      // (RoPE table rows of the block requested before its MFMAs)
            const int d = 32 * t + 8 * g + 4 * h, dd = (p.rope_cos && d < D) ? d : 0;
            rcv[g] = *reinterpret_cast<const float4*>(rc + dd);
            rsv[g] = *reinterpret_cast<const float4*>(rsn + dd);
          }
        }
        f32x16 acc = zero16();
        acc = mfma32(trq(pass == 0 ? sdO : sQ, 0, 32 * t), pass == 0 ? pf0 : dsf0, acc);      // dV^T = dO^T . P   |   dK^T = Q^T . dDot
        if (two_steps) acc = mfma32(trq(pass == 0 ? sdO : sQ, 1, 32 * t), pass == 0 ? pf1 : dsf1, acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * t + 8 * g + 4 * h;
          float a0 = acc[4 * g], b0 = acc[4 * g + 1], a1 = acc[4 * g + 2], b1 = acc[4 * g + 3];
          if (pass == 1 && p.rope_cos) {                     // inverse interleaved RoPE (rope_inter_bwd_acc, one block)
            const float4 c = rcv[g], sn = rsv[g];
            const float x0 = a0 * c.x + b0 * sn.y, y0 = b0 * c.y - a0 * sn.x, x1 = a1 * c.z + b1 * sn.w, y1 = b1 * c.w - a1 * sn.z;
            a0 = x0; b0 = y0; a1 = x1; b1 = y1;
          }
          *reinterpret_cast<uint2*>(stg + (lane & 31) * G::LD + d) = uint2{pack2(a0, b0), pack2(a1, b1)};
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < (32 * G::CPR + 63) / 64; ++i) {    // 32 rows x CPR chunks of 16 B
        const int c = lane + 64 * i;
        if (c < 32 * G::CPR) {
          const int r = c / G::CPR, ch = c - r * G::CPR, kr = n0 + r;
          if (kr < N) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + r * G::LD + ch * 8);
            bf16_t* dst = const_cast<bf16_t*>(pass == 0 ? hrow(p.dvs, p.dva, p.dvt, p, b, kr, hoff) : hrow(p.dks, p.dka, p.dkt, p, b, kr, hoff));
            *reinterpret_cast<u32x4*>(dst + ch * 8) = v;
          }
        }
      }
    }
  }
}

// dq[b, q, head, :] = RoPE^T(sum over key tiles, in tile order, of the fp32 partials); dgate += (1 - tanh^2 g) * sum of the
// gate partials (one atomic per (sample, head), as the combined kernel did).  One workgroup per (sample, head).
template <int D>
__global__ __launch_bounds__(256) void head_dq_reduce(HP p, const float* __restrict__ ws_dq, const float* __restrict__ ws_gate) {
  const int gid = blockIdx.x, b = gid / p.H, hd = gid - b * p.H;
  const int N = p.T + p.Ka + p.Kt, ntile = (N + 31) / 32;
  for (int i = threadIdx.x; i < p.T * (D / 4); i += 256) {
    const int q = i / (D / 4), d = 4 * (i - q * (D / 4));
    float4 a = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < ntile; t0 += 12) {                 // twelve partials in flight (eleven tiles at 329 keys), added in tile order
      float4 v[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int t = min(t0 + j, ntile - 1);
        v[j] = *reinterpret_cast<const float4*>(ws_dq + ((long long)(gid * ntile + t) * p.T + q) * D + d);
      }
#pragma unroll
      for (int j = 0; j < 12; ++j)
        if (t0 + j < ntile) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    if (p.rope_cos) {
      const float4 c = *reinterpret_cast<const float4*>(p.rope_cos + (long long)q * D + d);
      const float4 s = *reinterpret_cast<const float4*>(p.rope_sin + (long long)q * D + d);
      a = float4{a.x * c.x + a.y * s.y, a.y * c.y - a.x * s.x, a.z * c.z + a.w * s.w, a.w * c.w - a.z * s.z};
    }
    *reinterpret_cast<uint2*>(p.dq + ((long long)b * p.T + q) * p.ld_q + hd * D + d) = uint2{pack2(a.x, a.y), pack2(a.z, a.w)};
  }
  if (threadIdx.x == 0 && p.dgate) {
    float g = 0.f;
    for (int t = 0; t < ntile; ++t) g += ws_gate[gid * ntile + t];
    const float th = tanhf(bf2f(p.gate[0]));
    atomicAdd(p.dgate, g * (1.f - th * th));
  }
}

// dQ (+ gate) and dK/dV in one launch: blocks [0, ndq) run the dQ body, the rest the dK/dV body
// Two waves per SIMD (<= 256 registers; unconstrained the compiler takes 380 and spills nothing).  Alone the launch is
// SLOWER this way (85 -> 117 us, 34 spilled dwords) but the training step is 0.27 ms FASTER (same box): the kernel sits on
// the backward's critical chain while the vision stream's GEMM workgroups hold 352 of a SIMD's 512 registers, and a
// 380-register wave can only start on a CU that has drained.  (<= 168 registers: 179 us alone, +1.1 ms on the step.)
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void head_bwd_mfma(HP p, int ndq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < ndq) head_dq_body<D>(p, blockIdx.x, smem);
  else head_dkv_body<D>(p, blockIdx.x - ndq, smem);
}

template <int D>
void launch_fwd(const HP& p, hipStream_t st) {
  const size_t lds = 4 * HG<D>::WAVE_BYTES;
  if (p.ref_softmax) hipLaunchKernelGGL(head_fwd_mfma_ref<D>, dim3(p.B * p.H), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(head_fwd_mfma<D>, dim3(p.B * p.H), dim3(256), lds, st, p);
}
template <int D>
void launch_bwd(const HP& p, hipStream_t st) {
  const int ntile = (p.T + p.Ka + p.Kt + 31) / 32;
  const long long need = (long long)p.B * p.H * ntile * ((long long)p.T * D + 1);
  if (p.ws != nullptr && p.ws_floats >= need && !getenv("VLA_HEAD_BWD_COMBINED")) {     // tile-uniform form (needs the workspace)
    float* ws_gate = p.ws + (long long)p.B * p.H * ntile * p.T * D;
    const size_t lds2 = (size_t)4 * (p.T + 1) * HG<D>::LD + 256 + (size_t)4 * HG<D>::TILE * 2;
    hipLaunchKernelGGL(head_bwd_tiles<D>, dim3(p.B * p.H * ((ntile + 3) / 4)), dim3(256), lds2, st, p, p.ws, ws_gate);
    hipLaunchKernelGGL(head_dq_reduce<D>, dim3(p.B * p.H), dim3(256), 0, st, p, (const float*)p.ws, (const float*)ws_gate);
    return;
  }
  const size_t lds = 4 * HG<D>::WAVE_BYTES;
  const int ndq = (p.B * p.H + 3) / 4, ndkv = (p.B * p.H * ntile + 3) / 4;
  hipLaunchKernelGGL(head_bwd_mfma<D>, dim3(ndq + ndkv), dim3(256), lds, st, p, ndq);
}
template <int D>
void set_attrs() {
  (void)hipFuncSetAttribute((const void*)head_fwd_mfma<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HG<D>::WAVE_BYTES);
  (void)hipFuncSetAttribute((const void*)head_fwd_mfma_ref<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HG<D>::WAVE_BYTES);
  (void)hipFuncSetAttribute((const void*)head_bwd_mfma<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HG<D>::WAVE_BYTES);
  (void)hipFuncSetAttribute((const void*)head_bwd_tiles<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 33 * HG<D>::LD + 256 + 4 * HG<D>::TILE * 2);
}
void set_all_attrs() {
  static bool done = false;
  if (done) return;
  set_attrs<16>(); set_attrs<32>(); set_attrs<64>(); set_attrs<112>(); set_attrs<128>(); set_attrs<192>();
  done = true;
}

}  // namespace

bool head_attn_mfma_supported(const HP& p) {
  return (p.dh == 16 || p.dh == 32 || p.dh == 64 || p.dh == 112 || p.dh == 128 || p.dh == 192) && p.T <= 32 && p.T >= 1 &&
         (long long)p.T * (p.T + p.Ka + p.Kt) >= 2 * p.T;   // probs slab must hold LSE + delta
}

void head_attn_mfma_fwd(const HP& p, hipStream_t st) {
  set_all_attrs();
  switch (p.dh) {
    case 16: launch_fwd<16>(p, st); break;
    case 32: launch_fwd<32>(p, st); break;
    case 64: launch_fwd<64>(p, st); break;
    case 112: launch_fwd<112>(p, st); break;
    case 192: launch_fwd<192>(p, st); break;      // Qwen2.5-1.5B: d 1536 / 8 heads (BASELINE configs[4])
    default: launch_fwd<128>(p, st); break;
  }
}
void head_attn_mfma_bwd(const HP& p, hipStream_t st) {
  set_all_attrs();
  switch (p.dh) {
    case 16: launch_bwd<16>(p, st); break;
    case 32: launch_bwd<32>(p, st); break;
    case 64: launch_bwd<64>(p, st); break;
    case 112: launch_bwd<112>(p, st); break;
    case 192: launch_bwd<192>(p, st); break;
    default: launch_bwd<128>(p, st); break;
  }
}
