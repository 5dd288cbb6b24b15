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
__global__ __launch_bounds__(256) void head_fwd_mfma(HP p) {
  using G = HG<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5;
  const int gid = blockIdx.x;                          // grid = B * H exactly
  const int b = gid / p.H, hd = gid - b * p.H, hoff = hd * D, N = p.T + p.Ka + p.Kt;
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem + w * G::WAVE_BYTES);
  bf16_t* sV = sK + G::TILE;
  for (int i = lane; i < 2 * G::TILE; i += 64) sK[i] = 0;
  const int qi = lane & 31, qc = min(qi, p.T - 1);
  bf16x8 qf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(p.q + ((long long)b * p.T + qc) * p.ld_q + hoff + 16 * ks + 8 * h);
  const float tg = rbf(tanhf(bf2f(p.gate[0]))), rs = sqrtf((float)D);
  f32x16 O[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) O[t] = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  u32x4 rk[G::NCH], rv[G::NCH];
  seg_prefetch_kv<D, false>(rk, rv, p, b, hoff, 32 * w, N, lane);
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
#pragma unroll
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
  for (int i = lane; i < 2 * G::TILE; i += 64) sK[i] = 0;
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
  for (int i = lane; i < 2 * G::TILE; i += 64) sQ[i] = 0;
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
    f32x16 dK[G::DT], dV[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) { dK[t] = zero16(); dV[t] = zero16(); }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack_acc(S, s), dsf = pack_acc(dS, s);
#pragma unroll
      for (int t = 0; t < G::DT; ++t) {
        dV[t] = mfma32(tr_frag(sdO, G::LD, s, 32 * t, lane), pf, dV[t]);   // dV^T[d x key] = dO^T . P
        dK[t] = mfma32(tr_frag(sQ, G::LD, s, 32 * t, lane), dsf, dK[t]);   // dK^T[d x key] = Q^T . dDot
      }
    }
    if (p.rope_cos) {
      const int pos = kc < p.T ? kc : (kc < p.T + p.Ka ? kc - p.T : kc - p.T - p.Ka);   // positions restart per segment
      rope_inter_bwd_acc<G::DT>(dK, p.rope_cos, p.rope_sin, pos, D, D, h);
    }
    if (key < N) {
      bf16_t* okp = const_cast<bf16_t*>(hrow(p.dks, p.dka, p.dkt, p, b, key, hoff));
      bf16_t* ovp = const_cast<bf16_t*>(hrow(p.dvs, p.dva, p.dvt, p, b, key, hoff));
#pragma unroll
      for (int t = 0; t < G::DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * t + 8 * g + 4 * h;
          if (d < D) {
            uint2 a = {pack2(dK[t][4 * g], dK[t][4 * g + 1]), pack2(dK[t][4 * g + 2], dK[t][4 * g + 3])};
            uint2 c = {pack2(dV[t][4 * g], dV[t][4 * g + 1]), pack2(dV[t][4 * g + 2], dV[t][4 * g + 3])};
            *reinterpret_cast<uint2*>(okp + d) = a;
            *reinterpret_cast<uint2*>(ovp + d) = c;
          }
        }
    }
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
  hipLaunchKernelGGL(head_fwd_mfma<D>, dim3(p.B * p.H), dim3(256), lds, st, p);
}
template <int D>
void launch_bwd(const HP& p, hipStream_t st) {
  const size_t lds = 4 * HG<D>::WAVE_BYTES;
  const int ntile = (p.T + p.Ka + p.Kt + 31) / 32;
  const int ndq = (p.B * p.H + 3) / 4, ndkv = (p.B * p.H * ntile + 3) / 4;
  hipLaunchKernelGGL(head_bwd_mfma<D>, dim3(ndq + ndkv), dim3(256), lds, st, p, ndq);
}
template <int D>
void set_attrs() {
  (void)hipFuncSetAttribute((const void*)head_fwd_mfma<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HG<D>::WAVE_BYTES);
  (void)hipFuncSetAttribute((const void*)head_bwd_mfma<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HG<D>::WAVE_BYTES);
}
void set_all_attrs() {
  static bool done = false;
  if (done) return;
  set_attrs<16>(); set_attrs<32>(); set_attrs<64>(); set_attrs<112>(); set_attrs<128>();
  done = true;
}

}  // namespace

bool head_attn_mfma_supported(const HP& p) {
  return (p.dh == 16 || p.dh == 32 || p.dh == 64 || p.dh == 112 || p.dh == 128) && p.T <= 32 && p.T >= 1 &&
         (long long)p.T * (p.T + p.Ka + p.Kt) >= 2 * p.T;   // probs slab must hold LSE + delta
}

void head_attn_mfma_fwd(const HP& p, hipStream_t st) {
  set_all_attrs();
  switch (p.dh) {
    case 16: launch_fwd<16>(p, st); break;
    case 32: launch_fwd<32>(p, st); break;
    case 64: launch_fwd<64>(p, st); break;
    case 112: launch_fwd<112>(p, st); break;
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
    default: launch_bwd<128>(p, st); break;
  }
}
