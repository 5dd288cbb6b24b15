// Flash-style attention forward / backward for gfx950 on v_mfma_f32_32x32x16_bf16.
//
// Replaces F.scaled_dot_product_attention inside timm Attention (SigLIP dh=72, DINOv2 dh=64; full attention) and
// flash-attn / eager attention of Qwen2Attention (dh=64, GQA 14:2, causal AND key-padding mask; call site
// modeling_prismatic.py:644-655), plus their backward.
//
// Layout trick (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"):
//   forward / dQ:  S^T[key x q] = K . Q^T  -> the query sits on the LANE, keys in the 16 accumulator registers,
//                  so row max / row sum are in-lane + one cross-half shuffle, and P^T feeds  O^T = V^T . P^T  as the
//                  B operand with no lane movement (V^T fragments come from ds_read_b64_tr_b16 on a row-major tile).
//   dK/dV:         S[q x key] = Q . K^T    -> the key sits on the lane; P and dS feed dV^T = dO^T.P and dK^T = Q^T.dS.
// 32 keys (fwd, dQ) or 32 queries (dK/dV) per LDS tile; each wave owns 32 queries (resp. 32 keys).
// The N x N score matrix is never materialised; backward recomputes P from the saved log-sum-exp.
#include "attn_common.h"
#include "../../include/vla_native.h"

int vla_num_cus();      // gemm256.hip

namespace {

struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse; const unsigned char* kmask;
  long long q_sb, k_sb, v_sb, o_sb; int q_ss, k_ss, v_ss, o_ss;
  int B, Sq, Sk, Hq, Hkv, dh, causal; float scale_log2;  // scale * log2(e)
  const bf16_t* dout; bf16_t* dq; bf16_t* dk; bf16_t* dv; float* delta; float scale;
  long long do_sb, dq_sb, dk_sb, dv_sb; int do_ss, dq_ss, dk_ss, dv_ss;
  const float* rope_cos; const float* rope_sin;   // optional: return dq/dk already through the inverse rotate_half RoPE
  int q_off;    // query i sits at sequence position q_off + i (causal: key j visible iff j <= q_off + i)
  int dkv_k0;   // dK/dV are produced for keys >= dkv_k0 only and stored at row (key - dkv_k0)
  int lse_hs;   // head stride of lse (f32 [B, Hq, lse_hs], query i at index i)
  // Causal launches, heavy blocks first: a query block of a causal attention walks (block index + 1) x 4 key tiles, a key tile of
  // dK / dV is visited by every query behind it - the blocks of one (sample, head) differ 3 : 1 in work.  With the block index as
  // the FASTEST grid dimension (x) light and heavy workgroups are dispatched alternately, the 1.3 rounds of the LLM's forward launch
  // end with heavy stragglers that started late (makespan 15 tile-steps where 10 would do).  lpt = 1: the block index moves to the
  // SLOWEST grid dimension (z), heaviest first; x carries the sample.  (hipcc dispatches x fastest, then y, then z.)
  int lpt;
};

// Tile staging split in two (cdna_hip_programming.md T14): the global loads of tile t+1 are issued into registers
// BEFORE the MFMAs of tile t and written to LDS after the next barrier, so HBM/L2 latency hides under compute.
template <int D, int NT>
struct TileGeo {
  static constexpr int CPR = D / 8;                       // 16-B chunks per row
  static constexpr int NCH = (32 * CPR + NT - 1) / NT;    // chunks per thread
};
template <int D, int NT>
__device__ __forceinline__ void tile_prefetch(u32x4* __restrict__ v, const bf16_t* src, long long row_stride,
                                              int row0, int nrows, int tid) {
  constexpr int CPR = TileGeo<D, NT>::CPR, NCH = TileGeo<D, NT>::NCH;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = min(tid + i * NT, 32 * CPR - 1);        // surplus threads re-load the last chunk (never stored)
    const int r = c / CPR, ch = c - r * CPR;
    const int gr = min(row0 + r, nrows - 1);              // clamp: masked later, must stay finite
    v[i] = *reinterpret_cast<const u32x4*>(src + (long long)gr * row_stride + ch * 8);
  }
}
template <int D, int NT, int LD>
__device__ __forceinline__ void tile_store(const u32x4* __restrict__ v, bf16_t* dst, int tid) {
  constexpr int CPR = TileGeo<D, NT>::CPR, NCH = TileGeo<D, NT>::NCH;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NT;
    if (c < 32 * CPR) {
      const int r = c / CPR, ch = c - r * CPR;
      *reinterpret_cast<u32x4*>(dst + r * LD + ch * 8) = v[i];
    }
  }
}

template <int D>
struct Geo {
  static constexpr int DQ = (D + 15) / 16 * 16;  // contraction width for QK^T (k-steps of 16)
  static constexpr int DV = (D + 31) / 32 * 32;  // output width for PV (tiles of 32)
  static constexpr int LD = DV + 8;              // LDS row stride (elements): conflict-free b128 row reads
  static constexpr int KS = DQ / 16, DT = DV / 32;
};

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 4 : D <= 72 ? 3 : 2)) void attn_fwd_kernel(AttnP p) {
  using G = Geo<D>;
  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * G::LD];
  __shared__ __attribute__((aligned(16))) bf16_t sV[32 * G::LD];
  __shared__ __attribute__((aligned(16))) bf16_t sO[4 * 32 * G::LD];     // per wave: the output tile as [query][d], for row-contiguous stores
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  const int hq = blockIdx.y, b = p.lpt ? blockIdx.x : blockIdx.z, hkv = hq / (p.Hq / p.Hkv);
  const int qblk = (p.lpt ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.x) * 128, q0 = qblk + w * 32;
  const int qi = q0 + (lane & 31);

  lds_zero16(sK, 32 * G::LD * 2, tid, 256);   // pad columns stay zero
  lds_zero16(sV, 32 * G::LD * 2, tid, 256);

  bf16x8 qf[G::KS];
  {
    const bf16_t* qp = p.q + (long long)b * p.q_sb + (long long)min(qi, p.Sq - 1) * p.q_ss + hq * D;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int d = 16 * ks + 8 * h;
      qf[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(qp + d) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  f32x16 O[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) O[t] = zero16();
  float m_run = -INFINITY, l_run = 0.f;

  const int kend = p.causal ? min(p.Sk, qblk + 128 + p.q_off) : p.Sk;
  const bf16_t* kb = p.k + (long long)b * p.k_sb + hkv * D;
  const bf16_t* vb = p.v + (long long)b * p.v_sb + hkv * D;
  u32x4 rk[TileGeo<D, 256>::NCH], rv[TileGeo<D, 256>::NCH];
  tile_prefetch<D, 256>(rk, kb, p.k_ss, 0, p.Sk, tid);
  tile_prefetch<D, 256>(rv, vb, p.v_ss, 0, p.Sk, tid);
  bool kok_next = (lane & 31) < p.Sk && (!p.kmask || p.kmask[(long long)b * p.Sk + (lane & 31)]);
  for (int k0 = 0; k0 < kend; k0 += 32) {
    __syncthreads();
    tile_store<D, 256, G::LD>(rk, sK, tid);
    tile_store<D, 256, G::LD>(rv, sV, tid);
    const unsigned km = (unsigned)__ballot(kok_next);
    __syncthreads();
    if (k0 + 32 < kend) {
      tile_prefetch<D, 256>(rk, kb, p.k_ss, k0 + 32, p.Sk, tid);
      tile_prefetch<D, 256>(rv, vb, p.v_ss, k0 + 32, p.Sk, tid);
      const int kk = k0 + 32 + (lane & 31);
      kok_next = kk < p.Sk && (!p.kmask || p.kmask[(long long)b * p.Sk + kk]);
    }
    if (p.causal && k0 > q0 + 31 + p.q_off) continue;  // wave-uniform: whole tile is in the future (barriers already passed)

    f32x16 S = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h);
      S = mfma32(kf, qf[ks], S);
    }
    // scale, then (only when the tile is not entirely visible to every query of the wave - a wave-uniform test) mask.
    // The mask-only branch replaces an earlier if/else of two complete loops; with that form accumulator register 15
    // (key rows 27 / 31) came out wrong whenever the masked arm ran with those rows unmasked.  The cause was NOT isolated
    // (no ISA diff or minimal repro was kept, so "miscompiled" is a guess, not a finding); what guards this code is the
    // sweep that caught it: tests/test_kernels_gpu.py::test_attention_fwd_sparse_key_masks (single keys, runs, both halves).
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] *= p.scale_log2;
    if (!(km == 0xffffffffu && (!p.causal || k0 + 31 <= q0 + p.q_off))) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = acc_row(r, h);
        const bool ok = ((km >> kr) & 1u) && (!p.causal || k0 + kr <= qi + p.q_off);
        S[r] = ok ? S[r] : -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, S[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fexp2(m_run - m_safe);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = fexp2(S[r] - m_safe);
      rs += S[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
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
  }
  // O^T accumulators: lane = query, registers = d.  Stored directly, every lane wrote 8-B pieces of ITS OWN row (64 scattered pieces
  // per store instruction: store-issue-bound, the same tail that cost the head's backward half its time); the wave's tile goes
  // through its LDS region instead and leaves as whole 16-B row chunks.
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  {
    bf16_t* so = sO + w * 32 * G::LD;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h;
        const uint2 o = {pack2(O[t][4 * g] * inv, O[t][4 * g + 1] * inv), pack2(O[t][4 * g + 2] * inv, O[t][4 * g + 3] * inv)};
        *reinterpret_cast<uint2*>(so + (lane & 31) * G::LD + d) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = D / 8;
    bf16_t* ob = p.o + (long long)b * p.o_sb + hq * D;
#pragma unroll
    for (int i = 0; i < (32 * CPR + 63) / 64; ++i) {
      const int c = lane + 64 * i, r = c / CPR, ch = c - r * CPR;
      if (c < 32 * CPR && q0 + r < p.Sq)
        *reinterpret_cast<u32x4*>(ob + (long long)(q0 + r) * p.o_ss + ch * 8) = *reinterpret_cast<const u32x4*>(so + r * G::LD + ch * 8);
    }
  }
  if (qi < p.Sq && p.lse && h == 0)
    p.lse[((long long)b * p.Hq + hq) * p.lse_hs + qi] = l_run > 0.f ? (m_run + log2f(l_run)) * 0.6931471805599453f : -INFINITY;
}

// ------------------------------------------------------------------------------------------------ forward, keys split over the waves
// The batch-1 pass (modeling_prismatic.py:892-972; under vla_gemm_latency_hint): with one sample the kernel above is 42 workgroups whose
// waves each walk up to 12 key tiles one after the other (14.6 us for 369 tokens).  Here a workgroup owns 32 queries and its four waves
// take every fourth key tile: K fragments straight from global memory (the same [key][d] rows the LDS image holds), the V tile through a
// wave-private LDS region (the transposed fragment read needs LDS), no workgroup barrier in the loop; the four partial (max, sum, O) meet
// in LDS and are merged flash-decoding style.  ceil(S / 32) x heads x B workgroups (168 for the LLM at batch 1), <= 3 key tiles per wave.
// Same masks, same scale, same rounding of P; the partial sums associate differently (fp32), so results agree with the kernel above to
// rounding, not bit for bit - which is why it is tied to the hint (a product's bits must not depend on the batch size in a training step).
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_split_kernel(AttnP p) {
  using G = Geo<D>;
  constexpr int LDO = G::DV + 1;
  extern __shared__ __attribute__((aligned(16))) char smem_split[];      // sV [4][32][LD] bf16 | sM [4][32] | sL [4][32] | sO [4][32][LDO] f32
  bf16_t* sV = reinterpret_cast<bf16_t*>(smem_split);
  float* sM = reinterpret_cast<float*>(smem_split + 4 * 32 * G::LD * 2);
  float* sL = sM + 4 * 32;
  float* sO = sL + 4 * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  const int hq = blockIdx.y, b = blockIdx.z, hkv = hq / (p.Hq / p.Hkv);
  const int q0 = blockIdx.x * 32, qi = q0 + (lane & 31);
  bf16_t* sv = sV + w * 32 * G::LD;
  if (D != G::DV) lds_zero16(sv, 32 * G::LD * 2, lane, 64);        // pad columns D .. DV-1 of the V image stay zero

  bf16x8 qf[G::KS];
  {
    const bf16_t* qp = p.q + (long long)b * p.q_sb + (long long)min(qi, p.Sq - 1) * p.q_ss + hq * D;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int d = 16 * ks + 8 * h;
      qf[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(qp + d) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  f32x16 O[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) O[t] = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const int kend = p.causal ? min(p.Sk, q0 + 32 + p.q_off) : p.Sk;
  const int ntile = (kend + 31) / 32;
  const bf16_t* kb = p.k + (long long)b * p.k_sb + hkv * D;
  const bf16_t* vb = p.v + (long long)b * p.v_sb + hkv * D;
  for (int kt = w; kt < ntile; kt += 4) {
    const int k0 = kt * 32, kk = k0 + (lane & 31);
    u32x4 rv[TileGeo<D, 64>::NCH];
    tile_prefetch<D, 64>(rv, vb, p.v_ss, k0, p.Sk, lane);
    bf16x8 kf[G::KS];
    {
      const bf16_t* kp = kb + (long long)min(kk, p.Sk - 1) * p.k_ss;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const int d = 16 * ks + 8 * h;
        kf[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(kp + d) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    const bool kok = kk < p.Sk && (!p.kmask || p.kmask[(long long)b * p.Sk + kk]);
    const unsigned km = (unsigned)__ballot(kok);             // (lanes 32..63 repeat 0..31: the low word is the tile's mask)
    f32x16 S = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) S = mfma32(kf[ks], qf[ks], S);
    tile_store<D, 64, G::LD>(rv, sv, lane);                  // (LDS is in order per wave: the previous tile's transposed reads are done)
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] *= p.scale_log2;
    if (!(km == 0xffffffffu && (!p.causal || k0 + 31 <= q0 + p.q_off))) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = acc_row(r, h);
        const bool ok = ((km >> kr) & 1u) && (!p.causal || k0 + kr <= qi + p.q_off);
        S[r] = ok ? S[r] : -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, S[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fexp2(m_run - m_safe);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = fexp2(S[r] - m_safe);
      rs += S[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack_acc(S, s);
#pragma unroll
      for (int t = 0; t < G::DT; ++t) O[t] = mfma32(tr_frag(sv, G::LD, s, 32 * t, lane), pf, O[t]);
    }
  }
  // partial results: lane = query (both halves hold the same max / sum), accumulator register 4 g + j of tile t = column 32 t + 8 g + 4 h + j
  if (h == 0) {
    sM[w * 32 + lane] = m_run;
    sL[w * 32 + lane] = l_run;
  }
  {
    float* so = sO + (w * 32 + (lane & 31)) * LDO;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) so[32 * t + 8 * g + 4 * h + j] = O[t][4 * g + j];
  }
  __syncthreads();
  constexpr int CPR = D / 8;
  bf16_t* ob = p.o + (long long)b * p.o_sb + hq * D;
  for (int c = tid; c < 32 * CPR; c += 256) {
    const int q = c / CPR, ch = c - q * CPR;
    if (q0 + q >= p.Sq) continue;
    float mw[4], M = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { mw[ww] = sM[ww * 32 + q]; M = fmaxf(M, mw[ww]); }
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float L = 0.f, f[4];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { f[ww] = fexp2(mw[ww] - Ms); L += sL[ww * 32 + q] * f[ww]; }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) a += sO[(ww * 32 + q) * LDO + ch * 8 + e] * f[ww];
      o[e] = a * inv;
    }
    *reinterpret_cast<u32x4*>(ob + (long long)(q0 + q) * p.o_ss + ch * 8) = u32x4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
    if (ch == 0 && p.lse) p.lse[((long long)b * p.Hq + hq) * p.lse_hs + q0 + q] = L > 0.f ? (M + log2f(L)) * 0.6931471805599453f : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 3 : 2)) void attn_bwd_dq_kernel(AttnP p) {
  using G = Geo<D>;
  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * G::LD];
  __shared__ __attribute__((aligned(16))) bf16_t sV[32 * G::LD];
  __shared__ __attribute__((aligned(16))) bf16_t sO[4 * 32 * G::LD];     // per wave: dQ as [query][d] for row-contiguous stores
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  const int hq = blockIdx.y, b = p.lpt ? blockIdx.x : blockIdx.z, hkv = hq / (p.Hq / p.Hkv);
  const int qblk = (p.lpt ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.x) * 128, q0 = qblk + w * 32;
  const int qi = q0 + (lane & 31), qc = min(qi, p.Sq - 1);
  lds_zero16(sK, 32 * G::LD * 2, tid, 256);
  lds_zero16(sV, 32 * G::LD * 2, tid, 256);

  bf16x8 qf[G::KS], dof[G::KS];
  {
    const bf16_t* qp = p.q + (long long)b * p.q_sb + (long long)qc * p.q_ss + hq * D;
    const bf16_t* dp = p.dout + (long long)b * p.do_sb + (long long)qc * p.do_ss + hq * D;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int d = 16 * ks + 8 * h;
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(qp + d) : z;
      dof[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(dp + d) : z;
    }
  }
  const long long sidx = ((long long)b * p.Hq + hq) * p.Sq + qc;
  const float lse2 = p.lse[((long long)b * p.Hq + hq) * p.lse_hs + qc] * 1.4426950408889634f;  // natural -> log2 domain
  // delta = rowsum(dO * O): each half-wave owns the d-chunks 16ks+8h.. of its query row; published for the dK/dV pass
  float delta = 0.f;
  {
    const bf16_t* op = p.o + (long long)b * p.o_sb + (long long)qc * p.o_ss + hq * D;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int d = 16 * ks + 8 * h;
      if (d < D) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(op + d);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += bf2f((bf16_t)ov[j]) * bf2f((bf16_t)dof[ks][j]);
      }
    }
    delta += __shfl_xor(delta, 32, 64);
    if (h == 0 && qi < p.Sq) p.delta[sidx] = delta;
  }
  f32x16 dQ[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) dQ[t] = zero16();

  const int kend = p.causal ? min(p.Sk, qblk + 128 + p.q_off) : p.Sk;
  const bf16_t* kb = p.k + (long long)b * p.k_sb + hkv * D;
  const bf16_t* vb = p.v + (long long)b * p.v_sb + hkv * D;
  u32x4 rk[TileGeo<D, 256>::NCH], rv[TileGeo<D, 256>::NCH];
  tile_prefetch<D, 256>(rk, kb, p.k_ss, 0, p.Sk, tid);
  tile_prefetch<D, 256>(rv, vb, p.v_ss, 0, p.Sk, tid);
  bool kok_next = (lane & 31) < p.Sk && (!p.kmask || p.kmask[(long long)b * p.Sk + (lane & 31)]);
  for (int k0 = 0; k0 < kend; k0 += 32) {
    __syncthreads();
    tile_store<D, 256, G::LD>(rk, sK, tid);
    tile_store<D, 256, G::LD>(rv, sV, tid);
    const unsigned km = (unsigned)__ballot(kok_next);
    __syncthreads();
    if (k0 + 32 < kend) {
      tile_prefetch<D, 256>(rk, kb, p.k_ss, k0 + 32, p.Sk, tid);
      tile_prefetch<D, 256>(rv, vb, p.v_ss, k0 + 32, p.Sk, tid);
      const int kk = k0 + 32 + (lane & 31);
      kok_next = kk < p.Sk && (!p.kmask || p.kmask[(long long)b * p.Sk + kk]);
    }
    if (p.causal && k0 > q0 + 31 + p.q_off) continue;
    f32x16 S = zero16(), dP = zero16();
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (lane & 31) * G::LD + 16 * ks + 8 * h);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (lane & 31) * G::LD + 16 * ks + 8 * h);
      S = mfma32(kf, qf[ks], S);
      dP = mfma32(vf, dof[ks], dP);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = acc_row(r, h);
      const bool ok = ((km >> kr) & 1u) && (!p.causal || k0 + kr <= qi + p.q_off);
      const float pr = ok ? fexp2(S[r] * p.scale_log2 - lse2) : 0.f;
      S[r] = pr * (dP[r] - delta) * p.scale;  // dS^T
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 df = pack_acc(S, s);
#pragma unroll
      for (int t = 0; t < G::DT; ++t) dQ[t] = mfma32(tr_frag(sK, G::LD, s, 32 * t, lane), df, dQ[t]);
    }
  }
  if constexpr (G::DT % 2 == 0 && D == 32 * G::DT) {   // d/dx of y = x*cos + rotate_half(x)*sin : inverse rotation (column d pairs with d + D/2: tile t with t + DT/2)
    if (p.rope_cos) {
      constexpr int HT = G::DT / 2, HALF = D / 2;
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int d = 32 * t + 8 * g + 4 * h + j;
            const float c = p.rope_cos[(long long)(qc + p.q_off) * HALF + d], sn = p.rope_sin[(long long)(qc + p.q_off) * HALF + d];
            const float a = dQ[t][4 * g + j], bb = dQ[t + HT][4 * g + j];
            dQ[t][4 * g + j] = a * c + bb * sn;
            dQ[t + HT][4 * g + j] = bb * c - a * sn;
          }
    }
  }
  {                                     // row-contiguous stores through the wave's LDS region (as attn_fwd_kernel)
    bf16_t* so = sO + w * 32 * G::LD;
#pragma unroll
    for (int t = 0; t < G::DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * h;
        const uint2 o = {pack2(dQ[t][4 * g], dQ[t][4 * g + 1]), pack2(dQ[t][4 * g + 2], dQ[t][4 * g + 3])};
        *reinterpret_cast<uint2*>(so + (lane & 31) * G::LD + d) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = D / 8;
    bf16_t* ob = p.dq + (long long)b * p.dq_sb + hq * D;
#pragma unroll
    for (int i = 0; i < (32 * CPR + 63) / 64; ++i) {
      const int c = lane + 64 * i, r = c / CPR, ch = c - r * CPR;
      if (c < 32 * CPR && q0 + r < p.Sq)
        *reinterpret_cast<u32x4*>(ob + (long long)(q0 + r) * p.dq_ss + ch * 8) = *reinterpret_cast<const u32x4*>(so + r * G::LD + ch * 8);
    }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// One workgroup = the `grp` query heads of one kv head (GQA) x KT 32-key tiles: wave w -> (head w % grp, key tile
// w / grp).  Every wave sweeps the 32-query tiles of ITS head with wave-private LDS tiles (no workgroup barrier in
// the loop: 7-14 independent waves per CU hide each other's latency); the grp partial dK^T/dV^T accumulators of a key
// tile are then summed through LDS (ds_add_f32) and written once - no global atomics, deterministic up to fp32 order.
template <int D>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(AttnP p, int grp, int KT) {
  using G = Geo<D>;
  constexpr int TILE = 32 * G::LD;                       // elements per LDS tile
  constexpr int WAVE_BYTES = 2 * TILE * 2 + 256;         // Q tile, dO tile, lse[32], delta[32]
  constexpr int ACC_LD = G::DV + 1;                      // padded: lanes (= keys) hit distinct banks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  const int hh = w % grp, kt = w / grp;
  const int hkv = blockIdx.y, b = p.lpt ? blockIdx.x : blockIdx.z, hq = hkv * grp + hh;
  const int kblk = p.lpt ? blockIdx.z : blockIdx.x;      // (causal: the first key tiles are the heavy ones - ascending IS heaviest first)
  const int k0 = p.dkv_k0 + (kblk * KT + kt) * 32;
  const int ki = k0 + (lane & 31), kc = min(ki, p.Sk - 1);
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem + w * WAVE_BYTES);
  bf16_t* sdO = sQ + TILE;
  float* sLse = reinterpret_cast<float*>(sdO + TILE);
  float* sDelta = sLse + 32;
  lds_zero16(sQ, 2 * TILE * 2, lane, 64);   // pad columns stay zero (wave-private region)

  // The key tile's K and V rows as MFMA fragments (lane = key, 8 consecutive d).  Head dims up to 72 keep them in registers for the
  // whole query loop; at 128 that is 64 registers beside 128 accumulator and 64 prefetch registers - the kernel spilled 115 dwords
  // INSIDE the loop (370 us per layer on the Qwen2.5-1.5B shape, 5 x its MFMA time) - so there the fragments live in LDS, one
  // [key][d] image per key tile behind the waves' regions (the waves of a key tile are the q heads of ONE kv head: same K, same V),
  // and are read per k-step like the Q / dO fragments.
  constexpr bool KV_LDS = D > 72;
  bf16x8 kf[KV_LDS ? 1 : G::KS], vf[KV_LDS ? 1 : G::KS];
  bf16_t* sKt = reinterpret_cast<bf16_t*>(smem + grp * KT * WAVE_BYTES) + kt * 2 * TILE;
  bf16_t* sVt = sKt + TILE;
  {
    const bf16_t* kp = p.k + (long long)b * p.k_sb + (long long)kc * p.k_ss + hkv * D;
    const bf16_t* vp = p.v + (long long)b * p.v_sb + (long long)kc * p.v_ss + hkv * D;
    if constexpr (KV_LDS) {
      if (hh == 0) {                       // one wave per key tile stages both images (pad columns: never read, d < DQ = D here)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          const int d = 16 * ks + 8 * h;
          *reinterpret_cast<bf16x8*>(sKt + (lane & 31) * G::LD + d) = *reinterpret_cast<const bf16x8*>(kp + d);
          *reinterpret_cast<bf16x8*>(sVt + (lane & 31) * G::LD + d) = *reinterpret_cast<const bf16x8*>(vp + d);
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const int d = 16 * ks + 8 * h;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        kf[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(kp + d) : z;
        vf[ks] = (d < D) ? *reinterpret_cast<const bf16x8*>(vp + d) : z;
      }
    }
  }
  const bool kok = ki < p.Sk && (!p.kmask || p.kmask[(long long)b * p.Sk + kc]);
  f32x16 dK[G::DT], dV[G::DT];
#pragma unroll
  for (int t = 0; t < G::DT; ++t) { dK[t] = zero16(); dV[t] = zero16(); }

  if (k0 < p.Sk) {
    const bf16_t* qb = p.q + (long long)b * p.q_sb + hq * D;
    const bf16_t* db = p.dout + (long long)b * p.do_sb + hq * D;
    const long long sbase = ((long long)b * p.Hq + hq) * p.Sq, lbase = ((long long)b * p.Hq + hq) * p.lse_hs;
    const int qstart = p.causal ? max(k0 - p.q_off, 0) : 0;   // k0, q_off are multiples of 32
    // (head dim 128: the next query tile is NOT held in registers across the MFMAs - those 64 registers were the other half of the
    //  spills; the tile is requested at the top of its own iteration and the workgroup's other waves cover the round trip)
    constexpr bool PF = !KV_LDS;
    u32x4 rq[TileGeo<D, 64>::NCH], rdo[TileGeo<D, 64>::NCH];
    if constexpr (PF) {
      tile_prefetch<D, 64>(rq, qb, p.q_ss, qstart, p.Sq, lane);
      tile_prefetch<D, 64>(rdo, db, p.do_ss, qstart, p.Sq, lane);
    }
    float lse_n = 0.f, delta_n = 0.f;
    if (lane < 32) {
      const int qq = min(qstart + lane, p.Sq - 1);
      lse_n = p.lse[lbase + qq];
      delta_n = p.delta[sbase + qq];
    }
    for (int q0 = qstart; q0 < p.Sq; q0 += 32) {
      if constexpr (!PF) {
        tile_prefetch<D, 64>(rq, qb, p.q_ss, q0, p.Sq, lane);
        tile_prefetch<D, 64>(rdo, db, p.do_ss, q0, p.Sq, lane);
      }
      tile_store<D, 64, G::LD>(rq, sQ, lane);
      tile_store<D, 64, G::LD>(rdo, sdO, lane);
      if (lane < 32) {
        sLse[lane] = lse_n * 1.4426950408889634f;
        sDelta[lane] = delta_n;
      }
      if (q0 + 32 < p.Sq) {
        if constexpr (PF) {
          tile_prefetch<D, 64>(rq, qb, p.q_ss, q0 + 32, p.Sq, lane);
          tile_prefetch<D, 64>(rdo, db, p.do_ss, q0 + 32, p.Sq, lane);
        }
        if (lane < 32) {
          const int qq = min(q0 + 32 + lane, p.Sq - 1);
          lse_n = p.lse[lbase + qq];
          delta_n = p.delta[sbase + qq];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // LDS tile written by this wave before it reads it
      __builtin_amdgcn_wave_barrier();
      f32x16 S = zero16(), dP = zero16();
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQ + (lane & 31) * G::LD + 16 * ks + 8 * h);
        const bf16x8 df = *reinterpret_cast<const bf16x8*>(sdO + (lane & 31) * G::LD + 16 * ks + 8 * h);
        if constexpr (KV_LDS) {
          S = mfma32(qf, *reinterpret_cast<const bf16x8*>(sKt + (lane & 31) * G::LD + 16 * ks + 8 * h), S);
          dP = mfma32(df, *reinterpret_cast<const bf16x8*>(sVt + (lane & 31) * G::LD + 16 * ks + 8 * h), dP);
        } else {
          S = mfma32(qf, kf[ks], S);     // S[q x key]: A = Q rows, B = K^T
          dP = mfma32(df, vf[ks], dP);   // dP[q x key] = dO . V^T
        }
      }
      f32x16 dS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = acc_row(r, h), qq = q0 + qr;
        const bool ok = kok && qq < p.Sq && (!p.causal || ki <= qq + p.q_off);
        const float pr = ok ? fexp2(S[r] * p.scale_log2 - sLse[qr]) : 0.f;
        S[r] = pr;
        dS[r] = pr * (dP[r] - sDelta[qr]) * p.scale;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = pack_acc(S, s), dsf = pack_acc(dS, s);
#pragma unroll
        for (int t = 0; t < G::DT; ++t) {
          dV[t] = mfma32(tr_frag(sdO, G::LD, s, 32 * t, lane), pf, dV[t]);  // dV^T[d x key] += dO^T . P
          dK[t] = mfma32(tr_frag(sQ, G::LD, s, 32 * t, lane), dsf, dK[t]);  // dK^T[d x key] += Q^T . dS
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads of this tile done before the next store
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- sum the grp heads of each key tile: every wave parks its partial tile in LDS (its own slab, no atomics -
  //      ds_add_f32 under 7-way contention cost 118 us here), then all threads add the grp slabs and store bf16.
  if (grp > 1) {
    float* part = reinterpret_cast<float*>(smem) + w * (32 * ACC_LD);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();                                    // tiles (pass 0) / previous pass's slabs are no longer read
#pragma unroll
      for (int t = 0; t < G::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          part[(lane & 31) * ACC_LD + 32 * t + acc_row(r, h)] = pass == 0 ? dK[t][r] : dV[t][r];
      __syncthreads();
      bf16_t* outp = pass == 0 ? p.dk : p.dv;
      const long long o_sb = pass == 0 ? p.dk_sb : p.dv_sb;
      const int o_ss = pass == 0 ? p.dk_ss : p.dv_ss;
      for (int i = tid; i < KT * 32 * G::DV; i += blockDim.x) {
        const int kt2 = i / (32 * G::DV), rem = i - kt2 * (32 * G::DV), key = rem / G::DV, d = rem - key * G::DV;
        const int kg = p.dkv_k0 + (kblk * KT + kt2) * 32 + key;
        if (kg < p.Sk && d < D) {
          const float* src = reinterpret_cast<const float*>(smem) + (kt2 * grp) * (32 * ACC_LD) + key * ACC_LD + d;
          float sum = 0.f;
          for (int g = 0; g < grp; ++g) sum += src[g * (32 * ACC_LD)];
          if (pass == 0 && p.rope_cos && (D == 64 || D == 128)) {   // inverse RoPE on dK: partner column d +- D/2
            constexpr int HALF = D / 2;
            const float* srp = src + (d < HALF ? HALF : -HALF);
            float other = 0.f;
            for (int g = 0; g < grp; ++g) other += srp[g * (32 * ACC_LD)];
            const float c = p.rope_cos[(long long)kg * HALF + (d & (HALF - 1))], sn = p.rope_sin[(long long)kg * HALF + (d & (HALF - 1))];
            sum = d < HALF ? sum * c + other * sn : sum * c - other * sn;
          }
          outp[(long long)b * o_sb + (long long)(kg - p.dkv_k0) * o_ss + hkv * D + d] = f2bf(sum);
        }
      }
    }
    return;
  }
  if constexpr (G::DT % 2 == 0 && D == 32 * G::DT) {
    if (p.rope_cos) {
      constexpr int HT = G::DT / 2, HALF = D / 2;
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int d = 32 * t + 8 * g + 4 * h + j;
            const float c = p.rope_cos[(long long)kc * HALF + d], sn = p.rope_sin[(long long)kc * HALF + d];
            const float a = dK[t][4 * g + j], bb = dK[t + HT][4 * g + j];
            dK[t][4 * g + j] = a * c + bb * sn;
            dK[t + HT][4 * g + j] = bb * c - a * sn;
          }
    }
  }
  if (hh == 0 && ki < p.Sk) {
    bf16_t* okp = p.dk + (long long)b * p.dk_sb + (long long)(ki - p.dkv_k0) * p.dk_ss + hkv * D;
    bf16_t* ovp = p.dv + (long long)b * p.dv_sb + (long long)(ki - p.dkv_k0) * p.dv_ss + hkv * D;
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

int fill(AttnP& p, const vla_attn_desc* d, bool bwd) {
  VLA_REQUIRE(d && d->q && d->k && d->v && d->o, "attn: null tensor");
  VLA_REQUIRE(d->B > 0 && d->Sq > 0 && d->Sk > 0 && d->Hq > 0 && d->Hkv > 0 && d->Hq % d->Hkv == 0, "attn: bad shape");
  VLA_REQUIRE(d->dh == 64 || d->dh == 72 || d->dh == 112 || d->dh == 128, "attn: dh must be 64, 72, 112 or 128");
  VLA_REQUIRE(d->q_ss % 8 == 0 && d->k_ss % 8 == 0 && d->v_ss % 8 == 0 && d->o_ss % 4 == 0, "attn: strides must keep 16-B rows");
  VLA_REQUIRE(d->q_sb % 8 == 0 && d->k_sb % 8 == 0 && d->v_sb % 8 == 0 && d->o_sb % 4 == 0, "attn: batch strides alignment");
  VLA_REQUIRE((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) == 0 && ((uintptr_t)d->o & 7) == 0, "attn: alignment");
  VLA_REQUIRE(d->q_ss >= d->Hq * d->dh && d->k_ss >= d->Hkv * d->dh && d->v_ss >= d->Hkv * d->dh, "attn: seq stride < heads*dh");
  p.q = (const bf16_t*)d->q; p.k = (const bf16_t*)d->k; p.v = (const bf16_t*)d->v; p.o = (bf16_t*)d->o;
  p.lse = d->lse; p.kmask = d->kmask;
  p.q_sb = d->q_sb; p.k_sb = d->k_sb; p.v_sb = d->v_sb; p.o_sb = d->o_sb;
  p.q_ss = d->q_ss; p.k_ss = d->k_ss; p.v_ss = d->v_ss; p.o_ss = d->o_ss;
  p.B = d->B; p.Sq = d->Sq; p.Sk = d->Sk; p.Hq = d->Hq; p.Hkv = d->Hkv; p.dh = d->dh; p.causal = d->causal;
  p.scale = d->scale; p.scale_log2 = d->scale * 1.4426950408889634f;
  p.q_off = d->q_off; p.dkv_k0 = d->dkv_k0; p.lse_hs = d->lse_hs > 0 ? d->lse_hs : d->Sq;
  VLA_REQUIRE(d->q_off >= 0 && d->q_off % 32 == 0 && (d->q_off == 0 || (d->causal && d->q_off + d->Sq <= d->Sk)),
              "attn: q_off must be a multiple of 32, causal only, q_off + Sq <= Sk");
  VLA_REQUIRE(d->dkv_k0 >= 0 && d->dkv_k0 % 32 == 0 && d->dkv_k0 < d->Sk && p.lse_hs >= d->Sq, "attn: bad dkv_k0 / lse_hs");
  if (bwd) {
    VLA_REQUIRE(d->dout && d->dq && d->dk && d->dv && d->delta && d->lse, "attn_bwd: null tensor");
    VLA_REQUIRE(d->do_ss % 8 == 0 && d->do_sb % 8 == 0 && ((uintptr_t)d->dout & 15) == 0, "attn_bwd: dout alignment");
    VLA_REQUIRE(d->dq_ss % 4 == 0 && d->dk_ss % 4 == 0 && d->dv_ss % 4 == 0 && d->dq_sb % 4 == 0 && d->dk_sb % 4 == 0 &&
                d->dv_sb % 4 == 0 && (((uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 7) == 0, "attn_bwd: grad alignment");
    VLA_REQUIRE(d->o_ss % 8 == 0 && d->o_sb % 8 == 0 && ((uintptr_t)d->o & 15) == 0, "attn_bwd: o must allow 16-B row loads");
    p.dout = (const bf16_t*)d->dout; p.dq = (bf16_t*)d->dq; p.dk = (bf16_t*)d->dk; p.dv = (bf16_t*)d->dv; p.delta = d->delta;
    p.do_sb = d->do_sb; p.dq_sb = d->dq_sb; p.dk_sb = d->dk_sb; p.dv_sb = d->dv_sb;
    p.do_ss = d->do_ss; p.dq_ss = d->dq_ss; p.dk_ss = d->dk_ss; p.dv_ss = d->dv_ss;
    p.rope_cos = d->rope_cos; p.rope_sin = d->rope_sin;
    if (d->rope_cos) VLA_REQUIRE(d->rope_sin && (d->dh == 64 || d->dh == 128) && d->Sq + d->q_off == d->Sk, "attn_bwd: fused inverse RoPE needs dh 64 or 128 (tables f32 [Sk, dh/2])");
  }
  return VLA_OK;
}

}  // namespace

extern "C" int vla_attn_fwd(void* stream, const vla_attn_desc* d) {
  AttnP p{};
  int rc = fill(p, d, false);
  if (rc) return rc;
  dim3 grid((p.Sq + 127) / 128, p.Hq, p.B);
  hipStream_t st = (hipStream_t)stream;
  // the batch-1 pass (caller's latency hint): fewer workgroups than half the CUs -> 32-query workgroups whose waves split the keys
  if (vla_gemm_latency_hint(-1) > 0 && (p.dh == 64 || p.dh == 72) && (long long)grid.x * grid.y * grid.z * 2 <= vla_num_cus() &&
      !getenv("VLA_NO_ATTN_SPLIT")) {
    const dim3 g2((p.Sq + 31) / 32, p.Hq, p.B);
    const int dv = (p.dh + 31) / 32 * 32;
    const size_t lds = (size_t)4 * 32 * (dv + 8) * 2 + 2 * 4 * 32 * 4 + (size_t)4 * 32 * (dv + 1) * 4;
    static bool split_attr = false;
    if (!split_attr) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_split_kernel<72>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      split_attr = true;
    }
    if (p.dh == 64) hipLaunchKernelGGL(attn_fwd_split_kernel<64>, g2, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(attn_fwd_split_kernel<72>, g2, dim3(256), lds, st, p);
    VLA_CHECK_LAUNCH("attn_fwd(split)");
    return VLA_OK;
  }
  if (p.causal && grid.x > 1 && grid.x <= 65535) { p.lpt = 1; grid = dim3(p.B, p.Hq, grid.x); }     // heavy query blocks first (AttnP::lpt)
  switch (p.dh) {
    case 64: hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, st, p); break;
    case 72: hipLaunchKernelGGL(attn_fwd_kernel<72>, grid, dim3(256), 0, st, p); break;
    case 112: hipLaunchKernelGGL(attn_fwd_kernel<112>, grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), 0, st, p); break;
  }
  VLA_CHECK_LAUNCH("attn_fwd");
  return VLA_OK;
}

extern "C" int vla_attn_bwd(void* stream, const vla_attn_desc* d) {
  AttnP p{};
  int rc = fill(p, d, true);
  if (rc) return rc;
  VLA_REQUIRE(p.dh == 64 || p.dh == 72 || p.dh == 128, "attn_bwd: dh 64, 72 or 128 only");
  VLA_REQUIRE(p.dh == 64 || p.dh == 128 || !p.rope_cos, "attn_bwd: the fused inverse RoPE needs head dim 64 or 128 (apply vla_rope_half with sign -1 otherwise)");
  hipStream_t st = (hipStream_t)stream;
  const int grp = p.Hq / p.Hkv;
  VLA_REQUIRE(grp <= 8, "attn_bwd: at most 8 query heads per kv head");
  const int KT = grp >= 4 ? 1 : 4 / grp;                   // waves per workgroup = grp * KT (4..8)
  dim3 gq((p.Sq + 127) / 128, p.Hq, p.B), gk((p.Sk - p.dkv_k0 + 32 * KT - 1) / (32 * KT), p.Hkv, p.B);
  if (p.causal && gq.x <= 65535 && gk.x <= 65535 && (gq.x > 1 || gk.x > 1)) {      // heavy blocks first (AttnP::lpt): both kernels read the flag
    p.lpt = 1;
    gq = dim3(p.B, p.Hq, gq.x);
    gk = dim3(p.B, p.Hkv, gk.x);
  }
  const int ld = ((p.dh + 31) / 32 * 32 + 8);
  const size_t lds = (size_t)grp * KT * (2 * 32 * ld * 2 + 256) + (p.dh > 72 ? (size_t)KT * 2 * 32 * ld * 2 : 0);   // (+ the K / V images at dh 128)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<72>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  VLA_REQUIRE(lds <= 160 * 1024, "attn_bwd: LDS budget exceeded");
  if (p.dh == 64) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, gk, dim3(64 * grp * KT), lds, st, p, grp, KT);
  } else if (p.dh == 72) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<72>, gq, dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<72>, gk, dim3(64 * grp * KT), lds, st, p, grp, KT);
  } else {                  // Qwen2.5-1.5B (BASELINE configs[4] backbone): 12 query / 2 kv heads x 128
    hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, gk, dim3(64 * grp * KT), lds, st, p, grp, KT);
  }
  VLA_CHECK_LAUNCH("attn_bwd");
  return VLA_OK;
}
