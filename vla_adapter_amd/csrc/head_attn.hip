// Action-head attention (MLPResNetBlock_Pro.forward, action_heads.py:391-401; MLPResNetBlock.forward :262-275):
// T=8 action queries per sample attend over three key/value segments [self (T) | adapter (Ka) | task (Kt)], the
// third segment's scores scaled by tanh(gating_factor); softmax over T+Ka+Kt; 8 heads of dh = D/8 (112 @ D=896).
//
// One 256-thread workgroup per (sample, head).  The problem is tiny (8 x 329..585 scores, dh 112) and
// latency-bound, so it runs on the vector ALU out of LDS: one thread per key for QK^T / dP / dK / dV rows, one
// thread per output column for PV / dQ; softmax rows reduced with wavefront shuffles.  Rounding points follow the
// reference's bf16 module: bf16(q.k) -> bf16(* tanh g) -> bf16(/ sqrt dh) -> softmax -> bf16 -> bf16(P.V).
#include "common.h"
#include "head_attn_params.h"
#include "../../include/vla_native.h"

namespace {

constexpr int TQ = 8;  // NUM_ACTIONS_CHUNK (LIBERO)

__device__ __forceinline__ const bf16_t* seg_row(const bf16_t* s0, const bf16_t* s1, const bf16_t* s2, const HP& p, int b,
                                                 int n, int hoff) {
  if (n < p.T) return s0 + ((long long)b * p.T + n) * p.ld_self + hoff;
  if (n < p.T + p.Ka) return s1 + ((long long)b * p.Ka + (n - p.T)) * p.ld_adp + hoff;
  return s2 + ((long long)b * p.Kt + (n - p.T - p.Ka)) * p.ld_task + hoff;
}
__device__ __forceinline__ bf16_t* seg_row_w(bf16_t* s0, bf16_t* s1, bf16_t* s2, const HP& p, int b, int n, int hoff) {
  return const_cast<bf16_t*>(seg_row(s0, s1, s2, p, b, n, hoff));
}

__device__ __forceinline__ void unpack8h(const uint4& u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}

// dynamic LDS layout (floats): sq[TQ][dh] | sdo[TQ][dh] | sS[TQ][N] | sX[TQ][N] | sY[TQ][N] | red[64]
__global__ __launch_bounds__(256) void head_attn_fwd_kernel(HP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = p.T + p.Ka + p.Kt, dh = p.dh;
  float* sq = lds;
  float* sS = lds + TQ * dh;
  const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H, hoff = hd * dh;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < p.T * dh; i += 256) sq[i] = bf2f(p.q[((long long)b * p.T + i / dh) * p.ld_q + hoff + i % dh]);
  const float tg = rbf(tanhf(bf2f(p.gate[0])));
  const float rs = sqrtf((float)dh);
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const bf16_t* kr = seg_row(p.ks, p.ka, p.kt, p, b, n, hoff);
    float acc[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) acc[t] = 0.f;
    for (int c = 0; c < dh; c += 8) {
      float kv[8];
      unpack8h(*reinterpret_cast<const uint4*>(kr + c), kv);
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        const float* qq = sq + t * dh + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t] += qq[j] * kv[j];
      }
    }
    const bool gated = n >= p.T + p.Ka;
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      float s = rbf(acc[t]);
      if (gated) s = rbf(s * tg);
      sS[t * N + n] = rbf(s / rs);
    }
  }
  __syncthreads();
  // softmax: wave w handles rows w and w+4
  for (int t = w; t < p.T; t += 4) {
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, sS[t * N + n]);
    m = wave_max(m);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += __expf(sS[t * N + n] - m);
    s = wave_sum(s);
    float* pr = p.probs + (((long long)b * p.H + hd) * p.T + t) * N;
    for (int n = lane; n < N; n += 64) {
      const float v = rbf(__expf(sS[t * N + n] - m) / s);
      sS[t * N + n] = v;
      pr[n] = v;
    }
  }
  __syncthreads();
  // out[t][d] = sum_n P[t][n] V[n][d]: thread -> column d, 4 rows
  for (int d0 = 0; d0 < dh; d0 += 128) {         // (dh 112 at D = 896: one pass; 192 at D = 1536: two)
    const int d = d0 + (tid & 127), tg4 = (tid >> 7) * 4;
    if (d < dh) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int n = 0; n < N; ++n) {
        const float v = bf2f(seg_row(p.vs, p.va, p.vt, p, b, n, hoff)[d]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += sS[(tg4 + j) * N + n] * v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (tg4 + j < p.T) p.out[((long long)b * p.T + tg4 + j) * p.ld_out + hoff + d] = f2bf(acc[j]);
    }
  }
}

__global__ __launch_bounds__(256) void head_attn_bwd_kernel(HP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = p.T + p.Ka + p.Kt, dh = p.dh;
  float* sq = lds;
  float* sdo = sq + TQ * dh;
  float* sP = sdo + TQ * dh;   // probabilities
  float* sD = sP + TQ * N;     // raw dots q.k, later dDot
  float* sG = sD + TQ * N;     // dP
  float* red = sG + TQ * N;    // [TQ] row sums + [4] gate partials
  const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H, hoff = hd * dh;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < p.T * dh; i += 256) {
    const long long o = ((long long)b * p.T + i / dh);
    sq[i] = bf2f(p.q[o * p.ld_q + hoff + i % dh]);
    sdo[i] = bf2f(p.dout[o * p.ld_out + hoff + i % dh]);
  }
  const float* pr = p.probs + ((long long)b * p.H + hd) * p.T * N;
  for (int i = tid; i < p.T * N; i += 256) sP[i] = pr[i];
  const float g0 = bf2f(p.gate[0]);
  const float tg = rbf(tanhf(g0));
  const float irs = 1.f / sqrtf((float)dh);
  __syncthreads();
  // phase 1: per key: raw dots, dP, and the dV row
  for (int n = tid; n < N; n += 256) {
    const bf16_t* kr = seg_row(p.ks, p.ka, p.kt, p, b, n, hoff);
    const bf16_t* vr = seg_row(p.vs, p.va, p.vt, p, b, n, hoff);
    bf16_t* dvr = seg_row_w(p.dvs, p.dva, p.dvt, p, b, n, hoff);
    float ad[TQ], ap[TQ], pn[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) { ad[t] = 0.f; ap[t] = 0.f; pn[t] = sP[t * N + n]; }
    for (int c = 0; c < dh; c += 8) {
      float kv[8], vv[8], dv[8];
      unpack8h(*reinterpret_cast<const uint4*>(kr + c), kv);
      unpack8h(*reinterpret_cast<const uint4*>(vr + c), vv);
#pragma unroll
      for (int j = 0; j < 8; ++j) dv[j] = 0.f;
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        const float* qq = sq + t * dh + c;
        const float* dd = sdo + t * dh + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ad[t] += qq[j] * kv[j];
          ap[t] += dd[j] * vv[j];
          dv[j] += pn[t] * dd[j];
        }
      }
      *reinterpret_cast<uint4*>(dvr + c) = uint4{pack2(dv[0], dv[1]), pack2(dv[2], dv[3]), pack2(dv[4], dv[5]), pack2(dv[6], dv[7])};
    }
#pragma unroll
    for (int t = 0; t < TQ; ++t) { sD[t * N + n] = rbf(ad[t]); sG[t * N + n] = ap[t]; }
  }
  __syncthreads();
  // phase 2: row sums of P*dP
  for (int t = w; t < p.T; t += 4) {
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += sP[t * N + n] * sG[t * N + n];
    s = wave_sum(s);
    if (lane == 0) red[t] = s;
  }
  __syncthreads();
  // phase 3: dS -> d(dot); gate gradient
  float gpart = 0.f;
  for (int i = tid; i < p.T * N; i += 256) {
    const int t = i / N, n = i - t * N;
    const float ds = sP[i] * (sG[i] - red[t]) * irs;   // d(score before /sqrt(dh))
    if (n >= p.T + p.Ka) {
      gpart += ds * sD[i];                               // d tanh(g)
      sD[i] = ds * tg;
    } else {
      sD[i] = ds;
    }
  }
  gpart = wave_sum(gpart);
  if (lane == 0) red[TQ + w] = gpart;
  __syncthreads();
  if (tid == 0 && p.dgate) {
    const float th = tanhf(g0);
    atomicAdd(p.dgate, (red[TQ] + red[TQ + 1] + red[TQ + 2] + red[TQ + 3]) * (1.f - th * th));
  }
  // phase 4: dK rows = sum_t dDot[t][n] q[t][:]
  for (int n = tid; n < N; n += 256) {
    bf16_t* dkr = seg_row_w(p.dks, p.dka, p.dkt, p, b, n, hoff);
    float dn[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) dn[t] = sD[t * N + n];
    for (int c = 0; c < dh; c += 8) {
      float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        const float* qq = sq + t * dh + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += dn[t] * qq[j];
      }
      *reinterpret_cast<uint4*>(dkr + c) = uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
    }
  }
  // phase 5: dQ[t][d] = sum_n dDot[t][n] K[n][d]
  for (int d0 = 0; d0 < dh; d0 += 128) {
    const int d = d0 + (tid & 127), tg4 = (tid >> 7) * 4;
    if (d < dh) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int n = 0; n < N; ++n) {
        const float kv = bf2f(seg_row(p.ks, p.ka, p.kt, p, b, n, hoff)[d]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += sD[(tg4 + j) * N + n] * kv;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (tg4 + j < p.T) p.dq[((long long)b * p.T + tg4 + j) * p.ld_q + hoff + d] = f2bf(acc[j]);
    }
  }
}

int fill(HP& p, const vla_head_attn_desc* d, bool bwd) {
  VLA_REQUIRE(d && d->q && d->k_self && d->v_self && d->k_adp && d->v_adp && d->k_task && d->v_task && d->gate && d->probs,
              "head_attn: null tensor");
  VLA_REQUIRE(d->T >= 1 && d->T <= 32, "head_attn: T must be in 1..32");
  VLA_REQUIRE(d->B > 0 && d->Ka > 0 && d->Kt > 0 && d->H > 0 && d->dh > 0 && d->dh % 8 == 0 && d->dh <= 256, "head_attn: bad shape");
  VLA_REQUIRE(d->ld_q % 8 == 0 && d->ld_self % 8 == 0 && d->ld_adp % 8 == 0 && d->ld_task % 8 == 0 && d->ld_out % 8 == 0,
              "head_attn: row strides must be multiples of 8");
  VLA_REQUIRE(d->ld_q >= d->H * d->dh && d->ld_self >= d->H * d->dh && d->ld_adp >= d->H * d->dh && d->ld_task >= d->H * d->dh,
              "head_attn: row stride < H*dh");
  p.q = (const bf16_t*)d->q; p.ks = (const bf16_t*)d->k_self; p.vs = (const bf16_t*)d->v_self;
  p.ka = (const bf16_t*)d->k_adp; p.va = (const bf16_t*)d->v_adp; p.kt = (const bf16_t*)d->k_task; p.vt = (const bf16_t*)d->v_task;
  p.gate = (const bf16_t*)d->gate; p.out = (bf16_t*)d->out; p.probs = d->probs;
  p.B = d->B; p.T = d->T; p.Ka = d->Ka; p.Kt = d->Kt; p.H = d->H; p.dh = d->dh;
  p.ld_q = d->ld_q; p.ld_self = d->ld_self; p.ld_adp = d->ld_adp; p.ld_task = d->ld_task; p.ld_out = d->ld_out;
  p.ref_softmax = d->ref_softmax;
  if (!bwd) {
    VLA_REQUIRE(d->out, "head_attn_fwd: null out");
  } else {
    VLA_REQUIRE(d->dout && d->dq && d->dk_self && d->dv_self && d->dk_adp && d->dv_adp && d->dk_task && d->dv_task,
                "head_attn_bwd: null grad tensor");
    p.out = (bf16_t*)d->out;   // forward output (needed by the MFMA backward for delta = rowsum(dO*O))
    p.dout = (const bf16_t*)d->dout; p.dq = (bf16_t*)d->dq; p.dks = (bf16_t*)d->dk_self; p.dvs = (bf16_t*)d->dv_self;
    p.dka = (bf16_t*)d->dk_adp; p.dva = (bf16_t*)d->dv_adp; p.dkt = (bf16_t*)d->dk_task; p.dvt = (bf16_t*)d->dv_task;
    p.dgate = d->dgate;
    p.rope_cos = d->rope_cos; p.rope_sin = d->rope_sin;
    p.ws = d->ws; p.ws_floats = d->ws ? d->ws_floats : 0;
    if (d->ws) VLA_REQUIRE(((uintptr_t)d->ws & 15) == 0 && d->ws_floats >= 0, "head_attn_bwd: workspace must be 16-B aligned");
    if (d->rope_cos) VLA_REQUIRE(d->rope_sin && (((uintptr_t)d->rope_cos | (uintptr_t)d->rope_sin) & 15) == 0 && d->dh % 4 == 0, "head_attn_bwd: rope tables");
  }
  return VLA_OK;
}

}  // namespace

extern "C" int vla_head_attn_fwd(void* stream, const vla_head_attn_desc* d) {
  HP p{};
  int rc = fill(p, d, false);
  if (rc) return rc;
  if (head_attn_mfma_supported(p) && !getenv("VLA_HEAD_ATTN_VALU")) {
    head_attn_mfma_fwd(p, (hipStream_t)stream);
    VLA_CHECK_LAUNCH("head_attn_fwd(mfma)");
    return VLA_OK;
  }
  VLA_REQUIRE(p.T == TQ, "head_attn (VALU fallback): T must be 8");
  const int N = p.T + p.Ka + p.Kt;
  const size_t lds = (size_t)(TQ * p.dh + TQ * N) * sizeof(float);
  VLA_REQUIRE(lds <= 64 * 1024, "head_attn_fwd: key count too large for LDS");
  hipLaunchKernelGGL(head_attn_fwd_kernel, dim3(p.B * p.H), dim3(256), lds, (hipStream_t)stream, p);
  VLA_CHECK_LAUNCH("head_attn_fwd");
  return VLA_OK;
}

extern "C" int vla_head_attn_bwd(void* stream, const vla_head_attn_desc* d) {
  HP p{};
  int rc = fill(p, d, true);
  if (rc) return rc;
  if (head_attn_mfma_supported(p) && !getenv("VLA_HEAD_ATTN_VALU")) {
    VLA_REQUIRE(p.out, "head_attn_bwd: the MFMA path needs the forward output (desc.out)");
    head_attn_mfma_bwd(p, (hipStream_t)stream);
    VLA_CHECK_LAUNCH("head_attn_bwd(mfma)");
    return VLA_OK;
  }
  VLA_REQUIRE(p.T == TQ, "head_attn (VALU fallback): T must be 8");
  const int N = p.T + p.Ka + p.Kt;
  const size_t lds = (size_t)(2 * TQ * p.dh + 3 * TQ * N + 64) * sizeof(float);
  VLA_REQUIRE(lds <= 160 * 1024, "head_attn_bwd: key count too large for LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)head_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(head_attn_bwd_kernel, dim3(p.B * p.H), dim3(256), lds, (hipStream_t)stream, p);
  VLA_CHECK_LAUNCH("head_attn_bwd");
  if (p.rope_cos) {   // VALU fallback: apply the RoPE transpose with the stand-alone kernel (same contract as the MFMA path)
    int rc = vla_rope_interleaved(stream, p.dq, p.rope_cos, p.rope_sin, p.B * p.T, p.T, p.H, p.dh, p.ld_q, 1);
    if (!rc) rc = vla_rope_interleaved(stream, p.dks, p.rope_cos, p.rope_sin, p.B * p.T, p.T, p.H, p.dh, p.ld_self, 1);
    if (!rc) rc = vla_rope_interleaved(stream, p.dka, p.rope_cos, p.rope_sin, p.B * p.Ka, p.Ka, p.H, p.dh, p.ld_adp, 1);
    if (!rc) rc = vla_rope_interleaved(stream, p.dkt, p.rope_cos, p.rope_sin, p.B * p.Kt, p.Kt, p.H, p.dh, p.ld_task, 1);
    if (rc) return rc;
  }
  return VLA_OK;
}
