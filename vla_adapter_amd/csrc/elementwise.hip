// HBM-bound glue kernels of the fine-tune hot path (gfx950): im2col, action masks, embedding splice, gathers,
// RoPE (both conventions), SwiGLU backward, transposes, casts, L1 loss, AdamW.  All bf16 traffic is 16 B per lane.
#include "common.h"
#include "../../include/vla_native.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return uint4{pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
}
inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// ---------------------------------------------------------------- im2col for the PxP/P patch-embed conv
__global__ void im2col_kernel(const void* __restrict__ px, bf16_t* __restrict__ out, int B, int Ctot, int c0, int H,
                              int W, int P, int ldo, int f32in) {
  const int gw = W / P, gh = H / P, KK = 3 * P * P;
  const long long total = (long long)B * gh * gw * ldo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int col = i % ldo;
    const long long prow = i / ldo;
    float v = 0.f;
    if (col < KK) {
      const int c = col / (P * P), rem = col - c * P * P, py = rem / P, pxx = rem - py * P;
      const int pw = prow % gw, ph = (prow / gw) % gh, b = prow / ((long long)gw * gh);
      const long long src = (((long long)b * Ctot + c0 + c) * H + ph * P + py) * W + pw * P + pxx;
      v = f32in ? ((const float*)px)[src] : bf2f(((const bf16_t*)px)[src]);
    }
    out[i] = f2bf(v);
  }
}

// ---------------------------------------------------------------- action masks (train_utils.py:8-41): one wave per row
__global__ void action_mask_kernel(const long long* __restrict__ labels, int* __restrict__ qidx, int* __restrict__ pos,
                                   int* __restrict__ count, int B, int L, int shift) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const long long* row = labels + (long long)b * L + shift;
  const int n = L - shift;
  // cumsum(labels != -100) only gates the masks through (cumsum >= 1); both masks also need id > 151386, and
  // (1<=c<=7) | (c>7) == (c>=1), which every id != -100 satisfies at its own position -> selected = id > 151386.
  // (The cumsum is still what splits current/next; the union is what the hot path consumes.)
  int base = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const bool sel = j < n && row[j] != -100 && row[j] > 151386;
    const unsigned long long bal = __ballot(sel);
    const int k = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (j < n) qidx[(long long)b * n + j] = sel ? k : -1;
    if (sel && k < 64) pos[b * 64 + k] = j;
    base += __popcll(bal);
  }
  if (lane == 0) count[b] = base;
  for (int k = base + lane; k < 64; k += 64) pos[b * 64 + k] = -1;
}

// ---------------------------------------------------------------- embedding gather + action-query splice
__global__ void embed_splice_kernel(const long long* __restrict__ ids, const unsigned char* __restrict__ am,
                                    const int* __restrict__ qidx, const bf16_t* __restrict__ table,
                                    const bf16_t* __restrict__ aq, bf16_t* __restrict__ out,
                                    unsigned char* __restrict__ mm, int B, int L, int Np, int D, int vocab) {
  const int S = L + Np;
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long long)B * S) return;
  const int b = r / S, s = r - (long long)b * S;
  if (s >= 1 && s <= Np) {
    if (lane == 0 && mm) mm[r] = 1;
    return;
  }
  const int j = s == 0 ? 0 : s - Np;
  const int qi = qidx[(long long)b * L + j];
  long long id = ids[(long long)b * L + j];
  if (id < 0 || id >= vocab) id = 0;  // never index outside the table
  const bf16_t* src = qi >= 0 ? aq + (long long)min(qi, 63) * D : table + id * D;
  bf16_t* dst = out + r * D;
  for (int e = lane * 8; e < D; e += 512) *reinterpret_cast<uint4*>(dst + e) = *reinterpret_cast<const uint4*>(src + e);
  if (lane == 0 && mm) mm[r] = am ? (am[(long long)b * L + j] != 0) : 1;
}

__global__ void action_query_grad_kernel(const bf16_t* __restrict__ dx, const int* __restrict__ pos, float* __restrict__ dq,
                                         int B, int S, int Np, int D, int row0) {
  const int k = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) {
      const int j = pos[b * 64 + k];
      if (j < 0) continue;
      const int s = (j == 0 ? 0 : Np + j) - row0;    // dx holds the rows >= row0 of every sequence
      if (s < 0) continue;                            // caller guarantees row0 <= first action position
      a += bf2f(dx[((long long)b * S + s) * D + d]);
    }
    dq[k * D + d] = a;
  }
}

__global__ void gather_rows_kernel(const bf16_t* __restrict__ in, const int* __restrict__ idx, bf16_t* __restrict__ out,
                                   int n, int D, int ldi, int ldo) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int s = idx[r];
  if (s == -2) return;                                   // -2: leave the output row alone (-1: zero it)
  for (int e = lane * 8; e < D; e += 512) {
    uint4 v = s >= 0 ? *reinterpret_cast<const uint4*>(in + (long long)s * ldi + e) : uint4{0, 0, 0, 0};
    *reinterpret_cast<uint4*>(out + r * ldo + e) = v;
  }
}

__global__ void scatter_add_rows_kernel(const bf16_t* __restrict__ in, const int* __restrict__ idx, bf16_t* __restrict__ out,
                                        int n, int D, int ldi, int ldo) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int s = idx[r];
  if (s < 0) return;
  for (int e = lane * 8; e < D; e += 512) {
    float a[8], c[8];
    unpack8(*reinterpret_cast<const uint4*>(in + r * ldi + e), a);
    unpack8(*reinterpret_cast<const uint4*>(out + (long long)s * ldo + e), c);
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] += a[k];
    *reinterpret_cast<uint4*>(out + (long long)s * ldo + e) = pack8(c);
  }
}

// ---------------------------------------------------------------- simple vector elementwise
enum { EW_ADD = 0, EW_GELU_F = 1, EW_GELU_B = 2, EW_RELU_B = 3 };
template <int OP>
__global__ void ew_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float fa[8], fb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
    unpack8(reinterpret_cast<const uint4*>(a)[i], fa);
    if (OP != EW_GELU_F) unpack8(reinterpret_cast<const uint4*>(b)[i], fb);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == EW_ADD) o[k] = fa[k] + fb[k];
      else if (OP == EW_GELU_F) o[k] = gelu_erf(fa[k]);
      else if (OP == EW_GELU_B) o[k] = fa[k] * gelu_erf_grad(fb[k]);
      else o[k] = fb[k] > 0.f ? fa[k] : 0.f;
    }
    reinterpret_cast<uint4*>(y)[i] = pack8(o);
  }
}

// dH [M, I], GU [M, 2I] with 16-column interleave (gate block, up block) -> dGU same layout
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                  long long nchunks, int I) {
  const int cpr = I / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;           // h column of this 8-chunk (never straddles a 16-block)
    const long long go = m * 2 * I + (c >> 4) * 32 + (c & 15);
    float d[8], g[8], u[8], dg[8], du[8];
    unpack8(*reinterpret_cast<const uint4*>(dh + m * I + c), d);
    unpack8(*reinterpret_cast<const uint4*>(gu + go), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + go + 16), u);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float sg = 1.f / (1.f + __expf(-g[k]));
      const float sl = g[k] * sg;
      du[k] = d[k] * sl;
      dg[k] = d[k] * u[k] * (sg * (1.f + g[k] * (1.f - sg)));
    }
    *reinterpret_cast<uint4*>(dgu + go) = pack8(dg);
    *reinterpret_cast<uint4*>(dgu + go + 16) = pack8(du);
  }
}

// ---------------------------------------------------------------- transpose through LDS (64x64 tiles)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows,
                                                        int cols, int ldi, int ldo, long long s_in, long long s_out) {
  __shared__ bf16_t t[64][66];
  const bf16_t* ib = in + (long long)blockIdx.z * s_in;
  bf16_t* ob = out + (long long)blockIdx.z * s_out;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    t[r][tx] = (r0 + r < rows && c0 + tx < cols) ? ib[(long long)(r0 + r) * ldi + c0 + tx] : (bf16_t)0;
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < cols && r0 + tx < rows) ob[(long long)(c0 + c) * ldo + r0 + tx] = t[tx][c];
}

// 16-B variant (cols % 8 == 0, rows % 8 == 0, ld % 8 == 0, 16-B aligned bases): a lane moves 8 elements per global access
// and both sides run in full 128-B lines (8 lanes per input row on the way in, 8 lanes per output row on the way out).
// LDS pitch 66 elements = 33 dwords (odd): the transposed 2-B gathers of a wave (8 row groups x 8 columns) fall on 32
// distinct banks.
__global__ __launch_bounds__(256) void transpose_vec_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows,
                                                            int cols, int ldi, int ldo, long long s_in, long long s_out) {
  __shared__ unsigned t32[64 * 33];
  const bf16_t* ib = in + (long long)blockIdx.z * s_in;
  bf16_t* ob = out + (long long)blockIdx.z * s_out;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int r = i >> 3, ch = i & 7;
    uint4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(ib + (long long)(r0 + r) * ldi + c0 + ch * 8);
    unsigned* d = t32 + r * 33 + ch * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const bf16_t* t = reinterpret_cast<const bf16_t*>(t32);
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int g = i & 7, c = i >> 3;                     // 8 consecutive lanes = one 128-B run of an output row
    if (c0 + c < cols && r0 + g * 8 < rows) {
      unsigned o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = (unsigned)t[(g * 8 + 2 * k) * 66 + c] | ((unsigned)t[(g * 8 + 2 * k + 1) * 66 + c] << 16);
      *reinterpret_cast<uint4*>(ob + (long long)(c0 + c) * ldo + r0 + g * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
  }
}

__global__ void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int rows, int cols, int ldx, int rows_per,
                              long long s_x, long long s_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  x += (long long)blockIdx.z * s_x;
  out += (long long)blockIdx.z * s_out;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float a = 0.f;
  for (int r = r0; r < r1; ++r) a += bf2f(x[(long long)r * ldx + c]);
  atomicAdd(out + c, a);
}

// vectorised variant (cols % 8 == 0): each lane owns 8 columns (16-B loads), the 4 waves of a block split the rows
// of the chunk, partial sums meet in LDS, one f32 atomic per column per block.
__global__ __launch_bounds__(256) void colsum_vec_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                                         int ldx, int rows_per, long long s_x, long long s_out) {
  __shared__ float sm[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  x += (long long)blockIdx.z * s_x;
  out += (long long)blockIdx.z * s_out;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < cols)
    for (int r = r0 + w; r < r1; r += 4) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(x + (long long)r * ldx + c), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += f[k];
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[w][lane * 8 + k] = a[k];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int cc = blockIdx.x * 512 + i;
    if (cc < cols) atomicAdd(out + cc, sm[0][i] + sm[1][i] + sm[2][i] + sm[3][i]);
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = bf2f(x[i]);
}

// ---------------------------------------------------------------- RoPE
// HF rotate_half: pairs (i, i+dh/2); q' = bf16(bf16(q*cos) + bf16(rot(q)*sin)); tables hold bf16-rounded values.
__global__ void rope_half_kernel(bf16_t* __restrict__ x, const float* __restrict__ ct, const float* __restrict__ st, long long rows,
                                 int S, int nheads, int dh, int ldx, float sign) {
  const int half = dh / 2;
  const long long total = rows * nheads * half;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = i % half;
    const long long t = i / half;
    const int hh = t % nheads;
    const long long r = t / nheads;
    const int pos = r % S;
    bf16_t* p = x + r * ldx + hh * dh + d;
    const float a = bf2f(p[0]), b = bf2f(p[half]);
    const float c = ct[pos * half + d], s = sign * st[pos * half + d];
    p[0] = f2bf(rbf(a * c) + rbf(-b * s));
    p[half] = f2bf(rbf(b * c) + rbf(a * s));
  }
}

// action_heads.py:125-146: y[2i] = x[2i]*c[2i] - x[2i+1]*s[2i];  y[2i+1] = x[2i+1]*c[2i+1] + x[2i]*s[2i+1]
// with c/s = cos/sin(cat([f,f])) (so the two lanes of a pair use DIFFERENT frequencies).  mode 1 = transpose map.
__global__ void rope_inter_kernel(bf16_t* __restrict__ x, const float* __restrict__ ct, const float* __restrict__ st,
                                  long long rows, int T, int nheads, int dh, int ldx, int mode) {
  const int half = dh / 2;
  const long long total = rows * nheads * half;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = i % half;
    const long long t = i / half;
    const int hh = t % nheads;
    const long long r = t / nheads;
    const int pos = r % T;
    bf16_t* p = x + r * ldx + hh * dh + 2 * d;
    const float a = bf2f(p[0]), b = bf2f(p[1]);
    const float ca = ct[pos * dh + 2 * d], cb = ct[pos * dh + 2 * d + 1];
    const float sa = st[pos * dh + 2 * d], sb = st[pos * dh + 2 * d + 1];
    if (mode == 0) {
      p[0] = f2bf(rbf(a * ca) + rbf(-b * sa));
      p[1] = f2bf(rbf(b * cb) + rbf(a * sb));
    } else {
      p[0] = f2bf(a * ca + b * sb);
      p[1] = f2bf(b * cb - a * sa);
    }
  }
}

// 16-B variant: a thread owns 8 consecutive elements (4 rotation pairs) of one head; 32-bit index math
__global__ void rope_inter_vec_kernel(bf16_t* __restrict__ x, const float* __restrict__ ct, const float* __restrict__ st,
                                      int rows, int T, int nheads, int dh, int ldx, int mode) {
  const int cpr = nheads * dh / 8;
  const long long total = (long long)rows * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cpr), col = (int)(i - (long long)r * cpr) * 8;
    const int d0 = col % dh, pos = r % T;
    uint4* px = reinterpret_cast<uint4*>(x + (long long)r * ldx + col);
    float f[8], o[8];
    unpack8(*px, f);
    const float4 c0 = *reinterpret_cast<const float4*>(ct + (long long)pos * dh + d0), c1 = *reinterpret_cast<const float4*>(ct + (long long)pos * dh + d0 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(st + (long long)pos * dh + d0), s1 = *reinterpret_cast<const float4*>(st + (long long)pos * dh + d0 + 4);
    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const float a = f[k], b = f[k + 1];
      if (mode == 0) {
        o[k] = rbf(a * cc[k]) + rbf(-b * ss[k]);
        o[k + 1] = rbf(b * cc[k + 1]) + rbf(a * ss[k + 1]);
      } else {
        o[k] = a * cc[k] + b * ss[k + 1];
        o[k + 1] = b * cc[k + 1] - a * ss[k];
      }
    }
    *px = pack8(o);
  }
}

// ---------------------------------------------------------------- L1 loss (finetune.py:418-444)
__global__ __launch_bounds__(256) void l1_loss_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ tgt,
                                                      float* __restrict__ loss3, bf16_t* __restrict__ dpred, int B, int C,
                                                      int Da, float gscale) {
  __shared__ float red[3][4];
  const int n = B * C * Da;
  float a = 0.f, cur = 0.f, nxt = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = bf2f(pred[i]) - bf2f(tgt[i]);
    const float ad = fabsf(d);
    a += ad;
    if ((i / Da) % C == 0) cur += ad; else nxt += ad;
    if (dpred) dpred[i] = f2bf((d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * gscale / n);
  }
  a = wave_sum(a); cur = wave_sum(cur); nxt = wave_sum(nxt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = cur; red[2][threadIdx.x >> 6] = nxt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss3[0] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
    loss3[1] = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (B * Da);
    loss3[2] = C > 1 ? (red[2][0] + red[2][1] + red[2][2] + red[2][3]) / (B * (C - 1) * Da) : 0.f;
  }
}

// ---------------------------------------------------------------- AdamW, bf16 state, torch op-by-op rounding
__global__ void adamw_kernel(bf16_t* __restrict__ p, const void* __restrict__ g, bf16_t* __restrict__ m, bf16_t* __restrict__ v,
                             long long n, float decay, float omb1, float beta2, float omb2, float bc2_sqrt, float eps,
                             float neg_step, int g_f32, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gr = g_f32 ? rbf(((const float*)g)[i] * gscale) : bf2f(((const bf16_t*)g)[i]);
    if (!g_f32 && gscale != 1.f) gr = rbf(gr * gscale);
    float pf = rbf(bf2f(p[i]) * decay);                          // param.mul_(1 - lr*wd)
    const float m0 = bf2f(m[i]);
    const float mf = rbf(__builtin_fmaf(omb1, gr - m0, m0));     // exp_avg.lerp_(grad, 1-beta1)
    const float vf = rbf(__builtin_fmaf(omb2 * gr, gr, rbf(bf2f(v[i]) * beta2)));  // mul_(beta2).addcmul_(g,g,1-beta2)
    const float den = rbf(rbf(rbf(sqrtf(vf)) / bc2_sqrt) + eps); // (sqrt / bias_correction2_sqrt).add_(eps)
    pf = rbf(__builtin_fmaf(neg_step, mf / den, pf));            // addcdiv_(exp_avg, denom, -lr/bc1)
    p[i] = f2bf(pf); m[i] = f2bf(mf); v[i] = f2bf(vf);
  }
}

// 8 parameters per thread, 16-B accesses (the scalar kernel above stays for tails and unaligned slices): same arithmetic
__device__ __forceinline__ void adamw_one(float& pf, float gr, float& mf, float& vf, float decay, float omb1, float beta2, float omb2,
                                          float bc2_sqrt, float eps, float neg_step) {
  pf = rbf(pf * decay);
  mf = rbf(__builtin_fmaf(omb1, gr - mf, mf));
  vf = rbf(__builtin_fmaf(omb2 * gr, gr, rbf(vf * beta2)));
  const float den = rbf(rbf(rbf(sqrtf(vf)) / bc2_sqrt) + eps);
  pf = rbf(__builtin_fmaf(neg_step, mf / den, pf));
}
__global__ void adamw_vec8_kernel(uint4* __restrict__ p, const void* __restrict__ g, uint4* __restrict__ m, uint4* __restrict__ v,
                                  long long n8, float decay, float omb1, float beta2, float omb2, float bc2_sqrt, float eps,
                                  float neg_step, int g_f32, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const uint4 pv = p[i], mv = m[i], vv = v[i];
    float gr[8];
    if (g_f32) {
      const float4 g0 = ((const float4*)g)[2 * i], g1 = ((const float4*)g)[2 * i + 1];
      const float t[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) gr[k] = rbf(t[k] * gscale);
    } else {
      const uint4 gv = ((const uint4*)g)[i];
      const unsigned t[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gr[2 * k] = bf2f((bf16_t)(t[k] & 0xffff));
        gr[2 * k + 1] = bf2f((bf16_t)(t[k] >> 16));
      }
      if (gscale != 1.f) {
#pragma unroll
        for (int k = 0; k < 8; ++k) gr[k] = rbf(gr[k] * gscale);
      }
    }
    const unsigned pw[4] = {pv.x, pv.y, pv.z, pv.w}, mw[4] = {mv.x, mv.y, mv.z, mv.w}, vw[4] = {vv.x, vv.y, vv.z, vv.w};
    unsigned po[4], mo[4], vo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float p0 = bf2f((bf16_t)(pw[k] & 0xffff)), p1 = bf2f((bf16_t)(pw[k] >> 16));
      float m0 = bf2f((bf16_t)(mw[k] & 0xffff)), m1 = bf2f((bf16_t)(mw[k] >> 16));
      float v0 = bf2f((bf16_t)(vw[k] & 0xffff)), v1 = bf2f((bf16_t)(vw[k] >> 16));
      adamw_one(p0, gr[2 * k], m0, v0, decay, omb1, beta2, omb2, bc2_sqrt, eps, neg_step);
      adamw_one(p1, gr[2 * k + 1], m1, v1, decay, omb1, beta2, omb2, bc2_sqrt, eps, neg_step);
      po[k] = pack2(p0, p1); mo[k] = pack2(m0, m1); vo[k] = pack2(v0, v1);
    }
    p[i] = uint4{po[0], po[1], po[2], po[3]};
    m[i] = uint4{mo[0], mo[1], mo[2], mo[3]};
    v[i] = uint4{vo[0], vo[1], vo[2], vo[3]};
  }
}

}  // namespace

#define GRID1D(n, per) dim3(min(nblk((n), (per)), 8192u))

// ---- input stage --------------------------------------------------------------------------------------
// ToTensor + Normalize of processing_prismatic.py:128-145 for images that already have the model's input size:
// u8 HWC -> (x / 255 - mean) / std in f32 (same operation order as torchvision) -> bf16 (finetune.py:339) or f32, CHW,
// written at channel offset c0 of a channel-stacked pixel tensor (fused backbones / wrist images share one tensor).
__global__ void image_normalize_kernel(const unsigned char* __restrict__ img, void* __restrict__ out, int B, int H, int W, int Ctot,
                                       int c0, float m0, float m1, float m2, float s0, float s1, float s2, int out_f32) {
  const long long total = (long long)B * 3 * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int c = (int)((i / ((long long)W * H)) % 3);
    const long long b = i / ((long long)3 * W * H);
    const float t = (float)img[((b * H + y) * W + x) * 3 + c] / 255.0f;
    const float v = (t - (c == 0 ? m0 : c == 1 ? m1 : m2)) / (c == 0 ? s0 : c == 1 ? s1 : s2);
    const long long o = ((b * Ctot + c0 + c) * H + y) * W + x;
    if (out_f32) reinterpret_cast<float*>(out)[o] = v;
    else reinterpret_cast<bf16_t*>(out)[o] = f2bf(v);
  }
}

// ActionTokenizer.__call__ (action_tokenizer.py:60-74, use_minivlm branch): clip to [lo, hi], np.digitize against the
// caller's bin edges (count of edges <= x, compared in f64 like numpy), token id = tokenizer_len - bin index.
__global__ void action_tokenize_kernel(const float* __restrict__ act, const double* __restrict__ bins, long long* __restrict__ ids,
                                       long long n, int nbins, float lo, float hi, long long tokenizer_len) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = (double)fminf(fmaxf(act[i], lo), hi);
  int a = 0, b = nbins;                       // first edge index with bins[idx] > x  (== np.digitize(x, bins), bins increasing)
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (bins[mid] <= x) a = mid + 1;
    else b = mid;
  }
  ids[i] = tokenizer_len - a;
}

extern "C" int vla_im2col_patch(void* stream, const void* pixels, void* cols, int B, int Ctot, int c0, int H, int W, int P,
                                int ldo, int px_f32) {
  VLA_REQUIRE(pixels && cols && B > 0 && P > 0 && H % P == 0 && W % P == 0, "im2col: bad shape");
  VLA_REQUIRE(c0 >= 0 && c0 + 3 <= Ctot && ldo >= 3 * P * P, "im2col: channel window / ldo");
  const long long total = (long long)B * (H / P) * (W / P) * ldo;
  hipLaunchKernelGGL(im2col_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, pixels, (bf16_t*)cols, B, Ctot, c0,
                     H, W, P, ldo, px_f32);
  VLA_CHECK_LAUNCH("im2col");
  return VLA_OK;
}

extern "C" int vla_action_mask(void* stream, const long long* labels, int* qidx, int* pos, int* count, int B, int L, int shift) {
  VLA_REQUIRE(labels && qidx && pos && count && B > 0 && L > shift && (shift == 0 || shift == 1), "action_mask: bad args");
  hipLaunchKernelGGL(action_mask_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, labels, qidx, pos, count, B, L, shift);
  VLA_CHECK_LAUNCH("action_mask");
  return VLA_OK;
}

extern "C" int vla_embed_splice(void* stream, const long long* ids, const unsigned char* attn_mask, const int* qidx,
                                const void* table, const void* action_queries, void* out, unsigned char* mm_mask, int B,
                                int L, int Np, int D, int vocab) {
  VLA_REQUIRE(ids && qidx && table && action_queries && out && B > 0 && L > 0 && Np >= 0 && D % 8 == 0 && vocab > 0,
              "embed_splice: bad args");
  const long long rows = (long long)B * (L + Np);
  hipLaunchKernelGGL(embed_splice_kernel, dim3(nblk(rows, 4)), dim3(256), 0, (hipStream_t)stream, ids, attn_mask, qidx,
                     (const bf16_t*)table, (const bf16_t*)action_queries, (bf16_t*)out, mm_mask, B, L, Np, D, vocab);
  VLA_CHECK_LAUNCH("embed_splice");
  return VLA_OK;
}

extern "C" int vla_action_query_grad(void* stream, const void* dx, const int* pos, float* dq, int B, int S, int Np, int D,
                                     int row0) {
  VLA_REQUIRE(dx && pos && dq && B > 0 && S > 0 && S + row0 > Np && D > 0 && row0 >= 0, "action_query_grad: bad args");
  hipLaunchKernelGGL(action_query_grad_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, pos, dq, B, S, Np, D, row0);
  VLA_CHECK_LAUNCH("action_query_grad");
  return VLA_OK;
}

extern "C" int vla_gather_rows(void* stream, const void* in, const int* idx, void* out, int n, int D, int ldi, int ldo) {
  VLA_REQUIRE(in && idx && out && n > 0 && D % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0, "gather_rows: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk(n, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, idx, (bf16_t*)out, n, D, ldi, ldo);
  VLA_CHECK_LAUNCH("gather_rows");
  return VLA_OK;
}

extern "C" int vla_scatter_add_rows(void* stream, const void* in, const int* idx, void* out, int n, int D, int ldi, int ldo) {
  VLA_REQUIRE(in && idx && out && n > 0 && D % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0, "scatter_add_rows: bad args");
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(nblk(n, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, idx, (bf16_t*)out, n, D, ldi, ldo);
  VLA_CHECK_LAUNCH("scatter_add_rows");
  return VLA_OK;
}

#define EW_ENTRY(NAME, OP, NEEDB)                                                                                  \
  extern "C" int NAME(void* stream, const void* a, const void* b, void* y, long long n) {                           \
    VLA_REQUIRE(a && y && (!(NEEDB) || b) && n > 0 && n % 8 == 0, #NAME ": null / n%8");                              \
    hipLaunchKernelGGL(ew_kernel<OP>, GRID1D(n / 8, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,     \
                       (const bf16_t*)b, (bf16_t*)y, n / 8);                                                       \
    VLA_CHECK_LAUNCH(#NAME);                                                                                       \
    return VLA_OK;                                                                                                 \
  }
EW_ENTRY(vla_add_bf16, EW_ADD, 1)
EW_ENTRY(vla_gelu_bwd, EW_GELU_B, 1)
EW_ENTRY(vla_relu_bwd, EW_RELU_B, 1)
extern "C" int vla_gelu_fwd(void* stream, const void* x, void* y, long long n) {
  VLA_REQUIRE(x && y && n > 0 && n % 8 == 0, "gelu_fwd: null / n%8");
  hipLaunchKernelGGL(ew_kernel<EW_GELU_F>, GRID1D(n / 8, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)nullptr, (bf16_t*)y, n / 8);
  VLA_CHECK_LAUNCH("gelu_fwd");
  return VLA_OK;
}

extern "C" int vla_swiglu_bwd(void* stream, const void* dh, const void* gu, void* dgu, int M, int I) {
  VLA_REQUIRE(dh && gu && dgu && M > 0 && I > 0 && I % 16 == 0, "swiglu_bwd: I%16");
  const long long nch = (long long)M * (I / 8);
  hipLaunchKernelGGL(swiglu_bwd_kernel, GRID1D(nch, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, (const bf16_t*)gu, (bf16_t*)dgu, nch, I);
  VLA_CHECK_LAUNCH("swiglu_bwd");
  return VLA_OK;
}

extern "C" int vla_transpose_bf16(void* stream, const void* in, void* out, int rows, int cols, int ldi, int ldo, int batch,
                                  long long s_in, long long s_out) {
  VLA_REQUIRE(in && out && rows > 0 && cols > 0 && ldi >= cols && ldo >= rows && batch > 0, "transpose: bad args");
  dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
  const bool vec = rows % 8 == 0 && cols % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && s_in % 8 == 0 && s_out % 8 == 0 &&
                   ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(transpose_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, rows, cols, ldi, ldo, s_in, s_out);
  else
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, rows, cols, ldi, ldo, s_in, s_out);
  VLA_CHECK_LAUNCH("transpose");
  return VLA_OK;
}

extern "C" int vla_colsum_bf16(void* stream, const void* x, float* out, int rows, int cols, int ldx, int batch, long long s_x,
                               long long s_out) {
  VLA_REQUIRE(x && out && rows > 0 && cols > 0 && ldx >= cols && batch > 0, "colsum: bad args");
  if (cols % 8 == 0 && ldx % 8 == 0 && s_x % 8 == 0 && ((uintptr_t)x & 15) == 0) {
    // rows per block: enough blocks to fill the chip (a [4096, 1152] bias gradient as 3 x 16 blocks of 256 rows ran 20 us on 48 CUs:
    // 470 GB/s); about a thousand blocks, at least 16 rows each (every block ends in one f32 atomic per column)
    const int gx = (cols + 511) / 512;
    int rows_per = (int)(((long long)rows * gx * batch + 1023) / 1024);
    rows_per = rows_per < 16 ? 16 : (rows_per > 256 ? 256 : (rows_per + 3) / 4 * 4);
    dim3 grid(gx, (rows + rows_per - 1) / rows_per, batch);
    hipLaunchKernelGGL(colsum_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, rows, cols, ldx, rows_per, s_x, s_out);
    VLA_CHECK_LAUNCH("colsum");
    return VLA_OK;
  }
  const int rows_per = 128;
  dim3 grid((cols + 255) / 256, (rows + rows_per - 1) / rows_per, batch);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, rows, cols, ldx, rows_per, s_x, s_out);
  VLA_CHECK_LAUNCH("colsum");
  return VLA_OK;
}

extern "C" int vla_cast_f32_bf16(void* stream, const float* x, void* y, long long n) {
  VLA_REQUIRE(x && y && n > 0, "cast: bad args");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, GRID1D(n, 256), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
  VLA_CHECK_LAUNCH("cast_f32_bf16");
  return VLA_OK;
}
extern "C" int vla_cast_bf16_f32(void* stream, const void* x, float* y, long long n) {
  VLA_REQUIRE(x && y && n > 0, "cast: bad args");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, GRID1D(n, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
  VLA_CHECK_LAUNCH("cast_bf16_f32");
  return VLA_OK;
}

extern "C" int vla_rope_half(void* stream, void* x, const float* cos_t, const float* sin_t, int rows, int S, int nheads, int dh,
                             int ldx, int sign) {
  VLA_REQUIRE(x && cos_t && sin_t && rows > 0 && S > 0 && nheads > 0 && dh % 2 == 0 && ldx >= nheads * dh, "rope_half: bad args");
  const long long total = (long long)rows * nheads * (dh / 2);
  hipLaunchKernelGGL(rope_half_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, cos_t, sin_t, (long long)rows, S, nheads, dh, ldx, sign >= 0 ? 1.f : -1.f);
  VLA_CHECK_LAUNCH("rope_half");
  return VLA_OK;
}

extern "C" int vla_rope_interleaved(void* stream, void* x, const float* cos_t, const float* sin_t, int rows, int T, int nheads,
                                    int dh, int ldx, int mode) {
  VLA_REQUIRE(x && cos_t && sin_t && rows > 0 && T > 0 && nheads > 0 && dh % 2 == 0 && ldx >= nheads * dh, "rope_interleaved: bad args");
  const long long total = (long long)rows * nheads * (dh / 2);
  if (dh % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)x | (uintptr_t)cos_t | (uintptr_t)sin_t) & 15) == 0) {
    const long long tv = (long long)rows * (nheads * dh / 8);
    hipLaunchKernelGGL(rope_inter_vec_kernel, GRID1D(tv, 256), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, cos_t, sin_t, (int)rows, T, nheads, dh, ldx, mode);
  } else
  hipLaunchKernelGGL(rope_inter_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, cos_t, sin_t, (long long)rows, T, nheads, dh, ldx, mode);
  VLA_CHECK_LAUNCH("rope_interleaved");
  return VLA_OK;
}

extern "C" int vla_l1_loss(void* stream, const void* pred, const void* target, float* loss3, void* dpred, int B, int C, int Da,
                           float gscale) {
  VLA_REQUIRE(pred && target && loss3 && B > 0 && C > 0 && Da > 0, "l1_loss: bad args");
  hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, (const bf16_t*)target, loss3, (bf16_t*)dpred, B, C, Da, gscale);
  VLA_CHECK_LAUNCH("l1_loss");
  return VLA_OK;
}

extern "C" int vla_adamw_bf16(void* stream, void* p, const void* g, void* m, void* v, long long n, double lr, double beta1,
                              double beta2, double eps, double wd, int step, int g_f32, float gscale) {
  VLA_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
  // scalars are formed in double and narrowed once, exactly as torch narrows its Python-double hyper-parameters
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  const float decay = (float)(1.0 - lr * wd);
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  const float neg_step = (float)(-(lr / bc1));
  const float gs = gscale == 0.f ? 1.f : gscale;
  const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const long long n8 = aligned ? n / 8 : 0, done = n8 * 8;
  if (n8 > 0)
    hipLaunchKernelGGL(adamw_vec8_kernel, GRID1D(n8, 256), dim3(256), 0, (hipStream_t)stream, (uint4*)p, g, (uint4*)m, (uint4*)v, n8,
                       decay, omb1, (float)beta2, omb2, (float)sqrt(bc2), (float)eps, neg_step, g_f32, gs);
  if (done < n)          // tail (< 8 elements) or an unaligned slice: scalar kernel
    hipLaunchKernelGGL(adamw_kernel, GRID1D(n - done, 256), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p + done,
                       g_f32 ? (const void*)((const float*)g + done) : (const void*)((const bf16_t*)g + done), (bf16_t*)m + done,
                       (bf16_t*)v + done, n - done, decay, omb1, (float)beta2, omb2, (float)sqrt(bc2), (float)eps, neg_step, g_f32, gs);
  VLA_CHECK_LAUNCH("adamw");
  return VLA_OK;
}

extern "C" int vla_image_normalize_u8(void* stream, const void* img, void* out, int B, int H, int W, int Ctot, int c0,
                                      const float* mean3, const float* std3, int out_f32) {
  VLA_REQUIRE(img && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && c0 >= 0 && c0 + 3 <= Ctot, "image_normalize: bad args");
  VLA_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "image_normalize: zero std");
  const long long total = (long long)B * 3 * H * W;
  hipLaunchKernelGGL(image_normalize_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)img, out, B,
                     H, W, Ctot, c0, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_f32);
  VLA_CHECK_LAUNCH("image_normalize");
  return VLA_OK;
}

extern "C" int vla_action_tokenize(void* stream, const float* actions, const double* bins, long long* ids, long long n, int nbins,
                                   float lo, float hi, long long tokenizer_len) {
  VLA_REQUIRE(actions && bins && ids && n > 0 && nbins > 1 && lo < hi, "action_tokenize: bad args");
  hipLaunchKernelGGL(action_tokenize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, actions, bins, ids, n,
                     nbins, lo, hi, tokenizer_len);
  VLA_CHECK_LAUNCH("action_tokenize");
  return VLA_OK;
}


// ---------------------------------------------------------------- host-glue replacements (no ATen kernel on the step)
// Strided 2-D copy with optional cast and row remapping: dst[row r] <- src[row r % src_mod (src_mod > 0)], dst row r lives at
// (r / d_group) * d_group_stride + (r % d_group) * ld_dst when d_group > 0 (the GEMM's row-group rule).  dtype codes: 0 bf16, 1 f32.
template <typename TS, typename TD>
__global__ void copy2d_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long rows, int cols, long long ld_src, long long ld_dst,
                              int src_mod, int d_group, long long d_group_stride) {
  const long long total = rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const int c = (int)(i - r * cols);
    const long long rs = src_mod > 0 ? r % src_mod : r;
    const long long ro = d_group > 0 ? (r / d_group) * d_group_stride + (r % d_group) * ld_dst : r * ld_dst;
    float v;
    if constexpr (sizeof(TS) == 2) v = bf2f(src[rs * ld_src + c]); else v = src[rs * ld_src + c];
    if constexpr (sizeof(TD) == 2) dst[ro + c] = f2bf(v); else dst[ro + c] = v;
  }
}

// Row-wise dynamic fp8 (OCP e4m3) quantisation: one wave per row, 8 elements per lane and pass; amax by wave reduction, then
// q = cvt(x * 448 / amax) through v_cvt_pk_fp8_f32 (RNE, saturating on gfx950), 8 bytes per lane per pass.
__global__ void quant_fp8_rows_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ q, float* __restrict__ scale, int rows, int cols,
                                      long long ldx, long long ldq) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (long long)row * ldx;
  float amax = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) amax = fmaxf(amax, fmaxf(fabsf(bf2f((bf16_t)(w[k] & 0xffff))), fabsf(bf2f((bf16_t)(w[k] >> 16)))));
  }
  amax = wave_max(amax);
  // (IEEE divisions, one per row: with the approximate form a value that sits one ulp above a rounding tie lands below it)
  const float inv = amax > 0.f ? __fdiv_rn(448.f, amax) : 1.f;
  if (lane == 0) scale[row] = amax > 0.f ? __fdiv_rn(amax, 448.f) : 1.f;
  unsigned char* qr = q + (long long)row * ldq;
  for (int c = lane * 8; c < cols; c += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)(w[0] & 0xffff)) * inv, bf2f((bf16_t)(w[0] >> 16)) * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)(w[1] & 0xffff)) * inv, bf2f((bf16_t)(w[1] >> 16)) * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)(w[2] & 0xffff)) * inv, bf2f((bf16_t)(w[2] >> 16)) * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f((bf16_t)(w[3] & 0xffff)) * inv, bf2f((bf16_t)(w[3] >> 16)) * inv, hi, true);
    *reinterpret_cast<uint2*>(qr + c) = uint2{(unsigned)lo, (unsigned)hi};
  }
}

extern "C" int vla_quant_fp8_rows(void* stream, const void* x, void* q, float* scale, int rows, int cols, int ldx, int ldq) {
  VLA_REQUIRE(x && q && scale && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldx >= cols && ldq >= cols && ldq % 16 == 0 &&
                  ((((uintptr_t)x) | ((uintptr_t)q)) & 15) == 0,
              "quant_fp8_rows: cols % 8, ldx % 8, ldq % 16, 16-B aligned pointers");
  hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (unsigned char*)q, scale, rows,
                     cols, (long long)ldx, (long long)ldq);
  VLA_CHECK_LAUNCH("quant_fp8_rows");
  return VLA_OK;
}

// same-type rows of whole 16-B chunks (the staged pixel tensor: one row of 9.6 MB): 16 B per lane
__global__ void copy2d_vec_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long rows, long long cols16, long long ld_src16,
                                  long long ld_dst16) {
  const long long total = rows * cols16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols16, c = i - r * cols16;
    dst[r * ld_dst16 + c] = src[r * ld_src16 + c];
  }
}

extern "C" int vla_copy2d(void* stream, const void* src, void* dst, long long rows, int cols, long long ld_src, long long ld_dst,
                          int src_dtype, int dst_dtype, int src_mod, int d_group, long long d_group_stride) {
  VLA_REQUIRE(src && dst && rows > 0 && cols > 0 && src_mod >= 0 && d_group >= 0, "copy2d: bad args");
  VLA_REQUIRE((src_dtype == 0 || src_dtype == 1) && (dst_dtype == 0 || dst_dtype == 1), "copy2d: dtype 0 (bf16) / 1 (f32)");
  hipStream_t st = (hipStream_t)stream;
  const int esz = src_dtype == 0 ? 2 : 4, per16 = 16 / esz;
  if (src_dtype == dst_dtype && src_mod == 0 && d_group == 0 && cols % per16 == 0 && ld_src % per16 == 0 && ld_dst % per16 == 0 &&
      ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
    const long long c16 = cols / per16;
    hipLaunchKernelGGL(copy2d_vec_kernel, GRID1D(rows * c16, 256), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, rows, c16, ld_src / per16,
                       ld_dst / per16);
    VLA_CHECK_LAUNCH("copy2d");
    return VLA_OK;
  }
  const dim3 g = GRID1D(rows * cols, 256);
#define CP(TS, TD) hipLaunchKernelGGL((copy2d_kernel<TS, TD>), g, dim3(256), 0, st, (const TS*)src, (TD*)dst, rows, cols, ld_src, ld_dst, src_mod, d_group, d_group_stride)
  if (src_dtype == 0 && dst_dtype == 0) CP(bf16_t, bf16_t);
  else if (src_dtype == 1 && dst_dtype == 0) CP(float, bf16_t);
  else if (src_dtype == 0 && dst_dtype == 1) CP(bf16_t, float);
  else CP(float, float);
#undef CP
  VLA_CHECK_LAUNCH("copy2d");
  return VLA_OK;
}

// A plain store kernel, not hipMemsetAsync: memset nodes recorded inside torch's stream capture were observed NOT to
// clear the buffer on replay (ROCm 7.2; the fp32 gradient accumulators kept the previous step's sums).
__global__ void fill_zero_kernel(uint4* __restrict__ p16, long long n16, unsigned char* __restrict__ tail, int ntail) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < n16; i += (long long)gridDim.x * blockDim.x) p16[i] = uint4{0, 0, 0, 0};
  if (i0 < ntail) tail[i0] = 0;
}

extern "C" int vla_fill_zero(void* stream, void* ptr, long long nbytes) {
  VLA_REQUIRE(ptr && nbytes > 0 && ((uintptr_t)ptr & 15) == 0, "fill_zero: null / empty / not 16-B aligned");
  const long long n16 = nbytes / 16;
  hipLaunchKernelGGL(fill_zero_kernel, GRID1D(n16 > 0 ? n16 : 1, 256), dim3(256), 0, (hipStream_t)stream, (uint4*)ptr, n16,
                     (unsigned char*)ptr + n16 * 16, (int)(nbytes - n16 * 16));
  VLA_CHECK_LAUNCH("fill_zero");
  return VLA_OK;
}

// ---- dropout on the input of a LoRA branch (peft Linear.forward: lora_B(lora_A(lora_dropout(x))), vla-scripts/finetune.py:110, 832-840) ----
// Counter-based mask: element (r, c) of a [rows, cols] tensor is kept iff a 16-bit uniform drawn from splitmix64(seed', 2 * chunk + half)
// is >= p * 65536, chunk = (r * cols + c) / 8 (four 16-bit draws per 64-bit hash, eight elements per thread); seed' = seed + step * odd
// constant with `step` read from DEVICE memory (a captured step draws fresh masks on every replay: vla_inc_i32 bumps it).  Forward and
// backward regenerate the same mask from (seed, step) - nothing is stored.  Kept values are scaled by 1 / (1 - p) in fp32 and rounded to
// bf16 once, as torch's dropout kernel does.  (torch's own Philox stream is not reproduced: parity with a peft run is statistical.)
__device__ __forceinline__ unsigned long long drop_hash(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <bool ADD>
__global__ void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long rows, int cols, long long ldx, long long ldy,
                               unsigned thr, float scale, unsigned long long seed, const int* __restrict__ step) {
  const int cpr = cols >> 3;
  const long long total = rows * cpr;
  const unsigned long long sd = seed + (unsigned long long)(step ? *step : 0) * 0xD1B54A32D192ED03ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    const unsigned long long h0 = drop_hash(sd, 2ull * (unsigned long long)i), h1 = drop_hash(sd, 2ull * (unsigned long long)i + 1);
    const uint4 xv = *reinterpret_cast<const uint4*>(x + r * ldx + c);
    const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
    uint4 yv = uint4{0, 0, 0, 0};
    if (ADD) yv = *reinterpret_cast<const uint4*>(y + r * ldy + c);
    unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long h = k < 2 ? h0 : h1;
      const bool k0 = ((unsigned)(h >> (32 * (k & 1))) & 0xffffu) >= thr, k1 = ((unsigned)(h >> (32 * (k & 1) + 16)) & 0xffffu) >= thr;
      const float a = k0 ? rbf(bf2f((bf16_t)(xw[k] & 0xffff)) * scale) : 0.f, b = k1 ? rbf(bf2f((bf16_t)(xw[k] >> 16)) * scale) : 0.f;
      if (ADD) yw[k] = pack2(bf2f((bf16_t)(yw[k] & 0xffff)) + a, bf2f((bf16_t)(yw[k] >> 16)) + b);
      else yw[k] = pack2(a, b);
    }
    *reinterpret_cast<uint4*>(y + r * ldy + c) = uint4{yw[0], yw[1], yw[2], yw[3]};
  }
}

static int dropout_launch(bool add, void* stream, const void* x, void* y, long long rows, int cols, long long ldx, long long ldy, float p,
                          unsigned long long seed, const int* step) {
  VLA_REQUIRE(x && y && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
              "dropout: null / empty / cols, row strides must be multiples of 8 and the tensors 16-B aligned");
  VLA_REQUIRE(p >= 0.f && p < 1.f, "dropout: p in [0, 1)");
  const unsigned thr = (unsigned)(p * 65536.f + 0.5f);
  const float scale = 1.f / (1.f - p);
  const long long total = rows * (cols / 8);
  if (add) hipLaunchKernelGGL(dropout_kernel<true>, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, rows, cols, ldx, ldy, thr, scale, seed, step);
  else hipLaunchKernelGGL(dropout_kernel<false>, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, rows, cols, ldx, ldy, thr, scale, seed, step);
  VLA_CHECK_LAUNCH("dropout");
  return VLA_OK;
}

extern "C" int vla_dropout_bf16(void* stream, const void* x, void* y, long long rows, int cols, long long ldx, long long ldy, float p,
                                unsigned long long seed, const int* step) {
  return dropout_launch(false, stream, x, y, rows, cols, ldx, ldy, p, seed, step);
}

extern "C" int vla_dropout_bwd_add_bf16(void* stream, const void* u, void* dx, long long rows, int cols, long long ldu, long long lddx, float p,
                                        unsigned long long seed, const int* step) {
  return dropout_launch(true, stream, u, dx, rows, cols, ldu, lddx, p, seed, step);
}

__global__ void inc_i32_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1; }

extern "C" int vla_inc_i32(void* stream, int* ptr) {
  VLA_REQUIRE(ptr, "inc_i32: null");
  hipLaunchKernelGGL(inc_i32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ptr);
  VLA_CHECK_LAUNCH("inc_i32");
  return VLA_OK;
}

// Index arrays of the action head for one batch (engine.Head): pos1[B,64] = text-coordinate positions of the action-query
// hidden states (-1: none).  gather[b, k] = b*S + Np + pos1 (k < 64), gather[b, 64] = -2 (proprio slot: leave the row alone);
// scatter[b, k] = b*(S-row0) + Np + pos1 - row0 or -1 (dead row / none), scatter[b, 64] = -1;
// guard = NaN if any sample's first action query (pos0[b,0] + Np, count > 0) lies before row0, else 0.
__global__ void head_index_prep_kernel(const int* __restrict__ pos1, const int* __restrict__ pos0, const int* __restrict__ cnt0,
                                       int* __restrict__ gather, int* __restrict__ scatter, float* __restrict__ guard,
                                       int B, int S, int Np, int row0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 65) {
    const int b = i / 65, k = i - b * 65;
    if (k == 64) {
      gather[i] = -2;
      scatter[i] = -1;
    } else {
      const int p = pos1[b * 64 + k];
      gather[i] = b * S + Np + p;
      const int loc = Np + p - row0;
      scatter[i] = (loc >= 0 && p >= 0) ? b * (S - row0) + loc : -1;
    }
  }
  if (i == 0 && guard) {
    bool bad = false;
    for (int b = 0; b < B; ++b) bad |= (row0 > 0) && ((cnt0[b] > 0 ? pos0[b * 64] + Np : 0) < row0);
    *guard = bad ? __int_as_float(0x7fc00000) : 0.f;
  }
}

extern "C" int vla_head_index_prep(void* stream, const int* pos1, const int* pos0, const int* cnt0, int* gather, int* scatter,
                                   float* guard, int B, int S, int Np, int row0) {
  VLA_REQUIRE(pos1 && pos0 && cnt0 && gather && scatter && B > 0 && S > row0 && row0 >= 0, "head_index_prep: bad args");
  hipLaunchKernelGGL(head_index_prep_kernel, dim3(nblk((long long)B * 65, 256)), dim3(256), 0, (hipStream_t)stream, pos1, pos0, cnt0, gather,
                     scatter, guard, B, S, Np, row0);
  VLA_CHECK_LAUNCH("head_index_prep");
  return VLA_OK;
}

// x[i] += *s  (the frozen live-row window's NaN guard folded into the reported loss without a host round trip)
__global__ void add_scalar_f32_kernel(float* __restrict__ x, const float* __restrict__ s, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += *s;
}

extern "C" int vla_add_scalar_f32(void* stream, float* x, const float* s, int n) {
  VLA_REQUIRE(x && s && n > 0, "add_scalar_f32: bad args");
  hipLaunchKernelGGL(add_scalar_f32_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x, s, n);
  VLA_CHECK_LAUNCH("add_scalar_f32");
  return VLA_OK;
}


// ---------------------------------------------------------------- embedding-table gradient (full fine-tune)
// d table[id] = sum over the token positions (b, j) with input_ids[b, j] == id (and no action query spliced in there) of
// dX0[b, row(j)], row(0) = 0, row(j) = Np + j - row0... here row0 = 0 (full backward): sequence row of token j is 0 for j = 0,
// Np + j otherwise.  One workgroup per token position; the FIRST position holding an id sums every position of that id in
// fp32, in position order, and writes the bf16 row once (deterministic, one rounding - torch's embedding_dense_backward also
// accumulates a row in the accumulate type); all other positions do nothing.  n = B * L is ~1.5 k: the O(n) scans are free.
__global__ void embed_grad_kernel(const bf16_t* __restrict__ dx, const long long* __restrict__ ids, const int* __restrict__ qidx,
                                  bf16_t* __restrict__ gtable, int B, int L, int Np, int D, int vocab) {
  const int t = blockIdx.x, n = B * L;
  if (qidx[t] >= 0) return;                       // slot overwritten by an action query: its gradient goes to action_queries
  long long id = ids[t];
  if (id < 0 || id >= vocab) id = 0;              // same clamp as the forward gather
  __shared__ int first;
  if (threadIdx.x == 0) {
    int f = 1;
    for (int u = 0; u < t && f; ++u) {
      long long iu = ids[u];
      if (iu < 0 || iu >= vocab) iu = 0;
      if (qidx[u] < 0 && iu == id) f = 0;
    }
    first = f;
  }
  __syncthreads();
  if (!first) return;
  const int S = L + Np;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float a = 0.f;
    for (int u = t; u < n; ++u) {
      long long iu = ids[u];
      if (iu < 0 || iu >= vocab) iu = 0;
      if (qidx[u] >= 0 || iu != id) continue;
      const int b = u / L, j = u - b * L;
      a += bf2f(dx[((long long)b * S + (j == 0 ? 0 : Np + j)) * D + c]);
    }
    gtable[id * D + c] = f2bf(a);
  }
}

extern "C" int vla_embed_grad(void* stream, const void* dx, const long long* ids, const int* qidx, void* grad_table, int B, int L,
                              int Np, int D, int vocab) {
  VLA_REQUIRE(dx && ids && qidx && grad_table && B > 0 && L > 0 && Np >= 0 && D > 0 && vocab > 0, "embed_grad: bad args");
  hipLaunchKernelGGL(embed_grad_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, ids, qidx, (bf16_t*)grad_table,
                     B, L, Np, D, vocab);
  VLA_CHECK_LAUNCH("embed_grad");
  return VLA_OK;
}


// ---------------------------------------------------------------- token cross-entropy
// HF causal-LM loss (transformers' ForCausalLMLoss as the reference's PrismaticVLM.forward reaches it through the LLM backbone,
// prismatic/models/vlms/prismatic.py:469-481): logits are bf16, upcast to fp32, shifted by one position, mean over the labels
// != -100.  One workgroup per logits row: row_loss = logsumexp(logits[row]) - logits[row, label]; loss_sum += row_loss,
// count += 1 (two f32 atomics per valid row); the caller divides.  labels = the SHIFTED targets, one per row (-100: ignore).
__global__ __launch_bounds__(256) void token_ce_kernel(const bf16_t* __restrict__ logits, long long ldl, const long long* __restrict__ labels,
                                                       int V, float* __restrict__ out2) {
  const long long row = blockIdx.x;
  const long long lab = labels[row];
  if (lab < 0 || lab >= V) return;                 // IGNORE_INDEX (-100)
  const bf16_t* x = logits + row * ldl;
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float m = -3.0e38f;
  for (int c = tid * 8; c < V; c += 256 * 8) {
    if (c + 8 <= V) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + c);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaxf(bf2f((bf16_t)(u[k] & 0xffff)), bf2f((bf16_t)(u[k] >> 16))));
    } else {
      for (int k = c; k < V; ++k) m = fmaxf(m, bf2f(x[k]));
    }
  }
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid * 8; c < V; c += 256 * 8) {
    if (c + 8 <= V) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + c);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) s += __expf(bf2f((bf16_t)(u[k] & 0xffff)) - m) + __expf(bf2f((bf16_t)(u[k] >> 16)) - m);
    } else {
      for (int k = c; k < V; ++k) s += __expf(bf2f(x[k]) - m);
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (tid == 0) {
    const float lse = m + __logf(red[0] + red[1] + red[2] + red[3]);
    atomicAdd(out2, lse - bf2f(x[lab]));
    atomicAdd(out2 + 1, 1.0f);
  }
}

extern "C" int vla_token_ce(void* stream, const void* logits, long long ld_logits, const long long* shifted_labels, int rows, int V,
                            float* loss_sum_and_count) {
  VLA_REQUIRE(logits && shifted_labels && loss_sum_and_count && rows > 0 && V > 0 && ld_logits % 8 == 0 && ((uintptr_t)logits & 15) == 0,
              "token_ce: bad args (row stride % 8, 16-B aligned logits)");
  hipLaunchKernelGGL(token_ce_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld_logits, shifted_labels, V,
                     loss_sum_and_count);
  VLA_CHECK_LAUNCH("token_ce");
  return VLA_OK;
}


// Backward of the above w.r.t. the logits (autograd of HF's shifted CE on ``logits.float()``): for a valid row
// dlogits[row, v] = bf16((softmax(float(logits[row]))[v] - [v == label]) * gscale / count), zeros for ignored rows; count is read from
// the device (loss_sum_and_count[1] of the forward: no host sync).  May run in place (dlogits == logits).  One workgroup per row.
__global__ __launch_bounds__(256) void token_ce_bwd_kernel(const bf16_t* __restrict__ logits, long long ldl, const long long* __restrict__ labels,
                                                           int V, const float* __restrict__ out2, float gscale, bf16_t* __restrict__ dl, long long ldd) {
  const long long row = blockIdx.x;
  const long long lab = labels[row];
  const bf16_t* x = logits + row * ldl;
  bf16_t* y = dl + row * ldd;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (lab < 0 || lab >= V) {                       // IGNORE_INDEX: no gradient
    for (int c = tid * 8; c < V; c += 256 * 8) {
      if (c + 8 <= V) *reinterpret_cast<uint4*>(y + c) = uint4{0, 0, 0, 0};
      else for (int k = c; k < V; ++k) y[k] = 0;
    }
    return;
  }
  __shared__ float red[8];
  float m = -3.0e38f;
  for (int c = tid * 8; c < V; c += 256 * 8) {
    if (c + 8 <= V) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + c);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaxf(bf2f((bf16_t)(u[k] & 0xffff)), bf2f((bf16_t)(u[k] >> 16))));
    } else {
      for (int k = c; k < V; ++k) m = fmaxf(m, bf2f(x[k]));
    }
  }
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid * 8; c < V; c += 256 * 8) {
    if (c + 8 <= V) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + c);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) s += __expf(bf2f((bf16_t)(u[k] & 0xffff)) - m) + __expf(bf2f((bf16_t)(u[k] >> 16)) - m);
    } else {
      for (int k = c; k < V; ++k) s += __expf(bf2f(x[k]) - m);
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + w] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  const float sc = gscale / out2[1];
  for (int c = tid * 8; c < V; c += 256 * 8) {
    if (c + 8 <= V) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + c);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
      unsigned o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float p0 = __expf(bf2f((bf16_t)(u[k] & 0xffff)) - m) * inv - (c + 2 * k == lab ? 1.f : 0.f);
        const float p1 = __expf(bf2f((bf16_t)(u[k] >> 16)) - m) * inv - (c + 2 * k + 1 == lab ? 1.f : 0.f);
        o[k] = pack2(p0 * sc, p1 * sc);
      }
      *reinterpret_cast<uint4*>(y + c) = uint4{o[0], o[1], o[2], o[3]};
    } else {
      for (int k = c; k < V; ++k) y[k] = f2bf((__expf(bf2f(x[k]) - m) * inv - (k == lab ? 1.f : 0.f)) * sc);
    }
  }
}

extern "C" int vla_token_ce_bwd(void* stream, const void* logits, long long ld_logits, const long long* shifted_labels, int rows, int V,
                                const float* loss_sum_and_count, float gscale, void* dlogits, long long ld_dlogits) {
  VLA_REQUIRE(logits && shifted_labels && loss_sum_and_count && dlogits && rows > 0 && V > 0 && ld_logits % 8 == 0 && ld_dlogits % 8 == 0 &&
                  ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) == 0,
              "token_ce_bwd: bad args (row strides % 8, 16-B aligned logits / dlogits)");
  hipLaunchKernelGGL(token_ce_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld_logits, shifted_labels, V,
                     loss_sum_and_count, gscale, (bf16_t*)dlogits, ld_dlogits);
  VLA_CHECK_LAUNCH("token_ce_bwd");
  return VLA_OK;
}


// ---------------------------------------------------------------- SwiGLU forward on interleaved pre-activations (LoRA path)
// h[m, 16t + c] = bf16(bf16(silu(g)) * u), g = GU[m, 32t + c], u = GU[m, 32t + 16 + c]: the product the fused GEMM epilogue forms;
// stand-alone because LoRA adds its low-rank deltas to the gate / up pre-activations BEFORE the activation (peft wraps each
// nn.Linear: vla-scripts/finetune.py:832-844).  8 h columns per thread.
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ h, long long nch, int I) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nch; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;                       // first h element of this chunk
    const long long m = e / I;
    const int c = (int)(e - m * I);                  // column in [0, I), multiple of 8
    const bf16_t* row = gu + m * 2 * I + (c >> 4) * 32 + (c & 15);
    const uint4 gv = *reinterpret_cast<const uint4*>(row), uv = *reinterpret_cast<const uint4*>(row + 16);
    const unsigned ga[4] = {gv.x, gv.y, gv.z, gv.w}, ua[4] = {uv.x, uv.y, uv.z, uv.w};
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g0 = bf2f((bf16_t)(ga[k] & 0xffff)), g1 = bf2f((bf16_t)(ga[k] >> 16));
      const float u0 = bf2f((bf16_t)(ua[k] & 0xffff)), u1 = bf2f((bf16_t)(ua[k] >> 16));
      o[k] = pack2(rbf(g0 * __builtin_amdgcn_rcpf(1.0f + __expf(-g0))) * u0, rbf(g1 * __builtin_amdgcn_rcpf(1.0f + __expf(-g1))) * u1);
    }
    *reinterpret_cast<uint4*>(h + e) = uint4{o[0], o[1], o[2], o[3]};
  }
}

extern "C" int vla_swiglu_fwd(void* stream, const void* gu, void* h, int M, int I) {
  VLA_REQUIRE(gu && h && M > 0 && I > 0 && I % 16 == 0, "swiglu_fwd: I%16");
  const long long nch = (long long)M * I / 8;
  hipLaunchKernelGGL(swiglu_fwd_kernel, GRID1D(nch, 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu, (bf16_t*)h, nch, I);
  VLA_CHECK_LAUNCH("swiglu_fwd");
  return VLA_OK;
}


// ---------------------------------------------------------------- image resize
// One 1-D pass of Pillow's 8-bit resampler (what PrismaticImageProcessor.apply_transform reaches through torchvision's
// TVF.resize(PIL image, BICUBIC, antialias=True), processing_prismatic.py:137): out[o, p, i] = clip8((2^21 + sum_k
// in[o, lo_p + k, i] * coef[p, k]) >> 22) on a [outer, len, inner] uint8 layout - horizontal pass: outer = B*H, inner = 3;
// vertical pass: outer = B, inner = out_w * 3.  Bounds / fixed-point taps come from the host (input_stage.pil_bicubic_coeffs,
// the same double arithmetic as Pillow's precompute_coeffs).  Bit-exact against PIL.Image.resize (tests).
__global__ void resample_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long outer, int in_len,
                                   int out_len, int inner, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize) {
  const long long total = outer * out_len * inner;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % inner);
    const int p = (int)((idx / inner) % out_len);
    const long long o = idx / ((long long)inner * out_len);
    const int lo = bounds[2 * p], cnt = bounds[2 * p + 1];
    const unsigned char* s = src + (o * in_len + lo) * inner + i;
    const int* k = coefs + (long long)p * ksize;
    int acc = 1 << 21;
    for (int t = 0; t < cnt; ++t) acc += (int)s[(long long)t * inner] * k[t];
    acc >>= 22;
    dst[idx] = (unsigned char)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
  }
}

extern "C" int vla_resample_u8(void* stream, const void* src, void* dst, long long outer, int in_len, int out_len, int inner,
                               const int* bounds, const int* coefs, int ksize) {
  VLA_REQUIRE(src && dst && bounds && coefs && outer > 0 && in_len > 0 && out_len > 0 && inner > 0 && ksize > 0, "resample_u8: bad args");
  hipLaunchKernelGGL(resample_u8_kernel, GRID1D(outer * out_len * inner, 256), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src,
                     (unsigned char*)dst, outer, in_len, out_len, inner, bounds, coefs, ksize);
  VLA_CHECK_LAUNCH("resample_u8");
  return VLA_OK;
}
