// LayerNorm / RMSNorm forward + backward for gfx950.  HBM-bound row kernels: one 64-lane wave per row, the row
// held in registers as packed bf16 (16-B loads, 8 elements per lane per chunk), fp32 statistics reduced with
// wavefront shuffles, 4 rows per 256-thread workgroup.
//
// Replaces nn.LayerNorm in timm Block (norm1/norm2, eps 1e-6), MLPResNet.layer_norm1/2 and ffn.0 in
// MLPResNetBlock(_Pro) (action_heads.py:96,108,306; eps 1e-5) and Qwen2RMSNorm (eps 1e-6).
#include "common.h"
#include "../../include/vla_native.h"

namespace {

template <int NCH>
__device__ __forceinline__ void load_row(const bf16_t* x, int cols, int lane, uint4 (&v)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    v[c] = (e < cols) ? *reinterpret_cast<const uint4*>(x + e) : uint4{0, 0, 0, 0};
  }
}
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


// Row-wise e4m3 quantisation of a normalised row that is still in registers (packed bf16, the values the bf16 output holds):
// the fused form of vla_quant_fp8_rows - same scale, same conversion, no second pass over the row.
template <int NCH>
__device__ __forceinline__ void quant_row_q8(const uint4 (&yo)[NCH], int cols, int lane, unsigned char* __restrict__ q, float* __restrict__ scale) {
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if ((c * 64 + lane) * 8 >= cols) continue;
    float f[8];
    unpack8(yo[c], f);
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
  }
  amax = wave_max(amax);
  const float inv = amax > 0.f ? __fdiv_rn(448.f, amax) : 1.f;
  if (lane == 0) *scale = amax > 0.f ? __fdiv_rn(amax, 448.f) : 1.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float f[8];
    unpack8(yo[c], f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
    *reinterpret_cast<uint2*>(q + e) = uint2{(unsigned)lo, (unsigned)hi};
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                            float* __restrict__ stats, int rows, int cols, int ldx,
                                                            int ldy, float eps, unsigned char* __restrict__ q8 = nullptr,
                                                            float* __restrict__ qscale = nullptr, int ldq = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  // the weight / bias segments are requested WITH the row: fetched where they are used, each chunk paid its own dependent
  // L2 round trip behind the two reductions (three in a row for 1152 columns: ~2 us of a 10-us kernel)
  uint4 v[NCH], vw[NCH], vb[NCH];
  load_row<NCH>(x + (long long)row * ldx, cols, lane, v);
  load_row<NCH>(w, cols, lane, vw);
  load_row<NCH>(b, cols, lane, vb);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float f[8];
    unpack8(v[c], f);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k];
  }
  const float mean = wave_sum(s) / cols;
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float f[8];
    unpack8(v[c], f);
    const int e = (c * 64 + lane) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = (e + k < cols) ? f[k] - mean : 0.f;
      s2 += d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(s2) / cols + eps);
  if (stats && lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float f[8], fw[8], fb[8], o[8];
    unpack8(v[c], f);
    unpack8(vw[c], fw);
    unpack8(vb[c], fb);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f[k] - mean) * rstd * fw[k] + fb[k];
    v[c] = pack8(o);                                           // (the input chunk is dead: keep the output for the fp8 pass)
    if (y) *reinterpret_cast<uint4*>(y + (long long)row * ldy + e) = v[c];
  }
  if (q8) quant_row_q8<NCH>(v, cols, lane, q8 + (long long)row * ldq, qscale + row);
}

template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w,
                                                               const float* __restrict__ stats, bf16_t* __restrict__ dx,
                                                               int rows, int cols, int ldx, int lddy, int lddx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint4 vx[NCH], vd[NCH], vw[NCH];
  load_row<NCH>(x + (long long)row * ldx, cols, lane, vx);
  load_row<NCH>(dy + (long long)row * lddy, cols, lane, vd);
  load_row<NCH>(w, cols, lane, vw);
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float fx[8], fd[8], fw[8];
    unpack8(vx[c], fx);
    unpack8(vd[c], fd);
    unpack8(vw[c], fw);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float g = fd[k] * fw[k], xh = (fx[k] - mean) * rstd;
      s1 += g;
      s2 += g * xh;
    }
  }
  s1 = wave_sum(s1) / cols;
  s2 = wave_sum(s2) / cols;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float fx[8], fd[8], fw[8], o[8];
    unpack8(vx[c], fx);
    unpack8(vd[c], fd);
    unpack8(vw[c], fw);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (fx[k] - mean) * rstd;
      o[k] = rstd * (fd[k] * fw[k] - s1 - xh * s2);
    }
    *reinterpret_cast<uint4*>(dx + (long long)row * lddx + e) = pack8(o);
  }
}

// dw[c] += sum_rows dy*xhat, db[c] += sum_rows dy.  grid (ceil(cols/256), row_splits)
__global__ __launch_bounds__(256) void layernorm_bwd_wb_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               const float* __restrict__ stats, float* __restrict__ dw,
                                                               float* __restrict__ db, int rows, int cols, int ldx,
                                                               int lddy, int rows_per) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float aw = 0.f, ab = 0.f;
#pragma unroll 4
  for (int r = r0; r < r1; ++r) {
    const float d = bf2f(dy[(long long)r * lddy + c]);
    const float xh = (bf2f(x[(long long)r * ldx + c]) - stats[2 * r]) * stats[2 * r + 1];
    aw += d * xh;
    ab += d;
  }
  if (dw) atomicAdd(dw + c, aw);
  if (db) atomicAdd(db + c, ab);
}

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out, int rows,
                                                          int cols, float eps, unsigned char* __restrict__ q8 = nullptr,
                                                          float* __restrict__ qscale = nullptr, int ldq = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint4 v[NCH], vw[NCH];               // (weight segments requested with the row: see layernorm_fwd_kernel)
  load_row<NCH>(x + (long long)row * cols, cols, lane, v);
  load_row<NCH>(w, cols, lane, vw);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float f[8];
    unpack8(v[c], f);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k] * f[k];
  }
  const float rstd = rsqrtf(wave_sum(s) / cols + eps);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float f[8], fw[8], o[8];
    unpack8(v[c], f);
    unpack8(vw[c], fw);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = fw[k] * rbf(f[k] * rstd);  // two rounding points, as Qwen2RMSNorm in bf16
    v[c] = pack8(o);
    if (y) *reinterpret_cast<uint4*>(y + (long long)row * cols + e) = v[c];
  }
  if (q8) quant_row_q8<NCH>(v, cols, lane, q8 + (long long)row * ldq, qscale + row);
}

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                          int rows, int cols, int xg, int xgr, int xr0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long xrow = xg > 0 ? (long long)(row / xg) * xgr + xr0 + (row % xg) : row;   // row of x / rstd
  uint4 vx[NCH], vd[NCH], vw[NCH], vr[NCH];
  load_row<NCH>(x + xrow * cols, cols, lane, vx);
  load_row<NCH>(dy + (long long)row * cols, cols, lane, vd);
  load_row<NCH>(w, cols, lane, vw);
  if (dres) load_row<NCH>(dres + (long long)row * cols, cols, lane, vr);
  const float rstd = rstd_in[xrow];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float fx[8], fd[8], fw[8];
    unpack8(vx[c], fx);
    unpack8(vd[c], fd);
    unpack8(vw[c], fw);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += fd[k] * fw[k] * fx[k] * rstd;
  }
  s = wave_sum(s) / cols;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e >= cols) continue;
    float fx[8], fd[8], fw[8], o[8];
    unpack8(vx[c], fx);
    unpack8(vd[c], fd);
    unpack8(vw[c], fw);
    float fr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (dres) unpack8(vr[c], fr);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = fr[k] + rstd * (fd[k] * fw[k] - fx[k] * rstd * s);
    *reinterpret_cast<uint4*>(dx + (long long)row * cols + e) = pack8(o);
  }
}

inline int nch_for(int cols) { return (cols + 511) / 512; }

#define DISPATCH_NCH(n, CALL)                    \
  if (n <= 1) { CALL(1); }                       \
  else if (n <= 2) { CALL(2); }                  \
  else if (n <= 3) { CALL(3); }                  \
  else if (n <= 4) { CALL(4); }                  \
  else if (n <= 8) { CALL(8); }                  \
  else if (n <= 16) { CALL(16); }                \
  else { CALL(24); }

}  // namespace

extern "C" int vla_layernorm_fwd(void* stream, const void* x, const void* w, const void* b, void* y, float* stats,
                                 int rows, int cols, int ldx, int ldy, float eps) {
  VLA_REQUIRE(x && w && b && y && rows > 0 && cols > 0, "layernorm_fwd: null/empty");
  VLA_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && cols <= 12288, "layernorm_fwd: cols%8, ld%8, cols<=12288");
  VLA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0, "layernorm_fwd: 16-B alignment");
  const int n = nch_for(cols);
  dim3 grid((rows + 3) / 4);
#define CALL(N) hipLaunchKernelGGL(layernorm_fwd_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, \
                                   (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, stats, rows, cols, ldx, ldy, eps)
  DISPATCH_NCH(n, CALL)
#undef CALL
  VLA_CHECK_LAUNCH("layernorm_fwd");
  return VLA_OK;
}

extern "C" int vla_layernorm_fwd_q8(void* stream, const void* x, const void* w, const void* b, void* y, float* stats, void* q8, float* qscale,
                                    int rows, int cols, int ldx, int ldy, int ldq, float eps) {
  VLA_REQUIRE(x && w && b && q8 && qscale && rows > 0 && cols > 0, "layernorm_fwd_q8: null/empty");
  VLA_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && (!y || ldy % 8 == 0) && ldq % 16 == 0 && ldq >= cols && cols <= 12288, "layernorm_fwd_q8: cols%8, ld%8, ldq%16");
  VLA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b | (uintptr_t)q8) & 15) == 0, "layernorm_fwd_q8: 16-B alignment");
  const int n = nch_for(cols);
  dim3 grid((rows + 3) / 4);
#define CALL(N) hipLaunchKernelGGL(layernorm_fwd_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, \
                                   (const bf16_t*)b, (bf16_t*)y, stats, rows, cols, ldx, ldy, eps, (unsigned char*)q8, qscale, ldq)
  DISPATCH_NCH(n, CALL)
#undef CALL
  VLA_CHECK_LAUNCH("layernorm_fwd_q8");
  return VLA_OK;
}

extern "C" int vla_layernorm_bwd(void* stream, const void* dy, const void* x, const void* w, const float* stats, void* dx,
                                 float* dw, float* db, int rows, int cols, int ldx, int lddy, int lddx) {
  VLA_REQUIRE(dy && x && w && stats && rows > 0 && cols > 0, "layernorm_bwd: null/empty");
  VLA_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && cols <= 12288, "layernorm_bwd: cols%8, ld%8");
  if (dx) {
    VLA_REQUIRE(lddx % 8 == 0, "layernorm_bwd: lddx%8");
    const int n = nch_for(cols);
    dim3 grid((rows + 3) / 4);
#define CALL(N) hipLaunchKernelGGL(layernorm_bwd_dx_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, \
                                   (const bf16_t*)x, (const bf16_t*)w, stats, (bf16_t*)dx, rows, cols, ldx, lddy, lddx)
    DISPATCH_NCH(n, CALL)
#undef CALL
    VLA_CHECK_LAUNCH("layernorm_bwd_dx");
  }
  if (dw || db) {
    // short serial row loops (the kernel sits on the action head's backward critical chain with rows = 256):
    // rows/32 per block, clamped to [8, 64]; partial sums meet through fp32 atomics
    const int rows_per = rows / 32 < 8 ? 8 : (rows / 32 > 64 ? 64 : rows / 32);
    dim3 grid((cols + 255) / 256, (rows + rows_per - 1) / rows_per);
    hipLaunchKernelGGL(layernorm_bwd_wb_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, stats, dw, db, rows, cols, ldx, lddy, rows_per);
    VLA_CHECK_LAUNCH("layernorm_bwd_wb");
  }
  return VLA_OK;
}

extern "C" int vla_rmsnorm_fwd(void* stream, const void* x, const void* w, void* y, float* rstd, int rows, int cols,
                               float eps) {
  VLA_REQUIRE(x && w && y && rows > 0 && cols > 0, "rmsnorm_fwd: null/empty");
  VLA_REQUIRE(cols % 8 == 0 && cols <= 12288, "rmsnorm_fwd: cols%8==0, cols<=12288");
  VLA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15) == 0, "rmsnorm_fwd: 16-B alignment");
  const int n = nch_for(cols);
  dim3 grid((rows + 3) / 4);
#define CALL(N) hipLaunchKernelGGL(rmsnorm_fwd_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, \
                                   (const bf16_t*)w, (bf16_t*)y, rstd, rows, cols, eps)
  DISPATCH_NCH(n, CALL)
#undef CALL
  VLA_CHECK_LAUNCH("rmsnorm_fwd");
  return VLA_OK;
}

extern "C" int vla_rmsnorm_fwd_q8(void* stream, const void* x, const void* w, void* y, float* rstd, void* q8, float* qscale, int rows, int cols,
                                  int ldq, float eps) {
  VLA_REQUIRE(x && w && q8 && qscale && rows > 0 && cols > 0, "rmsnorm_fwd_q8: null/empty");
  VLA_REQUIRE(cols % 8 == 0 && cols <= 12288 && ldq % 16 == 0 && ldq >= cols, "rmsnorm_fwd_q8: cols%8==0, cols<=12288, ldq%16");
  VLA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)q8) & 15) == 0, "rmsnorm_fwd_q8: 16-B alignment");
  const int n = nch_for(cols);
  dim3 grid((rows + 3) / 4);
#define CALL(N) hipLaunchKernelGGL(rmsnorm_fwd_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, \
                                   rstd, rows, cols, eps, (unsigned char*)q8, qscale, ldq)
  DISPATCH_NCH(n, CALL)
#undef CALL
  VLA_CHECK_LAUNCH("rmsnorm_fwd_q8");
  return VLA_OK;
}

extern "C" int vla_rmsnorm_bwd(void* stream, const void* dy, const void* x, const void* w, const float* rstd,
                               const void* dres, void* dx, int rows, int cols, int x_group, int x_group_rows, int x_row0) {
  VLA_REQUIRE(dy && x && w && rstd && dx && rows > 0 && cols > 0, "rmsnorm_bwd: null/empty");
  VLA_REQUIRE(x_group >= 0 && (x_group == 0 || (x_row0 >= 0 && x_row0 + x_group <= x_group_rows)), "rmsnorm_bwd: bad row window");
  VLA_REQUIRE(cols % 8 == 0 && cols <= 12288, "rmsnorm_bwd: cols%8==0");
  const int n = nch_for(cols);
  dim3 grid((rows + 3) / 4);
#define CALL(N) hipLaunchKernelGGL(rmsnorm_bwd_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, \
                                   (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, rows, cols, \
                                   x_group, x_group_rows, x_row0)
  DISPATCH_NCH(n, CALL)
#undef CALL
  VLA_CHECK_LAUNCH("rmsnorm_bwd");
  return VLA_OK;
}


// ---------------------------------------------------------------- RMSNorm weight gradient (full fine-tune, BASELINE config 4)
// Qwen2RMSNorm: y = w * bf16(x * rstd)  ->  dw[c] += sum_rows dy[r, c] * bf16(x[r, c] * rstd[r])   (fp32 accumulator, += ).
// grid (ceil(cols / 256), row splits): same shape as layernorm_bwd_wb_kernel.
__global__ __launch_bounds__(256) void rmsnorm_dw_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                         const float* __restrict__ rstd, float* __restrict__ dw, int rows, int cols,
                                                         int rows_per) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float a = 0.f;
#pragma unroll 4
  for (int r = r0; r < r1; ++r) a += bf2f(dy[(long long)r * cols + c]) * rbf(bf2f(x[(long long)r * cols + c]) * rstd[r]);
  atomicAdd(dw + c, a);
}

extern "C" int vla_rmsnorm_dw(void* stream, const void* dy, const void* x, const float* rstd, float* dw, int rows, int cols) {
  VLA_REQUIRE(dy && x && rstd && dw && rows > 0 && cols > 0, "rmsnorm_dw: null/empty");
  const int rows_per = 64;
  dim3 grid((cols + 255) / 256, (rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(rmsnorm_dw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, rstd, dw, rows, cols, rows_per);
  VLA_CHECK_LAUNCH("rmsnorm_dw");
  return VLA_OK;
}
