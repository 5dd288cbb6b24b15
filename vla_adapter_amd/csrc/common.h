// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  One wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8;     // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;      // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;    // 32x32 accumulator

#define VLA_OK 0
#define VLA_ERR_ARG (-1)       // bad shape / alignment / null pointer
#define VLA_ERR_LAUNCH (-2)    // hipGetLastError() after launch
#define VLA_ERR_UNSUPPORTED (-3)

extern "C" void vla_set_error(const char* msg);

#define VLA_REQUIRE(cond, msg)            \
  do {                                    \
    if (!(cond)) {                        \
      vla_set_error(msg);                 \
      return VLA_ERR_ARG;                 \
    }                                     \
  } while (0)

#define VLA_CHECK_LAUNCH(name)                          \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) {                            \
      vla_set_error(name ": launch failed");            \
      return VLA_ERR_LAUNCH;                            \
    }                                                   \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }  // bf16 rounding point
__device__ __forceinline__ unsigned pack2(float lo, float hi) {   // ONE v_cvt_pk_bf16_f32 (two RNE conversions, lo in bits 0..15)
  typedef float f32x2_v __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
  const bf16x2_v b = __builtin_convertvector(f32x2_v{lo, hi}, bf16x2_v);
  return __builtin_bit_cast(unsigned, b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below a bf16 ulp): one v_exp + one v_rcp instead of the
// ~25-instruction libm erff - the GELU epilogue of the ViT fc1 GEMM evaluates it 64x per thread per tile.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
// gelu(x) = x * Phi(x) with Phi through the same A&S erf, folded: for z = |x|/sqrt2, t = 1/(1 + p z),
// erf(z) = 1 - poly(t) exp(-z^2)  =>  gelu(x) = max(x, 0) - |x| * (poly(t)/2) * exp(-x^2/2)   (both signs of x; no
// 1 - (1 - eps) cancellation on the negative side).  11 VALU + v_rcp + v_exp per element - the fc1 epilogue of the ViT
// evaluates it 32x per thread per tile, right on the tile's tail.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);       // exp(-x^2/2)
  return __builtin_fmaf(-ax * poly, e, fmaxf(x, 0.f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float c = 0.3989422804014327f;  // 1/sqrt(2 pi)
  return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f)) + x * c * __expf(-0.5f * x * x);
}
// 0.5 x (1 + tanh u) = x * sigmoid(2u), u = k (x + 0.044715 x^3): one v_exp + one v_rcp instead of libm tanhf
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u2 = x * __builtin_fmaf(x * x, 0.044715f * 1.5957691216057308f, 1.5957691216057308f);   // 2u
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u2 * -1.4426950408889634f));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Zero `bytes` (a multiple of 16, 16-B aligned) of LDS with 16-B stores, strided over `n` threads.  The attention kernels cleared
// their tile images with one ds_write_b16 per element: 136 stores per lane and tile pair in the head kernels - 8.3k of
// head_fwd_mfma's 32k cycles (tools/diag/hf_stamps.py, round 3).
__device__ __forceinline__ void lds_zero16(void* p, int bytes, int idx, int n) {
  for (int i = idx * 16; i < bytes; i += n * 16) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p) + i) = uint4{0, 0, 0, 0};
}

// async global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
