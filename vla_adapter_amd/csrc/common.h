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
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
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
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float c = 0.3989422804014327f;  // 1/sqrt(2 pi)
  return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f)) + x * c * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// async global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
