// Layout probes: tiny kernels that dump what each lane receives from ds_read_b64_tr_b16 and where each MFMA
// accumulator element lands, so tests/test_probe.py can assert the lane maps the kernels rely on
// (cdna_hip_programming.md section 3) on the actual device instead of trusting documentation.
#include "common.h"
#include "../../include/vla_native.h"

namespace {
__global__ void probe_tr_kernel(short* out) {
  // LDS tile [16 rows][64 cols] of int16 = row*256 + col ; each lane issues one tr read at &tile[(i>>2)][16*g + 4*(i&3)]
  __shared__ __attribute__((aligned(16))) short t[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) t[i] = (short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  const short* p = t + (i >> 2) * 64 + 16 * g + 4 * (i & 3);
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
  for (int k = 0; k < 4; ++k) out[lane * 4 + k] = v[k];
}
// C = A.B with A[i][k] = (i==k) (identity-like, 32x16 / 16x32) and B[k][j] = 100*k + j (asymmetric):
// C[i][j] = B[i][j] for i < K -> reveals the accumulator (reg, lane) -> (row, col) map.
__global__ void probe_mfma32_kernel(float* out) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * h + j;
    a[j] = (short)f2bf(r == k ? 1.f : 0.f);        // A[row r][k]
    b[j] = (short)f2bf((float)(8 * k + (r & 7)) ); // B[k][col r] = 8k + (col&7)  (exact in bf16: < 256)
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
__global__ void probe_mfma16_kernel(float* out) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j;
    a[j] = (short)f2bf(r == k ? 1.f : 0.f);
    b[j] = (short)f2bf((float)(8 * k + (r & 7)));
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}
// One v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit block scales) on operands given PER LANE (8 dwords = 32 fp8 each):
// tests/test_kernels_gpu.py derives the (lane, byte) -> (row, k) map of the fp8 GEMM's fragments from it.
typedef int v8i_probe __attribute__((ext_vector_type(8)));
__global__ void probe_mfma_f8_kernel(const v8i_probe* a, const v8i_probe* b, float* out) {
  const int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[lane], b[lane], c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}
}  // namespace

extern "C" int vla_probe_mfma_f8(void* stream, const void* a /*[64][32] fp8*/, const void* b /*[64][32] fp8*/, float* out /*[64*4]*/) {
  VLA_REQUIRE(a && b && out, "probe_f8: null");
  hipLaunchKernelGGL(probe_mfma_f8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const v8i_probe*)a, (const v8i_probe*)b, out);
  VLA_CHECK_LAUNCH("probe_f8");
  return VLA_OK;
}

/* Test-only entry points (not part of the product ABI; declared in tests via ctypes). */
extern "C" int vla_probe_layouts(void* stream, short* tr_out /*[64*4]*/, float* mfma32_out /*[64*16]*/, float* mfma16_out /*[64*4]*/) {
  VLA_REQUIRE(tr_out && mfma32_out && mfma16_out, "probe: null");
  hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tr_out);
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mfma32_out);
  hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mfma16_out);
  VLA_CHECK_LAUNCH("probe");
  return VLA_OK;
}
