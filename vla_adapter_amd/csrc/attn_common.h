// Shared helpers of the MFMA attention kernels (attention.hip, head_attn_mfma.hip): 32x32x16 bf16 MFMA wrappers,
// accumulator <-> operand conversions and the transposed LDS fragment read (ds_read_b64_tr_b16).
#pragma once
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs (HIP's uint4 struct did not)

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (args <= 0 here)
// accumulator register -> row of the 32x32 tile (column = lane & 31)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Transposed fragment: rows R0..R0+3 and R0+8..R0+11 of a row-major [rows][ld] bf16 LDS tile, column block
// c0..c0+15 per 16-lane group -> lane i of the group receives column c0+i of those 8 rows (ds_read_b64_tr_b16).
// Used as the A operand (rows = tile columns) of a 32x32x16 MFMA whose k index runs over the tile's rows with the
// permutation  j -> 16s + 8(j>>2) + 4h + (j&3)  that matches pack_acc() below.
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int ld, int s, int col0, int lane) {
  const int h = lane >> 5, gi = (lane >> 4) & 1, i = lane & 15;
  const bf16_t* p0 = tile + (16 * s + 4 * h + (i >> 2)) * ld + col0 + 16 * gi + 4 * (i & 3);
  bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p0);
  bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p0 + 8 * ld));
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// accumulator registers 8s..8s+7 -> bf16 fragment for k-step s (B operand of A.X / A operand of X^T.B)
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(x[8 * s + j]);
  return r;
}

