// Kernel-side parameter block shared by the GEMM translation units (gemm.hip: 128-row tiles, gemm256.hip: the 256x256
// 8-phase kernel).  Filled by vla_gemm_bf16_nt from the public vla_gemm_desc.
#pragma once
#include "common.h"

struct GemmP {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  const bf16_t* bias; const bf16_t* R; bf16_t* C2;
  int M, N, K, lda, ldb, ldc, ldr, ldc2, res_mod, act;
  long long sA, sB, sC, sR, sC2, sBias;
  int tiles_n, ntiles;
  int batch;                                // gemm256.hip: batches in the flattened tile list
  float alpha;
  int gA, gC, gR; long long sgA, sgC, sgR;  // row-group addressing: row r -> (r / g) * sg + (r % g) * ld
  int c_live_mod, c_live_from;              // C rows with (m % c_live_mod) < c_live_from are not stored
  float* ws;                                // split-K: fp32 [M, N] accumulator (blockIdx.z = K slice), finalised by a second kernel
  int Ktot;                                 // split-K: the whole contraction length (slice z covers [z K, min((z + 1) K, Ktot)); 0 = no slices
  int bias_post;                            // 1: round alpha * acc to bf16 before adding the bias (torch CPU Linear on a strided input)
  int rope_mode, rope_T, rope_dh, rope_cols; const float* rope_cos; const float* rope_sin;
  const float* scaleA; const float* scaleB;   // fp8 operands (A, B point at OCP e4m3 bytes): per-row dequantisation scales [M], [N]
  int stagger;                              // gemm256.hip: start delay (units of 1024 cycles) of the workgroups that walk one tile fewer than the others (0 = none)
  const bf16_t* A2; const bf16_t* B2; int K2, lda2, ldb2;   // K extension (gemm.hip EXT): C = epilogue(A . B^T + A2 . B2^T)
};

// gemm256.hip: 256x256x64 tile, 8 waves, staggered 8-phase schedule.  epi: 0 plain, 1 SwiGLU forward, 2 SwiGLU backward.
int vla_gemm256_launch(const GemmP& p, int epi, int batch, hipStream_t st);
// gemm_skinny.hip: small-output products (tall-skinny N = 64 / 128 / 192, or - under the latency hint - M <= 512; K < 2048; bias / activation / residual /
// interleaved-RoPE epilogue): 1 = launched there, 0 = not its shape.  `p` must be completely filled.
int vla_gemm_skinny_try(const GemmP& p, bool simple_addressing, bool latency_hint, hipStream_t st);
int vla_num_cus();      // compute units of the current device (cached)
