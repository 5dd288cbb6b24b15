// Parameter block shared by the VALU (head_attn.hip) and MFMA (head_attn_mfma.hip) action-head attention kernels.
#pragma once
#include "common.h"

struct HP {
  const bf16_t* q; const bf16_t* ks; const bf16_t* vs; const bf16_t* ka; const bf16_t* va; const bf16_t* kt; const bf16_t* vt;
  const bf16_t* gate; bf16_t* out; float* probs;
  int B, T, Ka, Kt, H, dh, ld_q, ld_self, ld_adp, ld_task, ld_out;
  const bf16_t* dout; bf16_t* dq; bf16_t* dks; bf16_t* dvs; bf16_t* dka; bf16_t* dva; bf16_t* dkt; bf16_t* dvt; float* dgate;
  const float* rope_cos; const float* rope_sin;   // optional [>= max(T,Ka,Kt), dh]: fold the RoPE transpose into dq / dk
  float* ws; long long ws_floats;                 // optional backward workspace (tile-uniform MFMA backward: dQ / gate partials)
  int ref_softmax;                                // forward: bf16 weights rounded after normalisation (two passes), as ATen's bf16 softmax
};

bool head_attn_mfma_supported(const HP& p);
void head_attn_mfma_fwd(const HP& p, hipStream_t st);
void head_attn_mfma_bwd(const HP& p, hipStream_t st);
