// bf16 NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)   (both operands K-contiguous, the
// nn.Linear layout; fp32 accumulation on v_mfma_f32_16x16x32_bf16).
//
// Replaces every cuBLAS call the reference reaches through nn.Linear: timm ViT qkv/proj/fc1/fc2,
// PrismaticProjector (modeling_prismatic.py:261-273), Qwen2 q/k/v/o/gate/up/down, the action head's Linears
// (action_heads.py:337-410), and - with pre-transposed operands - their dX / dW products.
//
// Structure (cdna_hip_programming.md section 5): 128x128 block tile, BK=64, 4 waves (2x2), each wave a 64x64
// sub-tile = 4x4 MFMA 16x16 tiles; A/B tiles stream HBM->LDS with global_load_lds_dwordx4 (1 KiB per
// wave-instruction) into a double buffer; LDS image is linear with the 16-B-chunk XOR swizzle applied on the
// SOURCE address and on the fragment read (rule 21); one barrier per K-tile, next tile's loads in flight under
// the current tile's MFMAs.  Operands are swapped at the MFMA (A-operand := B rows) so each lane owns 4
// consecutive n of one row m: the epilogue stages the wave's tile through LDS and stores whole 128-B row
// segments with 16-B lanes.
//
// Epilogue rounding points follow the reference under bf16 autocast: bf16(acc+bias) -> act -> bf16 -> (+residual) -> bf16.
#include "common.h"
#include "../../include/vla_native.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // double buffer = 64 KiB
constexpr int EPI_STRIDE = 144;                // bytes per staged row (64 bf16 + 16 B pad, keeps 16-B alignment)

struct GemmP {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  const bf16_t* bias; const bf16_t* R; bf16_t* C2;
  int M, N, K, lda, ldb, ldc, ldr, ldc2, res_mod, act;
  long long sA, sB, sC, sR, sC2, sBias;
  int tiles_n, ntiles;
  float alpha;
  int gA, gC; long long sgA, sgC;  // row-group addressing: row r -> (r / g) * sg + (r % g) * ld
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case VLA_ACT_GELU: return rbf(gelu_erf(v));
    case VLA_ACT_RELU: return fmaxf(v, 0.f);
    case VLA_ACT_GELU_TANH: return rbf(gelu_tanh(v));
    default: return v;
  }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
  // contiguous run of tiles so neighbouring tiles (same A row-panel) hit the same L2.
  const int nwg = p.ntiles, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int bm = swz / p.tiles_n, bn = swz - bm * p.tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;
  const bf16_t* Ab = p.A + (long long)z * p.sA;
  const bf16_t* Bb = p.B + (long long)z * p.sB;

  // ---- staging addresses: piece pc = wid*4+i covers LDS rows 8pc..8pc+7; lane -> row 8pc+(lane>>3),
  //      LDS chunk lane&7 holds global chunk (lane&7)^(row&7)
  const int kc = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
  const bf16_t* pa[4];
  const bf16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wid * 4 + i) * 8 + (lane >> 3);
    const int ra = min(m0 + row, p.M - 1), rb = min(n0 + row, p.N - 1);
    pa[i] = Ab + (p.gA > 0 ? (long long)(ra / p.gA) * p.sgA + (long long)(ra % p.gA) * p.lda : (long long)ra * p.lda) + kc;
    pb[i] = Bb + (long long)rb * p.ldb + kc;
  }
  auto stage = [&](int buf, int k0) {
    char* base = smem + buf * BUF_BYTES + wid * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(pa[i] + k0, base + i * 1024);
      glds16(pb[i] + k0, base + TILE_BYTES + i * 1024);
    }
  };

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside an operand tile, per k-step s: row*128 + ((4s + (lane>>4)) ^ (lane&7))*16
  const int frow = lane & 15;
  int foff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) foff[s] = frow * 128 + (((4 * s + (lane >> 4)) ^ (lane & 7)) << 4);

  const int nt = p.K / BK;
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t landed for every wave; every wave is done reading buffer (t+1)&1
    if (t + 1 < nt) stage((t + 1) & 1, (t + 1) * BK);
    const char* sa = smem + (t & 1) * BUF_BYTES + wr * 64 * 128;
    const char* sb = smem + (t & 1) * BUF_BYTES + TILE_BYTES + wc * 64 * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fm[4], fn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fm[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 128 + foff[s]);
        fn[i] = *reinterpret_cast<const bf16x8*>(sb + i * 16 * 128 + foff[s]);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[ni], fm[mi], acc[ni][mi], 0, 0, 0);
    }
  }
  __syncthreads();  // all waves done with the operand tiles before the staging regions are overwritten

  // ---------------- epilogue ----------------
  // lane owns, for tile (ni, mi): m = 16mi + (lane&15), n = 16ni + 4(lane>>4) + {0..3}
  const int wm0 = m0 + wr * 64, wn0 = n0 + wc * 64;
  const int lq = lane >> 4, lr = lane & 15;
  const bf16_t* bias = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
  char* reg = smem + wid * (64 * EPI_STRIDE);

  float bv[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wn0 + ni * 16 + lq * 4 + j;
      bv[ni][j] = (bias && n < p.N) ? bf2f(bias[n]) : 0.f;
    }

  if (p.act == VLA_ACT_SWIGLU) {
    // columns interleaved in 16s: tiles ni=0,2 are gate, ni=1,3 the matching up columns
    bf16_t* C2 = p.C2 + (long long)z * p.sC2;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = wm0 + mi * 16 + lr;
        float h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = rbf(acc[2 * pr][mi][j] * p.alpha + bv[2 * pr][j]);
          const float u = rbf(acc[2 * pr + 1][mi][j] * p.alpha + bv[2 * pr + 1][j]);
          acc[2 * pr][mi][j] = g;
          acc[2 * pr + 1][mi][j] = u;
          h[j] = rbf(silu(g)) * u;
        }
        const int hc = (wn0 >> 1) + pr * 16 + lq * 4;
        if (m < p.M && hc + 3 < (p.N >> 1)) {
          uint2 o = {pack2(h[0], h[1]), pack2(h[2], h[3])};
          *reinterpret_cast<uint2*>(C2 + (long long)m * p.ldc2 + hc) = o;
        }
      }
    if (p.C == nullptr) return;
  } else {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[ni][mi][j] * p.alpha + bv[ni][j];
          if (p.act != VLA_ACT_NONE) v = apply_act(rbf(v), p.act);
          acc[ni][mi][j] = v;
        }
  }

  // stage the wave's 64x64 tile (bf16) through its private LDS region, then store 16 B per lane
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      uint2 o = {pack2(acc[ni][mi][0], acc[ni][mi][1]), pack2(acc[ni][mi][2], acc[ni][mi][3])};
      *reinterpret_cast<uint2*>(reg + (mi * 16 + lr) * EPI_STRIDE + (ni * 16 + lq * 4) * 2) = o;
    }
  bf16_t* Cb = p.C + (long long)z * p.sC;
  const bf16_t* Rb = p.R ? p.R + (long long)z * p.sR : nullptr;
  const bool vec_ok = ((p.ldc & 7) == 0) && (!Rb || (p.ldr & 7) == 0);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), ch = lane & 7;
    const int m = wm0 + row, n = wn0 + ch * 8;
    uint4 v = *reinterpret_cast<const uint4*>(reg + row * EPI_STRIDE + ch * 16);
    if (m >= p.M || n >= p.N) continue;
    const int rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
    const long long crow = p.gC > 0 ? (long long)(m / p.gC) * p.sgC + (long long)(m % p.gC) * p.ldc : (long long)m * p.ldc;
    if (vec_ok && n + 8 <= p.N) {
      if (Rb) {
        const uint4 rv = *reinterpret_cast<const uint4*>(Rb + (long long)rrow * p.ldr + n);
        const unsigned a[4] = {v.x, v.y, v.z, v.w};
        const unsigned b[4] = {rv.x, rv.y, rv.z, rv.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[k] = pack2(bf2f((bf16_t)(a[k] & 0xffff)) + bf2f((bf16_t)(b[k] & 0xffff)),
                       bf2f((bf16_t)(a[k] >> 16)) + bf2f((bf16_t)(b[k] >> 16)));
        v = uint4{o[0], o[1], o[2], o[3]};
      }
      *reinterpret_cast<uint4*>(Cb + crow + n) = v;
    } else {
      const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (n + k < p.N) {
          float f = bf2f((bf16_t)((k & 1) ? (wv[k >> 1] >> 16) : (wv[k >> 1] & 0xffffu)));
          if (Rb) f += bf2f(Rb[(long long)rrow * p.ldr + n + k]);
          Cb[crow + n + k] = f2bf(f);
        }
      }
    }
  }
}

}  // namespace

extern "C" int vla_gemm_bf16_nt(void* stream, const vla_gemm_desc* d) {
  VLA_REQUIRE(d && d->A && d->B, "gemm: null operand");
  VLA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0, "gemm: empty problem");
  VLA_REQUIRE(d->K % BK == 0, "gemm: K must be a multiple of 64 (pad the operands)");
  VLA_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements (16-B rows)");
  VLA_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "gemm: A/B must be 16-B aligned");
  VLA_REQUIRE(d->sA % 8 == 0 && d->sB % 8 == 0, "gemm: batch strides of A/B must keep 16-B alignment");
  if (d->act == VLA_ACT_SWIGLU) {
    VLA_REQUIRE(d->C2 && d->N % 32 == 0 && d->ldc2 % 4 == 0 && ((uintptr_t)d->C2 & 7) == 0 && d->sC2 % 4 == 0,
                "gemm: swiglu needs C2, N%32==0, ldc2%4==0");
    VLA_REQUIRE(!d->R, "gemm: swiglu epilogue takes no residual");
  } else {
    VLA_REQUIRE(d->C, "gemm: null C");
  }
  if (d->C) VLA_REQUIRE(((uintptr_t)d->C & 15) == 0 || (d->ldc % 8) != 0, "gemm: C must be 16-B aligned");
  if (d->C && d->ldc % 8 == 0) VLA_REQUIRE(d->sC % 8 == 0, "gemm: sC must keep 16-B alignment");
  if (d->R && d->ldc % 8 == 0 && d->ldr % 8 == 0)
    VLA_REQUIRE(((uintptr_t)d->R & 15) == 0 && d->sR % 8 == 0, "gemm: R must be 16-B aligned");
  GemmP p;
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = (bf16_t*)d->C;
  p.bias = (const bf16_t*)d->bias; p.R = (const bf16_t*)d->R; p.C2 = (bf16_t*)d->C2;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.ldc2 = d->ldc2; p.res_mod = d->res_mod; p.act = d->act;
  p.sA = d->sA; p.sB = d->sB; p.sC = d->sC; p.sR = d->sR; p.sC2 = d->sC2; p.sBias = d->sBias;
  const int tm = (d->M + BM - 1) / BM;
  p.tiles_n = (d->N + BN - 1) / BN;
  p.ntiles = tm * p.tiles_n;
  p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  p.gA = d->a_group; p.sgA = d->a_group_stride; p.gC = d->c_group; p.sgC = d->c_group_stride;
  VLA_REQUIRE(d->a_group >= 0 && d->c_group >= 0 && d->a_group_stride % 8 == 0 && (d->c_group == 0 || d->ldc % 8 != 0 || d->c_group_stride % 8 == 0),
              "gemm: row-group strides must keep 16-B alignment");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(p.ntiles, 1, d->batch);
  hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  VLA_CHECK_LAUNCH("gemm_bf16_nt");
  return VLA_OK;
}
