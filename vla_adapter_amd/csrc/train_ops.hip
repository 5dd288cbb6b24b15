// Glue kernels of the backbone-training steps (LoRA and full fine-tune through DINOv2 + SigLIP, two images): LayerScale as a
// parameter, strided 3-level row copies (feature-buffer <-> per-backbone layouts, prefix-token placement).  HBM-bound, 16 B per
// lane wherever the addresses allow it.
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

// dst[g][r][0:cols] = src[g][r][0:cols]  (bf16; 16-B lanes when VEC)
template <bool VEC>
__global__ void copy_rows3d_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int groups, int rows, int cols, long long s_sg,
                                   long long s_sr, long long d_sg, long long d_sr) {
  const int per = VEC ? cols >> 3 : cols;
  const long long total = (long long)groups * rows * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long gr = i / per;
    const int c = (int)(i - gr * per);
    const int g = (int)(gr / rows), r = (int)(gr - (long long)g * rows);
    const bf16_t* s = src + g * s_sg + r * s_sr;
    bf16_t* d = dst + g * d_sg + r * d_sr;
    if (VEC) reinterpret_cast<uint4*>(d)[c] = reinterpret_cast<const uint4*>(s)[c];
    else d[c] = s[c];
  }
}

// out = bf16(x + bf16(a * ls)): LayerScale (modeling_prismatic.py:58-66, `x * self.scale_factor`) + the block's residual add
__global__ void layerscale_fwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ ls, const bf16_t* __restrict__ x,
                                      bf16_t* __restrict__ out, long long rows, int cols8) {
  const long long total = rows * cols8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols8);
    float fa[8], fl[8], fx[8], o[8];
    unpack8(reinterpret_cast<const uint4*>(a)[i], fa);
    unpack8(reinterpret_cast<const uint4*>(ls)[c], fl);
    unpack8(reinterpret_cast<const uint4*>(x)[i], fx);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = fx[k] + rbf(fa[k] * fl[k]);
    reinterpret_cast<uint4*>(out)[i] = pack8(o);
  }
}

// da = bf16(dy * ls) ; dls[c] += sum_r dy[r, c] * a[r, c]   (each lane owns 8 columns; the 4 waves of a block split the rows of
// its chunk, partial sums meet in LDS, one f32 atomic per column per block - the colsum_vec structure)
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ a, const bf16_t* __restrict__ ls,
                                                             bf16_t* __restrict__ da, float* __restrict__ dls, int rows, int cols, int rows_per) {
  __shared__ float sm[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < cols) {
    float fl[8];
    unpack8(*reinterpret_cast<const uint4*>(ls + c), fl);
    for (int r = r0 + w; r < r1; r += 4) {
      float fd[8], fa[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + (long long)r * cols + c), fd);
      if (dls) unpack8(*reinterpret_cast<const uint4*>(a + (long long)r * cols + c), fa);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o[k] = fd[k] * fl[k];
        if (dls) acc[k] += fd[k] * fa[k];
      }
      *reinterpret_cast<uint4*>(da + (long long)r * cols + c) = pack8(o);
    }
  }
  if (!dls) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[w][lane * 8 + k] = acc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int cc = blockIdx.x * 512 + i;
    if (cc < cols) atomicAdd(dls + cc, sm[0][i] + sm[1][i] + sm[2][i] + sm[3][i]);
  }
}

inline dim3 grid1d(long long n, int per) {
  long long b = (n + per - 1) / per;
  return dim3((unsigned)(b < 1 ? 1 : (b > 65535 * 8 ? 65535 * 8 : b)));
}

}  // namespace

extern "C" int vla_copy_rows3d(void* stream, const void* src, void* dst, int groups, int rows, int cols, long long src_group_stride,
                               long long src_row_stride, long long dst_group_stride, long long dst_row_stride) {
  VLA_REQUIRE(src && dst && groups > 0 && rows > 0 && cols > 0, "copy_rows3d: bad args");
  const bool vec = cols % 8 == 0 && src_group_stride % 8 == 0 && src_row_stride % 8 == 0 && dst_group_stride % 8 == 0 && dst_row_stride % 8 == 0 &&
                   ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)groups * rows * (vec ? cols / 8 : cols);
  if (vec)
    hipLaunchKernelGGL(copy_rows3d_kernel<true>, grid1d(total, 256), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, groups, rows, cols,
                       src_group_stride, src_row_stride, dst_group_stride, dst_row_stride);
  else
    hipLaunchKernelGGL(copy_rows3d_kernel<false>, grid1d(total, 256), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, groups, rows, cols,
                       src_group_stride, src_row_stride, dst_group_stride, dst_row_stride);
  VLA_CHECK_LAUNCH("copy_rows3d");
  return VLA_OK;
}

extern "C" int vla_layerscale_fwd(void* stream, const void* a, const void* ls, const void* x, void* out, long long rows, int cols) {
  VLA_REQUIRE(a && ls && x && out && rows > 0 && cols > 0 && cols % 8 == 0, "layerscale_fwd: cols must be a multiple of 8");
  VLA_REQUIRE(((((uintptr_t)a) | ((uintptr_t)ls) | ((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0, "layerscale_fwd: 16-B aligned contiguous tensors");
  hipLaunchKernelGGL(layerscale_fwd_kernel, grid1d(rows * (cols / 8), 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)ls,
                     (const bf16_t*)x, (bf16_t*)out, rows, cols / 8);
  VLA_CHECK_LAUNCH("layerscale_fwd");
  return VLA_OK;
}

extern "C" int vla_layerscale_bwd(void* stream, const void* dy, const void* a, const void* ls, void* da, float* dls, int rows, int cols) {
  VLA_REQUIRE(dy && ls && da && rows > 0 && cols > 0 && cols % 8 == 0 && (dls == nullptr || a != nullptr), "layerscale_bwd: bad args (cols % 8 == 0)");
  VLA_REQUIRE(((((uintptr_t)dy) | ((uintptr_t)ls) | ((uintptr_t)da) | ((uintptr_t)a)) & 15) == 0, "layerscale_bwd: 16-B aligned contiguous tensors");
  const int gx = (cols + 511) / 512;
  int gy = (rows + 63) / 64;
  if (gy > 1024) gy = 1024;
  const int rows_per = (rows + gy - 1) / gy;
  gy = (rows + rows_per - 1) / rows_per;
  hipLaunchKernelGGL(layerscale_bwd_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)a, (const bf16_t*)ls,
                     (bf16_t*)da, dls, rows, cols, rows_per);
  VLA_CHECK_LAUNCH("layerscale_bwd");
  return VLA_OK;
}
