// Row-wise HBM-bound kernels of the DINOv2 path: LayerNorm (fp32 residual
// stream -> 16-bit GEMM operand, or -> fp32 for the final norm), the im2col
// gather that feeds the patch-embedding GEMM, and the CLS-token row.
// One 64-lane wave per row, 16-byte loads, wave-shuffle reductions.
#include "common.h"

namespace {

constexpr int LN_MAX_VEC = 8;   // float4 per lane -> D <= 2048

template <class OutT, bool F32OUT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out,
                                                        int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  const int nv = D >> 2;
  float4 v[LN_MAX_VEC];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      v[i] = xr[c];
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c];
      const float4 b = reinterpret_cast<const float4*>(beta)[c];
      const float y0 = (v[i].x - mean) * rstd * g.x + b.x, y1 = (v[i].y - mean) * rstd * g.y + b.y;
      const float y2 = (v[i].z - mean) * rstd * g.z + b.z, y3 = (v[i].w - mean) * rstd * g.w + b.w;
      if constexpr (F32OUT)
        reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * D)[c] = make_float4(y0, y1, y2, y3);
      else
        reinterpret_cast<Half4<OutT>*>(reinterpret_cast<OutT*>(out) + (size_t)row * D)[c] =
            to_half4<OutT>(y0, y1, y2, y3);
    }
  }
}

// Sum over the 64 lanes, result in every lane, without the LDS permute path: rotations inside rows of 16 lanes
// (DPP), then the two row-swap instructions of gfx950.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define COGAIM_ROR(ctrl_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, 0xf, 0xf, false))
  v += COGAIM_ROR(0x128);   // row_ror:8
  v += COGAIM_ROR(0x124);   // row_ror:4
  v += COGAIM_ROR(0x122);   // row_ror:2
  v += COGAIM_ROR(0x121);   // row_ror:1
#undef COGAIM_ROR
  unsigned u = __float_as_uint(v);
  auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // rows (0,1) and (2,3) meet
  v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
  u = __float_as_uint(v);
  auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // the two halves meet
  return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
}

// LayerNorm for D = 256 * NV (768: NV = 3, 1024: NV = 4): every lane holds exactly NV float4 of its row, so there is
// no per-chunk predicate; gamma / beta are requested before the reductions so that their latency lies under them.
template <class OutT, bool F32OUT, int NV>
__global__ __launch_bounds__(256) void layernorm_fixed_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, void* __restrict__ out,
                                                              int M, float eps) {
  constexpr int D = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[NV], g[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[lane + 64 * i];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
    b[i] = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum_dpp(sum) * (1.0f / (float)D);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(wave_sum_dpp(sq) * (1.0f / (float)D) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    const float y0 = v[i].x * rstd * g[i].x + b[i].x, y1 = v[i].y * rstd * g[i].y + b[i].y;
    const float y2 = v[i].z * rstd * g[i].z + b[i].z, y3 = v[i].w * rstd * g[i].w + b[i].w;
    if constexpr (F32OUT)
      reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * D)[c] = make_float4(y0, y1, y2, y3);
    else
      reinterpret_cast<Half4<OutT>*>(reinterpret_cast<OutT*>(out) + (size_t)row * D)[c] = to_half4<OutT>(y0, y1, y2, y3);
  }
}

// images [B,3,H,W] fp32 -> cols [B*gh*gw, Kp] 16-bit, column c*p*p + ky*p + kx, zero padded to Kp.
template <class T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int B,
                                                     int Himg, int Wimg, int gh, int gw, int p, int Kp) {
  const int chunks = Kp >> 3;
  const long long total = (long long)B * gh * gw * chunks;
  const int K = 3 * p * p;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    const long long row = i / chunks;
    const int px = (int)(row % gw);
    const int py = (int)((row / gw) % gh);
    const int b = (int)(row / ((long long)gw * gh));
    alignas(16) T o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ch * 8 + j;
      float val = 0.f;
      if (k < K) {
        const int c = k / (p * p), rem = k - c * p * p, ky = rem / p, kx = rem - ky * p;
        val = img[(((size_t)b * 3 + c) * Himg + (py * p + ky)) * Wimg + px * p + kx];
      }
      o[j] = (T)val;
    }
    *reinterpret_cast<uint4*>(cols + row * Kp + ch * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// The same gather one strip of patches at a time: a workgroup owns `seg` adjacent patches of one patch row of one
// image, reads the 3 * p image-row segments they share as whole contiguous runs (converted to 16 bits in LDS), then
// writes the patches' rows of `cols` as 16-byte pieces -- every image byte is fetched once, where the per-chunk gather
// above touches each 128-byte line from several waves (2.2 x the image bytes in HBM reads at 518 pixels).
// Work is dealt in units of (image row, 64 pixels), 16 units' loads in flight per wave; the position of column k inside
// the LDS tile comes from a table built once per workgroup, so neither loop divides.
constexpr int IM2COL_LDS_ELEMS = 22528;   // 44 KB of 16-bit pixels: 37 patches of 14 x 14 x 3
constexpr int IM2COL_MAX_K = 1024;        // 3 p^2 handled by the strip kernel (p <= 18)
constexpr int IM2COL_MAX_ROWS = 64;       // 3 p
constexpr int IM2COL_THREADS = 512;       // 8 waves a workgroup: enough loads in flight per CU to cover HBM latency
constexpr int IM2COL_WAVES = IM2COL_THREADS / 64;

template <class T>
__global__ __launch_bounds__(IM2COL_THREADS) void im2col_strip_kernel(const float* __restrict__ img, T* __restrict__ cols,
                                                           int Himg, int Wimg, int gh, int gw, int p, int Kp, int seg) {
  __shared__ __attribute__((aligned(16))) T tile[IM2COL_LDS_ELEMS];
  __shared__ __attribute__((aligned(16))) unsigned short koff[IM2COL_MAX_K + 8];
  __shared__ int rowoff[IM2COL_MAX_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px0 = blockIdx.x * seg, py = blockIdx.y, b = blockIdx.z;
  const int np = min(seg, gw - px0);          // patches in this strip
  const int wcols = np * p;                   // pixels per image-row segment
  const int rows = 3 * p, K = rows * p, pp = p * p;
  if (tid < rows) {
    const int c = tid / p, ky = tid - c * p;
    rowoff[tid] = (c * Himg + ky) * Wimg;
  }
  for (int k = tid; k < IM2COL_MAX_K + 8; k += IM2COL_THREADS) {
    int v = 0;
    if (k < K) {
      const int c = k / pp, rem = k - c * pp, ky = rem / p, kx = rem - ky * p;
      v = (c * p + ky) * wcols + kx;
    }
    koff[k] = (unsigned short)v;
  }
  __syncthreads();
  const float* src = img + ((size_t)b * 3 * Himg + (size_t)py * p) * Wimg + (size_t)px0 * p;
  {
    const int nseg = (wcols + 63) >> 6, units = rows * nseg;
    int r = 0, sg = wave;
    while (sg >= nseg) { sg -= nseg; ++r; }
    for (int u = wave; u < units; u += 16 * IM2COL_WAVES) {
      float v[16];
      int at[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int xx = sg * 64 + lane;
        const bool ok = (u + IM2COL_WAVES * j < units) && xx < wcols;
        at[j] = ok ? r * wcols + xx : -1;
        v[j] = ok ? src[rowoff[ok ? r : 0] + xx] : 0.f;
        sg += IM2COL_WAVES;
        while (sg >= nseg) { sg -= nseg; ++r; }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (at[j] >= 0) tile[at[j]] = (T)v[j];
    }
  }
  __syncthreads();
  const int chunks = Kp >> 3;
  T* dst = cols + (((size_t)b * gh + py) * gw + px0) * Kp;
  const int dpl = IM2COL_THREADS / chunks, dch = IM2COL_THREADS - dpl * chunks;
  int pl = tid / chunks, ch = tid - pl * chunks;
  for (; pl < np; pl += dpl) {
    alignas(16) T o[8];
    const int k = ch * 8;
    if (k < IM2COL_MAX_K) {
      const uint4 q = *reinterpret_cast<const uint4*>(koff + k);
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int at = (int)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu) + pl * p;
        o[j] = (k + j < K) ? tile[at] : (T)0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)0.f;
    }
    *reinterpret_cast<uint4*>(dst + (size_t)pl * Kp + k) = *reinterpret_cast<const uint4*>(o);
    ch += dch;
    if (ch >= chunks) { ch -= chunks; ++pl; }
  }
}

__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                               int B, int T, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  x[(size_t)b * T * D + d] = cls[d] + pos[d];
}

}  // namespace

extern "C" {

// out[M,D] = LayerNorm(x[M,D]) * gamma + beta.  out_dtype: 0 bf16, 1 fp16, 2 fp32.
int cogaim_layernorm(const float* x, const float* gamma, const float* beta, void* out, int M, int D, float eps,
                     int out_dtype, void* stream) {
  if (M <= 0 || (D & 3) || D > 256 * LN_MAX_VEC) return COGAIM_EINVAL;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype < 0 || out_dtype > 2) return COGAIM_EINVAL;
#define COGAIM_LN_FIXED(NV_)                                                                                                   \
  {                                                                                                                           \
    if (out_dtype == 0) hipLaunchKernelGGL((layernorm_fixed_kernel<__bf16, false, NV_>), grid, block, 0, st, x, gamma, beta, out, M, eps); \
    else if (out_dtype == 1) hipLaunchKernelGGL((layernorm_fixed_kernel<_Float16, false, NV_>), grid, block, 0, st, x, gamma, beta, out, M, eps); \
    else hipLaunchKernelGGL((layernorm_fixed_kernel<float, true, NV_>), grid, block, 0, st, x, gamma, beta, out, M, eps);      \
    return COGAIM_LAUNCH_CHECK();                                                                                              \
  }
  if (D == 768) COGAIM_LN_FIXED(3)
  if (D == 1024) COGAIM_LN_FIXED(4)
#undef COGAIM_LN_FIXED
  if (out_dtype == 0) hipLaunchKernelGGL((layernorm_kernel<__bf16, false>), grid, block, 0, st, x, gamma, beta, out, M, D, eps);
  else if (out_dtype == 1) hipLaunchKernelGGL((layernorm_kernel<_Float16, false>), grid, block, 0, st, x, gamma, beta, out, M, D, eps);
  else if (out_dtype == 2) hipLaunchKernelGGL((layernorm_kernel<float, true>), grid, block, 0, st, x, gamma, beta, out, M, D, eps);
  else return COGAIM_EINVAL;
  return COGAIM_LAUNCH_CHECK();
}

int cogaim_im2col(const float* images, void* cols, int B, int H, int W, int patch, int Kp, int dtype, void* stream) {
  const int gh = H / patch, gw = W / patch;
  if (B <= 0 || gh <= 0 || gw <= 0 || (Kp & 7) || Kp < 3 * patch * patch) return COGAIM_EINVAL;
  if (dtype != 0 && dtype != 1) return COGAIM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int seg = IM2COL_LDS_ELEMS / (3 * patch * patch);       // patches per strip that fit the LDS tile
  if (seg >= 1 && 3 * patch * patch <= IM2COL_MAX_K && gh <= 65535 && B <= 65535 && 3LL * H * W < (1LL << 31)) {
    const int s = seg < gw ? seg : gw;
    dim3 grid((gw + s - 1) / s, gh, B);
    if (dtype == 0) hipLaunchKernelGGL(im2col_strip_kernel<__bf16>, grid, dim3(IM2COL_THREADS), 0, st, images, (__bf16*)cols, H, W, gh, gw, patch, Kp, s);
    else hipLaunchKernelGGL(im2col_strip_kernel<_Float16>, grid, dim3(IM2COL_THREADS), 0, st, images, (_Float16*)cols, H, W, gh, gw, patch, Kp, s);
    return COGAIM_LAUNCH_CHECK();
  }
  const long long total = (long long)B * gh * gw * (Kp >> 3);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (dtype == 0) hipLaunchKernelGGL(im2col_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, images, (__bf16*)cols, B, H, W, gh, gw, patch, Kp);
  else if (dtype == 1) hipLaunchKernelGGL(im2col_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, images, (_Float16*)cols, B, H, W, gh, gw, patch, Kp);
  else return COGAIM_EINVAL;
  return COGAIM_LAUNCH_CHECK();
}

// x[b, 0, :] = cls + pos[0]   (x is [B, T, D] fp32)
int cogaim_cls_row(const float* cls, const float* pos, float* x, int B, int T, int D, void* stream) {
  if (B <= 0 || T <= 0 || D <= 0) return COGAIM_EINVAL;
  hipLaunchKernelGGL(cls_row_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos, x, B, T, D);
  return COGAIM_LAUNCH_CHECK();
}

}  // extern "C"
