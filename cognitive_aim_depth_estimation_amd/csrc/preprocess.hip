// Image preprocessing in front of the forward pass: 8-bit RGB image -> bilinear resize (Pillow's
// integer resampler, bit for bit) -> / 255 -> (x - mean) / std -> planar fp32 [3, H, W].
// Two passes like the host library: horizontal into an 8-bit intermediate, then vertical; every output
// is (2^21 + sum of byte * 22-bit weight) >> 22 clipped to 0..255.  The last step is a 256-entry table per
// channel, so the float arithmetic is the host's own.  Pure byte/integer streaming: bound by HBM.
#include "common.h"

namespace {

constexpr int PREC = 22;

__device__ __forceinline__ int clip8(int v) {
  v >>= PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[y][xx][c] = resample of src row y at output column xx.  One thread per (xx, y); threads of a wave
// walk neighbouring output columns, whose source windows overlap (reads served from L1/L2).
__global__ __launch_bounds__(256) void resize_rows_kernel(const uint8_t* __restrict__ src, long long src_row_bytes,
                                                          int px, int src_h, const int* __restrict__ bounds,
                                                          const int* __restrict__ coefs, int ksize,
                                                          uint8_t* __restrict__ tmp, int out_w) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (xx >= out_w || y >= src_h) return;
  const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = coefs + (size_t)xx * ksize;
  const uint8_t* p = src + (size_t)y * src_row_bytes + (size_t)x0 * px;      // px bytes per source pixel, R G B first
  int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
    p += px;
  }
  uint8_t* q = tmp + ((size_t)y * out_w + xx) * 3;
  q[0] = (uint8_t)clip8(s0); q[1] = (uint8_t)clip8(s1); q[2] = (uint8_t)clip8(s2);
}

// dst[c][yy][x] = lut[c][resample of tmp column x at output row yy].  One thread per (x, yy): a wave reads
// 192 contiguous bytes of each tmp row it needs and writes 256 contiguous bytes per plane.
__global__ __launch_bounds__(256) void resize_cols_normalize_kernel(const uint8_t* __restrict__ tmp, int out_w,
                                                                    const int* __restrict__ bounds,
                                                                    const int* __restrict__ coefs, int ksize,
                                                                    const float* __restrict__ lut,
                                                                    float* __restrict__ dst, int out_h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int yy = blockIdx.y;
  if (x >= out_w || yy >= out_h) return;
  const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = coefs + (size_t)yy * ksize;
  const uint8_t* p = tmp + ((size_t)y0 * out_w + x) * 3;
  const size_t row = (size_t)out_w * 3;
  int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
    p += row;
  }
  const size_t plane = (size_t)out_h * out_w, o = (size_t)yy * out_w + x;
  dst[o] = lut[clip8(s0)];
  dst[plane + o] = lut[256 + clip8(s1)];
  dst[2 * plane + o] = lut[512 + clip8(s2)];
}

// One image of a batch: what cogaim_preprocess_rgb8 takes as arguments, as a record in device memory.
struct PreprocessItem {
  const uint8_t* src;      // [src_h, src_w, px] bytes, src_row_bytes between rows
  long long src_row_bytes;
  const int* xbounds; const int* xcoefs;
  const int* ybounds; const int* ycoefs;
  uint8_t* tmp;            // [src_h, out_w, 3] scratch of this image
  float* dst;              // [3, out_h, out_w]
  int src_h, src_w, px, xk, yk, pad;
};
static_assert(sizeof(PreprocessItem) == 88, "host code packs this record by hand");

// The two passes above for a whole batch in one launch each: blockIdx.z picks the image.
__global__ __launch_bounds__(256) void resize_rows_batch_kernel(const PreprocessItem* __restrict__ items, int out_w) {
  const PreprocessItem it = items[blockIdx.z];
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (xx >= out_w || y >= it.src_h) return;
  const int x0 = it.xbounds[2 * xx], n = it.xbounds[2 * xx + 1];
  const int* k = it.xcoefs + (size_t)xx * it.xk;
  const uint8_t* p = it.src + (size_t)y * it.src_row_bytes + (size_t)x0 * it.px;
  int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
    p += it.px;
  }
  uint8_t* q = it.tmp + ((size_t)y * out_w + xx) * 3;
  q[0] = (uint8_t)clip8(s0); q[1] = (uint8_t)clip8(s1); q[2] = (uint8_t)clip8(s2);
}

__global__ __launch_bounds__(256) void resize_cols_normalize_batch_kernel(const PreprocessItem* __restrict__ items, int out_w,
                                                                          const float* __restrict__ lut, int out_h) {
  const PreprocessItem it = items[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int yy = blockIdx.y;
  if (x >= out_w || yy >= out_h) return;
  const int y0 = it.ybounds[2 * yy], n = it.ybounds[2 * yy + 1];
  const int* k = it.ycoefs + (size_t)yy * it.yk;
  const uint8_t* p = it.tmp + ((size_t)y0 * out_w + x) * 3;
  const size_t row = (size_t)out_w * 3;
  int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
    p += row;
  }
  const size_t plane = (size_t)out_h * out_w, o = (size_t)yy * out_w + x;
  it.dst[o] = lut[clip8(s0)];
  it.dst[plane + o] = lut[256 + clip8(s1)];
  it.dst[2 * plane + o] = lut[512 + clip8(s2)];
}

}  // namespace

extern "C" {

// src: uint8 [src_h, src_w, src_pixel_bytes] (3 = packed RGB, 4 = RGBX as the host imaging library keeps it in
// memory; the fourth byte is ignored) with src_row_bytes between rows; xbounds/ybounds int32 [out, 2] = (first
// source index, taps), xcoefs/ycoefs int32 [out, xk|yk] 22-bit weights; tmp: uint8 [src_h, out_w, 3] scratch;
// lut: fp32 [3, 256]; dst: fp32 [3, out_h, out_w].
int cogaim_preprocess_rgb8(const void* src, int src_h, int src_w, long long src_row_bytes, int src_pixel_bytes,
                           const int* xbounds, const int* xcoefs, int xk, const int* ybounds, const int* ycoefs, int yk, void* tmp,
                           const float* lut, float* dst, int out_h, int out_w, void* stream) {
  if (src_h <= 0 || src_w <= 0 || out_h <= 0 || out_w <= 0 || xk <= 0 || yk <= 0 ||
      (src_pixel_bytes != 3 && src_pixel_bytes != 4) || src_row_bytes < (long long)src_pixel_bytes * src_w ||
      src_h > 65535 || out_h > 65535)
    return COGAIM_EINVAL;
  const dim3 block(256);
  hipLaunchKernelGGL(resize_rows_kernel, dim3((out_w + 255) / 256, src_h), block, 0, (hipStream_t)stream,
                     (const uint8_t*)src, src_row_bytes, src_pixel_bytes, src_h, xbounds, xcoefs, xk, (uint8_t*)tmp, out_w);
  hipLaunchKernelGGL(resize_cols_normalize_kernel, dim3((out_w + 255) / 256, out_h), block, 0, (hipStream_t)stream,
                     (const uint8_t*)tmp, out_w, ybounds, ycoefs, yk, lut, dst, out_h);
  return COGAIM_LAUNCH_CHECK();
}

// The same for n images in two launches.  items: n records of 88 bytes in DEVICE memory, each
//   { u64 src, i64 src_row_bytes, u64 xbounds, u64 xcoefs, u64 ybounds, u64 ycoefs, u64 tmp, u64 dst,
//     i32 src_h, i32 src_w, i32 src_pixel_bytes, i32 xk, i32 yk, i32 0 }
// with the meaning of the arguments above (validated by the caller: records are read on the device);
// max_src_h = largest src_h among them.
int cogaim_preprocess_rgb8_batch(const void* items, int n, int max_src_h, const float* lut, int out_h, int out_w,
                                 void* stream) {
  if (!items || n <= 0 || n > 65535 || max_src_h <= 0 || max_src_h > 65535 || out_h <= 0 || out_h > 65535 || out_w <= 0)
    return COGAIM_EINVAL;
  const dim3 block(256);
  hipLaunchKernelGGL(resize_rows_batch_kernel, dim3((out_w + 255) / 256, max_src_h, n), block, 0, (hipStream_t)stream,
                     (const PreprocessItem*)items, out_w);
  hipLaunchKernelGGL(resize_cols_normalize_batch_kernel, dim3((out_w + 255) / 256, out_h, n), block, 0, (hipStream_t)stream,
                     (const PreprocessItem*)items, out_w, lut, out_h);
  return COGAIM_LAUNCH_CHECK();
}

}  // extern "C"
