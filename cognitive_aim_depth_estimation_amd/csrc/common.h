// Shared device helpers for the gfx950 (MI355X) kernels of the Cognitive-Aim
// depth forward pass.  Wave size is 64 everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define COGAIM_OK 0
#define COGAIM_EINVAL (-22)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-bit operand traits: bf16 (dtype 0) and fp16 (dtype 1) share every kernel.
template <class T> struct Half;
template <> struct Half<__bf16> {
  using v8 = bf16x8;
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Half<_Float16> {
  using v8 = f16x8;
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <class T> struct alignas(8) Half4 { T v[4]; };

template <class T>
__device__ __forceinline__ Half4<T> to_half4(float a, float b, float c, float d) {
  Half4<T> r;
  r.v[0] = (T)a; r.v[1] = (T)b; r.v[2] = (T)c; r.v[3] = (T)d;
  return r;
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
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a 256-thread block; result valid in every thread.  `red` is 4+ floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// 16-byte-per-lane DMA from global memory straight into LDS (no VGPR round trip): lane i's 16 bytes
// land at lds_wave_base + 16 i, the source address is free per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same DMA through a buffer resource (buffer_load_dwordx4 ... lds): byte offsets instead of
// pointers, reads past num_bytes return zero instead of faulting, and -- unlike the global_ form, which
// the compiler treats as a flat access and answers with lgkmcnt(0) before every later LDS read -- it
// leaves the compiler's LDS wait counting exact.
// (base and extent must be wave-uniform; they are forced into scalar registers here, otherwise the compiler
// wraps every use of the resource in a per-lane "waterfall" loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned num_bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(num_bytes), 0x00020000);
}
__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t rsrc, int lane_byte_offset, int wave_byte_offset,
                                              void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           lane_byte_offset, wave_byte_offset, 0, 0);
}

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? COGAIM_OK : -1000 - (int)e; }
#define COGAIM_LAUNCH_CHECK() hip_rc(hipGetLastError())
