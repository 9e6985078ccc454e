// Micro-benchmark: waves that alternate a block of MFMAs with a block of vector work (the shape of an attention
// tile: 16 x v_mfma_f32_32x32x16_f16, then exp2 / add / convert on the accumulator), several waves per SIMD, no
// LDS, no barriers.  Does the SIMD run one wave's vector block under another wave's MFMA block?
//   hipcc --offload-arch=gfx950 -O3 -o phase_overlap phase_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NMFMA, int NEXP, int NADD, int SYNC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  float e[8], s[8];
  for (int i = 0; i < 8; ++i) { e[i] = threadIdx.x * 0.01f + i; s[i] = 0.f; }
  // de-phase the waves of a SIMD
  for (int w = 0; w < (int)(blockIdx.x % 3) * 40; ++w) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[0]) : "v"(e[1]));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NEXP; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(e[j & 3]) : "v"(e[4 + (j & 3)]));
#pragma unroll
    for (int j = 0; j < NADD; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[j & 7]) : "v"(e[4 + (j & 3)]));
    __builtin_amdgcn_sched_barrier(0);
    if (SYNC) __syncthreads();
  }
  float r = 0.f;
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) r += acc[q][i];
  for (int i = 0; i < 8; ++i) r += e[i] + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NMFMA, int NEXP, int NADD, int SYNC>
void run(const char* name, int blocks) {
  float* out; (void)hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NMFMA, NEXP, NADD, SYNC>), dim3(blocks), dim3(256), 0, 0, out, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NMFMA, NEXP, NADD, SYNC>), dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s waves/SIMD=%d  %.3f ms  -> %.0f ns per tile per wave, %.0f ns of SIMD time per tile\n", name, blocks / 256, ms,
         ms * 1e6 / iters, ms * 1e6 / iters / (blocks / 256));
  (void)hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 768}) {
    run<16, 0, 0, 0>("16 mfma", blocks);
    run<0, 32, 0, 0>("32 exp", blocks);
    run<0, 0, 100, 0>("100 add", blocks);
    run<0, 32, 100, 0>("32 exp + 100 add", blocks);
    run<16, 32, 0, 0>("16 mfma | 32 exp", blocks);
    run<16, 0, 100, 0>("16 mfma | 100 add", blocks);
    run<16, 32, 100, 0>("16 mfma | 32 exp + 100 add", blocks);
    run<16, 32, 100, 1>("16 mfma | 32 exp + 100 add, barrier", blocks);
  }
  return 0;
}
