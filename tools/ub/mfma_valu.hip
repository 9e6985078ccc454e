// Micro-benchmark: does vector work issue under a running MFMA on gfx950?
// Each wave runs N iterations of {1 MFMA 32x32x16 f16, NV v_exp_f32 / v_add_f32 on independent registers}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NEXP, int NADD, int NMFMA, int ACCS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  f32x16 acc[2] = {{0}, {0}};
  float e[8], s[8];
  for (int i = 0; i < 8; ++i) { e[i] = threadIdx.x * 0.01f + i; s[i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (NMFMA) acc[m % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % ACCS], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NEXP; ++j) { asm volatile("v_exp_f32 %0, %1" : "=v"(e[j & 3]) : "v"(e[4 + (j & 3)])); }
#pragma unroll
      for (int j = 0; j < NADD; ++j) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[j & 7]) : "v"(e[4 + (j & 3)])); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[0][i] + acc[1][i];
  for (int i = 0; i < 8; ++i) r += e[i] + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NEXP, int NADD, int NMFMA, int ACCS>
void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 5000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NEXP, NADD, NMFMA, ACCS>), dim3(blocks), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NEXP, NADD, NMFMA, ACCS>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD=%d  %.3f ms  -> %.1f ns per group per wave\n", name, blocks / 256, ms, ms * 1e6 / iters / 8);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512}) {
    run<0, 0, 1, 1>("mfma only, 1 acc", blocks);
    run<0, 0, 1, 2>("mfma only, 2 acc", blocks);
    run<2, 0, 0, 1>("2 exp", blocks);
    run<4, 0, 0, 1>("4 exp", blocks);
    run<0, 6, 0, 1>("6 add", blocks);
    run<2, 0, 1, 2>("mfma + 2 exp", blocks);
    run<2, 2, 1, 2>("mfma + 2 exp + 2 add", blocks);
    run<3, 0, 1, 2>("mfma + 3 exp", blocks);
    run<4, 0, 1, 2>("mfma + 4 exp", blocks);
    run<0, 6, 1, 2>("mfma + 6 add", blocks);
    run<2, 2, 1, 1>("mfma(1 acc) + 2 exp + 2 add", blocks);
  }
  return 0;
}
