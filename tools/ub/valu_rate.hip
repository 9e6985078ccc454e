// Issue-rate micro-benchmark: cycles per wave64 instruction for v_fma_f32, v_pk_fma_f32, v_exp_f32,
// v_rcp_f32 and v_med3_f32 with 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const float c = 0.999f, d = 1e-3f;
  const f2 pc = {c, c}, pd = {d, d};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if constexpr (KIND == 1) {
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
    } else if constexpr (KIND == 2) {
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 3) {
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 4) {
      REP8(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                        "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if constexpr (KIND == 5) {
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
    } else if constexpr (KIND == 7) {   // 4 exp + 4 fma interleaved, independent chains
      REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n"
                        "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if constexpr (KIND == 8) {   // 2 exp + 6 fma
      REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                        "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if constexpr (KIND == 9) {
      REP8(asm volatile("v_exp_f16_e32 %0, %0\n v_exp_f16_e32 %1, %1\n v_exp_f16_e32 %2, %2\n v_exp_f16_e32 %3, %3\n"
                        "v_exp_f16_e32 %4, %4\n v_exp_f16_e32 %5, %5\n v_exp_f16_e32 %6, %6\n v_exp_f16_e32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 10) {
      REP8(asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                        "v_exp_f16_sdwa %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                        "v_exp_f16_sdwa %2, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                        "v_exp_f16_sdwa %3, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                        "v_exp_f16_sdwa %4, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n"
                        "v_exp_f16_sdwa %5, %5 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n"
                        "v_exp_f16_sdwa %6, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n"
                        "v_exp_f16_sdwa %7, %7 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 11) {
      REP8(asm volatile("v_dot2_f32_f16 %0, %8, %9, %0\n v_dot2_f32_f16 %1, %8, %9, %1\n v_dot2_f32_f16 %2, %8, %9, %2\n v_dot2_f32_f16 %3, %8, %9, %3\n"
                        "v_dot2_f32_f16 %4, %8, %9, %4\n v_dot2_f32_f16 %5, %8, %9, %5\n v_dot2_f32_f16 %6, %8, %9, %6\n v_dot2_f32_f16 %7, %8, %9, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if constexpr (KIND == 12) {
      REP8(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else {
      REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                        "v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int KIND> void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 4 * 4); hipMalloc(&cyc, 64);
  for (int threads : {256, 512, 768}) {
    const int iters = 1000;
    k<KIND><<<1, threads>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND><<<1, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // readcyclecounter ticks at a fixed 100 MHz; report ns per instruction per wave on a SIMD instead
    const double n_instr = 64.0 * iters * (threads / 256);   // per SIMD
    printf("%-14s waves/SIMD %d: %.3f ns per wave-instruction (kernel %.1f us, counter %lld)\n", name, threads / 256,
           ms * 1e6 / n_instr, ms * 1e3, c);
  }
}

int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_exp_f32"); run<3>("v_rcp_f32"); run<4>("v_med3_f32");
  run<5>("v_pk_mul/add"); run<6>("v_cvt_pk_f16"); run<7>("4exp+4fma"); run<8>("2exp+6fma");
  run<9>("v_exp_f16"); run<10>("v_exp_f16_sdwa"); run<11>("v_dot2_f32_f16"); run<12>("4max3+4add");
  return 0;
}
