// Attention output projection of a DINOv2 block together with the LayerNorm that follows it, for a model width
// of 768:
//     x[m, :] += (A[m, :] W^T + bias) * ls                    (fp32 residual stream, in place)
//     h[m, :]  = LayerNorm(x[m, :]) * gamma + beta            (16-bit operand of the next GEMM)
// LayerNorm needs whole rows, so a workgroup owns 96 complete rows: 8 waves as 2 (rows) x 4 (columns), each wave
// 48 rows x 192 columns = 3 x 12 v_mfma_f32_16x16x32 tiles (144 accumulator registers).  The price of whole rows is
// that every K step brings ALL 768 rows of W into LDS; with K steps of 32 that is 48 KiB of W and 6 KiB of A per
// step, double buffered (108 KiB), staged by 16-byte global_load_lds.
//
// LDS image of an operand: [row][4 x 16-byte chunk]; the chunk index is XORed with g[(row >> 2) & 3],
// g = {0, 3, 2, 1}, on the DMA source address and again on the ds_read_b128 fragment reads.  A ds_read_b128 is
// served in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32; with this
// permutation the 16 lanes of every group fall on 16 different 16-byte columns of the 256-byte bank row.
//
// After the last K step the accumulators take the residual update, the new x is written, and the row statistics are
// formed over the lane's 48 values, the 4 lanes of a row (shuffles) and the 4 column waves (LDS): mean first, then
// the centred sum of squares, the same two-pass form the stand-alone LayerNorm uses.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int RL_N = 768, RL_BM = 96, RL_BK = 32;
constexpr int RL_A_BYTES = RL_BM * RL_BK * 2;          // 6 KiB
constexpr int RL_W_BYTES = RL_N * RL_BK * 2;           // 48 KiB
constexpr int RL_STAGE = RL_A_BYTES + RL_W_BYTES;      // 54 KiB

__device__ __forceinline__ int rl_swz(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

#define RL_BARRIER()                               \
  do {                                             \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    __builtin_amdgcn_sched_barrier(0);             \
  } while (0)

// acc[i][jj][e] is row wm*48 + 16 i + fr, column wn*192 + 16 jj + 4 fq + e of the workgroup's 96 x 768 tile.
// colv: [4][768] bias, LayerScale, gamma, beta in LDS; red: [2][4][96] floats of LDS.  Rows past M (FULL == false
// only) are clamped to row M - 1 for loads and skip their stores.
// One group of 16 rows is taken all the way to its h rows before the next one starts: the updated values then live
// in 48 registers next to the 144 accumulators.  (With the whole tile updated first, the register allocator does not
// reuse the accumulator registers as they die and spills 100+ registers.)
template <bool FULL, class T>
__device__ __forceinline__ void rowln_epilogue(const f32x4 (&acc)[3][12], const float* colv, float* red,
                                               float* __restrict__ x, T* __restrict__ h, int M, int ldx, int ldh, float eps,
                                               int row0, int wm, int wn, int fr, int fq) {
  const int ncol0 = wn * 192 + 4 * fq;
  const f32x4* const cv = reinterpret_cast<const f32x4*>(colv);      // [4][192] float4
  float* const red2 = red + 4 * RL_BM;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int rl = wm * 48 + 16 * i + fr;
    const int m = row0 + rl;
    const bool ok = FULL || m < M;
    float* xrow = x + (size_t)(ok ? m : M - 1) * ldx + ncol0;
    T* hrow = h + (size_t)(ok ? m : M - 1) * ldh + ncol0;
    // the column constants are re-read from LDS for every row group: the index is made opaque so that the
    // compiler does not keep all of them in registers across the row groups
    int c4 = ncol0 >> 2;
    asm volatile("" : "+v"(c4));
    f32x4 nv[12];
    float s = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 r[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) r[j] = *reinterpret_cast<const f32x4*>(xrow + 16 * (6 * half + j));
      __builtin_amdgcn_sched_barrier(0);       // six loads of x in flight, no more
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if ((j & 1) == 0) __builtin_amdgcn_sched_barrier(0);     // and the constants of two fragments at a time
        const int jj = 6 * half + j;
        const f32x4 b = cv[c4 + 4 * jj], l = cv[192 + c4 + 4 * jj];
        const f32x4 a = __builtin_elementwise_fma(acc[i][jj] + b, l, r[j]);
        nv[jj] = a;
        if (ok) *reinterpret_cast<f32x4*>(xrow + 16 * jj) = a;
        s += (a[0] + a[1]) + (a[2] + a[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (fq == 0) red[wn * RL_BM + rl] = s;
    RL_BARRIER();
    const float mean = ((red[rl] + red[RL_BM + rl]) + (red[2 * RL_BM + rl] + red[3 * RL_BM + rl])) * (1.0f / (float)RL_N);
    float q = 0.f;
#pragma unroll
    for (int jj = 0; jj < 12; ++jj) {
      const f32x4 a = nv[jj] - mean;
      nv[jj] = a;
      q += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (fq == 0) red2[wn * RL_BM + rl] = q;
    RL_BARRIER();
    const float var = ((red2[rl] + red2[RL_BM + rl]) + (red2[2 * RL_BM + rl] + red2[3 * RL_BM + rl])) * (1.0f / (float)RL_N);
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int jj = 0; jj < 12; ++jj) {
      if ((jj & 1) == 0) __builtin_amdgcn_sched_barrier(0);
      const f32x4 g = cv[384 + c4 + 4 * jj], b = cv[576 + c4 + 4 * jj];
      const f32x4 y = __builtin_elementwise_fma(nv[jj] * rstd, g, b);
      const Half4<T> o = to_half4<T>(y[0], y[1], y[2], y[3]);
      if (ok) *reinterpret_cast<Half4<T>*>(hrow + 16 * jj) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class T>
__global__ __launch_bounds__(512, 2) void gemm_rowln_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ ls,
                                                            float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ h, int M, int K,
                                                            int lda, int ldw, int ldx, int ldh, float eps, int dbg) {
  using H = Half<T>;
  using v8 = typename H::v8;
  __shared__ __attribute__((aligned(1024))) char smem[2 * RL_STAGE];
  __shared__ __attribute__((aligned(16))) float colv[4][RL_N];        // bias, ls, gamma, beta
  __shared__ float red[2][4][RL_BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
  const int row0 = blockIdx.x * RL_BM;

  for (int i = tid; i < RL_N; i += 512) {
    colv[0][i] = bias[i]; colv[1][i] = ls[i]; colv[2][i] = gamma[i]; colv[3][i] = beta[i];
  }

  // ---- DMA plan: piece p of a stage covers its bytes [1024 p, 1024 p + 1024) = 16 operand rows.  Wave w brings the
  // W pieces w, w + 8, ..., w + 40 (rows 128 apart: one per-lane pointer and uniform steps) and, for w < 6, A piece w.
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const T* srcW;
  const T* srcA;
  {
    const int cpos = lane & 3;
    const int n = uwave * 16 + (lane >> 2);
    srcW = W + (size_t)n * ldw + ((cpos ^ rl_swz(n)) << 3);
    const int r = (uwave < 6 ? uwave : 0) * 16 + (lane >> 2);
    srcA = A + (size_t)min(row0 + r, M - 1) * lda + ((cpos ^ rl_swz(r)) << 3);
  }
  const size_t wstep = (size_t)128 * ldw;
#define RL_ISSUE(stage_)                                                                        \
  do {                                                                                          \
    char* const sb_ = smem + (stage_) * RL_STAGE;                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_)                                            \
      glds16(srcW + j_ * wstep, sb_ + RL_A_BYTES + (uwave + 8 * j_) * 1024);                    \
    if (uwave < 6) glds16(srcA, sb_ + uwave * 1024);                                            \
    srcW += RL_BK; srcA += RL_BK;                                                               \
  } while (0)

  // fragment addresses: rows wm*48 + 16 i + fr (A) and wn*192 + 16 jj + fr (W); (row >> 2) & 3 == (fr >> 2) & 3
  const int c0 = (fq ^ rl_swz(fr)) << 4;
  const char* const a_base = smem + (wm * 48 + fr) * 64 + c0;                          // + i * 1024
  const char* const w_base = smem + RL_A_BYTES + (wn * 192 + fr) * 64 + c0;            // + jj * 1024

  f32x4 acc[3][12];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jj = 0; jj < 12; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = dbg == 2 ? 1 : K / RL_BK;
  RL_ISSUE(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step kt have landed
    RL_BARRIER();                                          // everyone's have; everyone is done reading step kt - 1
    if (kt + 1 < nk) { RL_ISSUE(cur ^ 1); }
    const int boff = cur * RL_STAGE;
    v8 af[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const v8*>(a_base + boff + i * 1024);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      v8 wf[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const v8*>(w_base + boff + (half * 6 + j) * 1024);
      __builtin_amdgcn_s_setprio(1);
      if (dbg != 1)
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i][half * 6 + j] = H::mfma16(wf[j], af[i], acc[i][half * 6 + j]);
      __builtin_amdgcn_s_setprio(0);
    }
  }
#undef RL_ISSUE

  // ---- epilogue (the last workgroup, whose rows may run past M, takes the predicated copy) ---------------------
  if (row0 + RL_BM <= M)
    rowln_epilogue<true, T>(acc, &colv[0][0], &red[0][0][0], x, h, M, ldx, ldh, eps, row0, wm, wn, fr, fq);
  else
    rowln_epilogue<false, T>(acc, &colv[0][0], &red[0][0][0], x, h, M, ldx, ldh, eps, row0, wm, wn, fr, fq);
#undef RL_BARRIER
}

}  // namespace

extern "C" int cogaim_gemm_residual_layernorm(const void* A, const void* W, const float* bias, const float* ls, float* x,
                                              const float* gamma, const float* beta, void* h, int M, int N, int K, int lda,
                                              int ldw, int ldx, int ldh, float eps, int dtype, void* stream) {
  if (!A || !W || !bias || !ls || !x || !gamma || !beta || !h) return COGAIM_EINVAL;
  if (M <= 0 || N != RL_N || K <= 0 || K % RL_BK) return COGAIM_EINVAL;
  if (lda < K || ldw < K || (lda & 7) || (ldw & 7) || ldx < N || ldh < N || (ldx & 3) || (ldh & 3)) return COGAIM_EINVAL;
  if (dtype != 0 && dtype != 1) return COGAIM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((M + RL_BM - 1) / RL_BM), block(512);
  int dbg = 0;
#ifdef COGAIM_TOOLS
  // tools builds only: RL_DBG=1 leaves the MFMAs out, RL_DBG=2 runs a single K step (wrong results, for timing the parts)
  if (const char* e = getenv("RL_DBG")) dbg = atoi(e);
#endif
  if (dtype == 0)
    hipLaunchKernelGGL(gemm_rowln_kernel<__bf16>, grid, block, 0, st, (const __bf16*)A, (const __bf16*)W, bias, ls, x, gamma,
                       beta, (__bf16*)h, M, K, lda, ldw, ldx, ldh, eps, dbg);
  else
    hipLaunchKernelGGL(gemm_rowln_kernel<_Float16>, grid, block, 0, st, (const _Float16*)A, (const _Float16*)W, bias, ls, x,
                       gamma, beta, (_Float16*)h, M, K, lda, ldw, ldx, ldh, eps, dbg);
  return COGAIM_LAUNCH_CHECK();
}
