// Focal-stream kernels (reference src/model.py:128-313, 391-455) for gfx950.
//
// One FocalStream call is evaluated as
//   Xt   = tokens * scale + pe                               (focal_prep, 16-bit GEMM operand)
//   Q,K  = Xt Wq^T + bq, Xt Wk^T + bk                        (MFMA GEMM, gemm.hip)
//   S    = Q K^T / sqrt(D/8) * log2(e)                       (batched MFMA GEMM, fp32 scores)
//   m,l  = row max / row sum of exp2(S - m)                  (focal_colsum<STATS>)
//   c    = column mean of P = exp2(S - m) / l                (same kernel + attention_vector)
//   a    = renorm(clamp(renorm(c + centre bias)))            (focal_attention_vector)
//   u    = a^T P                                             (focal_colsum<weighted>)
//   w    = (u^T Xt) Wv^T + (sum u) bv                        (focal_weighted_tokens + small linears)
// The last two lines use sum_n a_n (P V)_n = (a^T P) V, so the N x N x D product
// P V and the V projection of every token are never formed.
#include "common.h"
#include <mutex>

// Partial sums of cogaim_weighted_tokens_static: device memory that belongs to the code object, one copy per GPU,
// held until the process ends.  Host code never passes this NAME to a kernel (on the host it is a stand-in,
// not a device address): the per-device address is looked up once with hipGetSymbolAddress below.
#define COGAIM_WT_STATIC_FLOATS (2 * 1024 * 1024)        /* 8 MiB */
static __device__ __align__(256) float scratch[COGAIM_WT_STATIC_FLOATS];

namespace {

// One wave per token row (no per-element index arithmetic), 16-byte loads, 8-byte stores.
// NV > 0: D = 256 * NV, every lane owns exactly NV float4 of a row; a wave takes token n of FOUR images, so the row
// of the position table is fetched once for the four (the table is read once per image otherwise: as many bytes
// again as the tokens) and all of the loads are requested before the first one is used.  NV = 0: any D (a multiple
// of 4), one wave per (image, token).
template <class T, int NV>
__global__ __launch_bounds__(256) void focal_prep_kernel(const float* __restrict__ tok, long long tok_stride_b,
                                                         const float* __restrict__ scale, const float* __restrict__ pe,
                                                         T* __restrict__ out, int B, int N, int D) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if constexpr (NV > 0) {
    const int groups = (B + 3) >> 2;
    if (item >= N * groups) return;
    const int n = item / groups, b0 = (item - n * groups) * 4;
    const float4* p4 = reinterpret_cast<const float4*>(pe + (size_t)n * D);
    float4 p[NV], x[4][NV];
    float s[4];
#pragma unroll
    for (int i = 0; i < NV; ++i) p[i] = p4[lane + 64 * i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = min(b0 + j, B - 1);
      const float4* x4 = reinterpret_cast<const float4*>(tok + b * tok_stride_b + (size_t)n * D);
#pragma unroll
      for (int i = 0; i < NV; ++i) x[j][i] = x4[lane + 64 * i];
      s[j] = scale ? scale[(size_t)b * N + n] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (b0 + j < B) {
        Half4<T>* o4 = reinterpret_cast<Half4<T>*>(out + ((size_t)(b0 + j) * N + n) * D);
#pragma unroll
        for (int i = 0; i < NV; ++i)
          o4[lane + 64 * i] = to_half4<T>(x[j][i].x * s[j] + p[i].x, x[j][i].y * s[j] + p[i].y, x[j][i].z * s[j] + p[i].z,
                                          x[j][i].w * s[j] + p[i].w);
      }
    }
  } else {
    const int row = item;
    if (row >= B * N) return;
    const int b = row / N, n = row - b * N;
    const float4* x4 = reinterpret_cast<const float4*>(tok + b * tok_stride_b + (size_t)n * D);
    const float4* p4 = reinterpret_cast<const float4*>(pe + (size_t)n * D);
    Half4<T>* o4 = reinterpret_cast<Half4<T>*>(out + (size_t)row * D);
    const float s = scale ? scale[row] : 1.0f;
    for (int c = lane; c < (D >> 2); c += 64) {
      const float4 x = x4[c], p = p4[c];
      o4[c] = to_half4<T>(x.x * s + p.x, x.y * s + p.y, x.z * s + p.z, x.w * s + p.w);
    }
  }
}

constexpr int CS_ROWS = 32;   // score rows per block

// part[b][blk][c] = sum over the block's rows n of w_n * exp2(S[n][c] - m_n) * inv_n
// STATS: m_n, inv_n are computed here (and stored); w_n = 1.   else: read m, inv; w_n = roww[b][n].
template <bool STATS>
__global__ __launch_bounds__(256) void focal_colsum_kernel(const float* __restrict__ S, int ldS, long long strideS,
                                                           float* __restrict__ rmax, float* __restrict__ rinv,
                                                           const float* __restrict__ roww, float* __restrict__ part,
                                                           int N, int nblk) {
  __shared__ float m_s[CS_ROWS], i_s[CS_ROWS];
  const int b = blockIdx.y, blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blk * CS_ROWS;
  const float* Sb = S + b * strideS;
  if constexpr (STATS) {
    // each wave owns 16 of the block's rows and walks them four at a time: the four rows' loads and the four
    // wave reductions are independent, which is what hides the memory latency of this short, strided pass
    const int n4 = N & ~3;
    for (int g = 0; g < CS_ROWS / 16; ++g) {
      const float* sr[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = r0 + wave + 4 * (4 * g + k);
        ok[k] = row < N;
        sr[k] = Sb + (size_t)(ok[k] ? row : N - 1) * ldS;       // rows are 16-B aligned (ldS % 4 == 0)
      }
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      for (int c = lane * 4; c < n4; c += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(sr[k] + c);
          mx[k] = fmaxf(fmaxf(mx[k], fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
      }
      for (int c = n4 + lane; c < N; c += 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], sr[k][c]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, 64));
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = lane * 4; c < n4; c += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(sr[k] + c);
          sum[k] += (__builtin_amdgcn_exp2f(v.x - mx[k]) + __builtin_amdgcn_exp2f(v.y - mx[k])) +
                    (__builtin_amdgcn_exp2f(v.z - mx[k]) + __builtin_amdgcn_exp2f(v.w - mx[k]));
        }
      }
      for (int c = n4 + lane; c < N; c += 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) sum[k] += __builtin_amdgcn_exp2f(sr[k][c] - mx[k]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) sum[k] += __shfl_xor(sum[k], o, 64);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = wave + 4 * (4 * g + k), row = r0 + rr;
          if (ok[k]) {
            m_s[rr] = mx[k]; i_s[rr] = 1.0f / sum[k];
            rmax[(size_t)b * N + row] = mx[k]; rinv[(size_t)b * N + row] = 1.0f / sum[k];
          } else { m_s[rr] = 0.f; i_s[rr] = 0.f; }
        }
      }
    }
  } else {
    if (tid < CS_ROWS) {
      const int row = r0 + tid;
      const bool ok = row < N;
      m_s[tid] = ok ? rmax[(size_t)b * N + row] : 0.f;
      i_s[tid] = ok ? rinv[(size_t)b * N + row] * roww[(size_t)b * N + row] : 0.f;
    }
  }
  __syncthreads();
  const int rows = min(CS_ROWS, N - r0);
  for (int c = tid; c < N; c += 256) {
    const float* sp = Sb + (size_t)r0 * ldS + c;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;      // row order inside each chain is fixed: deterministic
    int rr = 0;
    for (; rr + 3 < rows; rr += 4) {
      const float v0 = sp[(size_t)rr * ldS], v1 = sp[(size_t)(rr + 1) * ldS];
      const float v2 = sp[(size_t)(rr + 2) * ldS], v3 = sp[(size_t)(rr + 3) * ldS];
      acc0 += __builtin_amdgcn_exp2f(v0 - m_s[rr]) * i_s[rr];
      acc1 += __builtin_amdgcn_exp2f(v1 - m_s[rr + 1]) * i_s[rr + 1];
      acc2 += __builtin_amdgcn_exp2f(v2 - m_s[rr + 2]) * i_s[rr + 2];
      acc3 += __builtin_amdgcn_exp2f(v3 - m_s[rr + 3]) * i_s[rr + 3];
    }
    for (; rr < rows; ++rr) acc0 += __builtin_amdgcn_exp2f(sp[(size_t)rr * ldS] - m_s[rr]) * i_s[rr];
    part[((size_t)b * nblk + blk) * N + c] = (acc0 + acc1) + (acc2 + acc3);
  }
}

// The STATS form of the kernel above for N <= 256 * NCH, with every score read ONCE: a wave keeps four whole rows
// in registers (NCH float4 per lane and row), takes their maxima and exp2 sums from there, and adds the
// normalised rows into per-lane column sums; the four waves' column sums meet in LDS.  Row order per column is
// fixed (wave by wave, rows ascending inside a wave), so the result does not depend on timing.
template <int NCH>
__global__ __launch_bounds__(256) void focal_stats_colsum_rows_kernel(const float* __restrict__ S, int ldS,
                                                                      long long strideS, float* __restrict__ rmax,
                                                                      float* __restrict__ rinv, float* __restrict__ part,
                                                                      int N, int nblk) {
  __shared__ float4 red[4][NCH * 64];
  const int b = blockIdx.y, blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blk * CS_ROWS;
  const float* Sb = S + b * strideS;
  float4 cs[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = 0; g < CS_ROWS / 16; ++g) {
    float4 v[4][NCH];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = r0 + wave + 4 * (4 * g + k);
      ok[k] = row < N;
      const float* sr = Sb + (size_t)(ok[k] ? row : N - 1) * ldS;   // rows are 16-B aligned (ldS % 4 == 0)
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane * 4 + 256 * j;
        v[k][j] = c < N ? *reinterpret_cast<const float4*>(sr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 1 >= N) v[k][j].y = -INFINITY;                       // also covers c >= N: the whole chunk
        if (c + 2 >= N) v[k][j].z = -INFINITY;
        if (c + 3 >= N) v[k][j].w = -INFINITY;
        if (c >= N) v[k][j].x = -INFINITY;
      }
    }
    float mx[4], sum[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mx[k] = -INFINITY;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
        mx[k] = fmaxf(fmaxf(mx[k], fmaxf(v[k][j].x, v[k][j].y)), fmaxf(v[k][j].z, v[k][j].w));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, 64));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sum[k] = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        v[k][j].x = __builtin_amdgcn_exp2f(v[k][j].x - mx[k]); v[k][j].y = __builtin_amdgcn_exp2f(v[k][j].y - mx[k]);
        v[k][j].z = __builtin_amdgcn_exp2f(v[k][j].z - mx[k]); v[k][j].w = __builtin_amdgcn_exp2f(v[k][j].w - mx[k]);
        sum[k] += (v[k][j].x + v[k][j].y) + (v[k][j].z + v[k][j].w);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) sum[k] += __shfl_xor(sum[k], o, 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float inv = ok[k] ? 1.0f / sum[k] : 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        cs[j].x += v[k][j].x * inv; cs[j].y += v[k][j].y * inv; cs[j].z += v[k][j].z * inv; cs[j].w += v[k][j].w * inv;
      }
      if (lane == 0 && ok[k]) {
        const int row = r0 + wave + 4 * (4 * g + k);
        rmax[(size_t)b * N + row] = mx[k]; rinv[(size_t)b * N + row] = inv;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) red[wave][j * 64 + lane] = cs[j];
  __syncthreads();
  const float* r = reinterpret_cast<const float*>(&red[0][0]);
  for (int c = tid; c < N; c += 256)
    part[((size_t)b * nblk + blk) * N + c] =
        (r[c] + r[NCH * 256 + c]) + (r[2 * NCH * 256 + c] + r[3 * NCH * 256 + c]);
}

// rmax[r] = max_p pmax[r][p];  rinv[r] = 1 / sum_p psum[r][p] exp2(pmax[r][p] - rmax[r])     (r over B*N rows)
__global__ __launch_bounds__(256) void rowstats_combine_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                               int parts, float* __restrict__ rmax, float* __restrict__ rinv,
                                                               long long rows) {
  const long long r = blockIdx.x * 256LL + threadIdx.x;
  if (r >= rows) return;
  const float* pm = pmax + r * parts;
  const float* pl = psum + r * parts;
  float m = -INFINITY;
  for (int p = 0; p < parts; ++p) m = fmaxf(m, pm[p]);
  float l = 0.f;
  for (int p = 0; p < parts; ++p) l += pl[p] * __builtin_amdgcn_exp2f(pm[p] - m);      // parts ascending: fixed order
  rmax[r] = m; rinv[r] = 1.0f / l;
}

// colsum[b][c] = sum_blk part[b][blk][c]      (wide grid: one thread per column)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nblk,
                                                              float* __restrict__ out, int N) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const float* p = part + (size_t)b * nblk * N + c;
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 1 < nblk; k += 2) { s0 += p[(size_t)k * N]; s1 += p[(size_t)(k + 1) * N]; }
  if (k < nblk) s0 += p[(size_t)k * N];
  out[(size_t)b * N + c] = s0 + s1;
}

// Sum of six doubles over a 256-thread block; result valid in every thread.  `redd` is 24 doubles of LDS.
__device__ __forceinline__ void block_sum6_d(double (&v)[6], double* redd) {
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = wave_sum_d(v[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) redd[(threadIdx.x >> 6) * 6 + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = redd[k] + redd[6 + k] + redd[12 + k] + redd[18 + k];
}

// The three deterministic summaries of the attention matrix the reference can end up using (model.py:234-249), per
// image: cand[0] = column mean + centre bias, cand[1] = diagonal of P + bias, cand[2] = row maximum of P + bias, and
// {sum, sum of squares} of each in fp64, per image and per 256-column block of this grid: stats [B][gridDim.x][6].
// The column sums are reduced here from the per-row-block partials (one thread per column over a wide grid: parts
// ascending in two running sums, the order of reduce_partials_kernel).  P = exp2(S - rmax) * rinv, so its row maximum
// is rinv itself.  Without a score matrix in memory (S == NULL) cand[1] repeats cand[0].
__global__ __launch_bounds__(256) void focal_attention_candidates_kernel(const float* __restrict__ part, int nblk,
                                                                         const float* __restrict__ cbias,
                                                                         const float* __restrict__ S, int ldS,
                                                                         long long strideS, const float* __restrict__ rmax,
                                                                         const float* __restrict__ rinv, float* cand,
                                                                         double* __restrict__ stats, int B, int N) {
  __shared__ double redd[24];
  const int b = blockIdx.y, tid = threadIdx.x, c = blockIdx.x * 256 + tid;
  const size_t plane = (size_t)B * N, row = (size_t)b * N;
  const float invN = 1.0f / (float)N;
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (c < N) {
    const float* p = part + (size_t)b * nblk * N + c;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < nblk; k += 2) { s0 += p[(size_t)k * N]; s1 += p[(size_t)(k + 1) * N]; }
    if (k < nblk) s0 += p[(size_t)k * N];
    const float bias = cbias[c], inv = rinv[row + c];
    const float a0 = (s0 + s1) * invN + bias;
    const float a1 = S ? __builtin_amdgcn_exp2f(S[b * strideS + (size_t)c * ldS + c] - rmax[row + c]) * inv + bias : a0;
    const float a2 = inv + bias;
    cand[row + c] = a0; cand[plane + row + c] = a1; cand[2 * plane + row + c] = a2;
    acc[0] = a0; acc[1] = (double)a0 * a0;
    acc[2] = a1; acc[3] = (double)a1 * a1;
    acc[4] = a2; acc[5] = (double)a2 * a2;
  }
  block_sum6_d(acc, redd);
  if (tid < 6) stats[((size_t)b * gridDim.x + blockIdx.x) * 6 + tid] = acc[tid];
}

// model.py:242-282 per image.  The cascade `if patch_attention.var() < 1e-6` (unbiased variance over the WHOLE batch)
// picks column mean, else diagonal, else row maximum; if that is degenerate too the reference draws noise (:252-257),
// which is not reproduced: flags[slot] = 1.  Then a = renorm(clamp(renorm(choice))) and the next iteration's token
// scale (:426).  Every workgroup evaluates the cascade from all images' statistics.
__global__ __launch_bounds__(256) void focal_attention_vector_kernel(const float* __restrict__ cand,
                                                                     const double* __restrict__ stats, int have_diag,
                                                                     float* __restrict__ a,
                                                                     const float* __restrict__ scale_in,
                                                                     float* __restrict__ scale_out, float focus_strength,
                                                                     int B, int N, int* __restrict__ flags, int slot) {
  __shared__ float red[4];
  __shared__ double redd[24];
  const int b = blockIdx.x, tid = threadIdx.x;
  double tot[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int nstat = B * ((N + 255) / 256);             // stats [B][ceil(N / 256)][6] of the candidates kernel
  for (int i = tid; i < nstat; i += 256) {
#pragma unroll
    for (int k = 0; k < 6; ++k) tot[k] += stats[i * 6 + k];
  }
  block_sum6_d(tot, redd);
  const double n = (double)B * N;
  int sel = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double var = (tot[2 * k + 1] - tot[2 * k] * tot[2 * k] / n) / (n - 1.0);      // NaN for a single element: no fallback
    if (sel == k && var < 1e-6) sel = k + 1;
  }
  if (!have_diag && sel >= 1) sel = 3;          // the diagonal would be needed and is not available
  if (b == 0 && tid == 0) flags[slot] = (sel == 3) ? 1 : 0;
  const float* src = cand + (size_t)(sel < 2 ? sel : 2) * B * N + (size_t)b * N;
  float* ab = a + (size_t)b * N;
  float s1 = 0.f;
  for (int c = tid; c < N; c += 256) s1 += src[c];
  const float d = block_sum_256(s1, red) + 1e-8f;
  float s2 = 0.f;
  for (int c = tid; c < N; c += 256) {
    const float v = fmaxf(src[c] / d, 1e-8f);
    ab[c] = v; s2 += v;
  }
  const float t2 = block_sum_256(s2, red) + 1e-8f;
  for (int c = tid; c < N; c += 256) {
    const float v = ab[c] / t2;
    ab[c] = v;
    if (scale_out) scale_out[(size_t)b * N + c] = (scale_in ? scale_in[(size_t)b * N + c] : 1.0f) * (1.0f + focus_strength * v);
  }
}

// usum[b] = sum_c u[b][c]
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ u, float* __restrict__ usum, int N) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int c = tid; c < N; c += 256) s += u[(size_t)b * N + c];
  const float t = block_sum_256(s, red);
  if (tid == 0) usum[b] = t;
}

// part[k][b][chunk][d] = sum_{n in chunk} w[k][b][n] * (tok[b][n][d] * scale[b][n] + pe[n][d])   (scale, pe optional;
// k = blockIdx.z: several weight vectors per image share the tokens)
constexpr int WT_ROWS = 32;
__global__ __launch_bounds__(256) void weighted_tokens_partial_kernel(const float* __restrict__ tok,
                                                                      long long tok_stride_b,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ pe,
                                                                      const float* __restrict__ w,
                                                                      float* __restrict__ part, int nchunk, int N, int D,
                                                                      const float* __restrict__ cpart, int cnblk,
                                                                      float* __restrict__ wsum_part) {
  __shared__ float w_s[WT_ROWS];
  __shared__ float w_g[8][WT_ROWS];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int n0 = chunk * WT_ROWS, n1 = min(N, n0 + WT_ROWS);
  const float* tb = tok + b * tok_stride_b;
  if (cpart) {
    // the chunk's weights straight from the column-sum partials [B][cnblk][N]: thread (g, r) adds blocks g, g + 8, ...
    // of row n0 + r (a handful of independent loads each instead of cnblk in one thread), then the eight group sums
    // are added in ascending order: a fixed order, so the result does not depend on timing
    {
      const int r = threadIdx.x & (WT_ROWS - 1), g = threadIdx.x / WT_ROWS;      // 256 threads = 8 groups x 32 rows
      const int n = n0 + r;
      float u = 0.f;
      if (n < N) {
        const float* p = cpart + (size_t)b * cnblk * N + n;
        for (int k = g; k < cnblk; k += 8) u += p[(size_t)k * N];
      }
      w_g[g][r] = u;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      float u = 0.f;
      if (threadIdx.x < WT_ROWS) {
#pragma unroll
        for (int g = 0; g < 8; ++g) u += w_g[g][threadIdx.x];
        w_s[threadIdx.x] = u;
      }
      const float t = wave_sum(u);
      if (threadIdx.x == 0 && wsum_part) wsum_part[(size_t)b * nchunk + chunk] = t;
    }
  } else {
    w += (size_t)blockIdx.z * gridDim.y * N;
    if (threadIdx.x < WT_ROWS) w_s[threadIdx.x] = w[(size_t)b * N + min(n0 + (int)threadIdx.x, N - 1)];
  }
  __syncthreads();
  part += (size_t)blockIdx.z * gridDim.y * nchunk * D;
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight rows' loads are requested together (a block is 32 rows of one image: few blocks per CU, so the loop is
    // bound by how many loads each thread keeps in flight); rows are added in ascending order whatever the batching
    for (int m0 = n0; m0 < n1; m0 += 8) {
      float4 x[8], p[8];
      float wn[8], sc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = min(m0 + j, n1 - 1);
        x[j] = *reinterpret_cast<const float4*>(tb + (size_t)n * D + c);
        wn[j] = w_s[n - n0];
        sc[j] = scale ? scale[(size_t)b * N + n] : 1.0f;
        p[j] = pe ? *reinterpret_cast<const float4*>(pe + (size_t)n * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (m0 + j < n1) {
          float4 v = x[j];
          if (scale) { v.x *= sc[j]; v.y *= sc[j]; v.z *= sc[j]; v.w *= sc[j]; }
          if (pe) { v.x += p[j].x; v.y += p[j].y; v.z += p[j].z; v.w += p[j].w; }
          acc.x += wn[j] * v.x; acc.y += wn[j] * v.y; acc.z += wn[j] * v.z; acc.w += wn[j] * v.w;
        }
      }
    }
    *reinterpret_cast<float4*>(part + ((size_t)b * nchunk + chunk) * D + c) = acc;
  }
}

// guided[k][b][c] = softmax_c((0.7 mask[k][c] + 0.3 a[b][c]) / 0.05)     (model.py:1401-1409; k = blockIdx.y)
__global__ __launch_bounds__(256) void guided_attention_kernel(const float* __restrict__ a, const float* __restrict__ mask,
                                                               float* __restrict__ out, int N, float alpha, float inv_temp) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* ab = a + (size_t)b * N;
  float* ob = out + ((size_t)blockIdx.y * gridDim.x + b) * N;
  mask += (size_t)blockIdx.y * N;
  float mx = -INFINITY;
  for (int c = tid; c < N; c += 256) {
    const float v = (alpha * mask[c] + (1.0f - alpha) * ab[c]) * inv_temp;
    ob[c] = v; mx = fmaxf(mx, v);
  }
  mx = block_max_256(mx, red);
  float s = 0.f;
  for (int c = tid; c < N; c += 256) { const float e = expf(ob[c] - mx); ob[c] = e; s += e; }
  const float t = block_sum_256(s, red);
  for (int c = tid; c < N; c += 256) ob[c] = ob[c] / t;
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int cogaim_guided_attention_multi(const float* a, const float* masks, float* out, int B, int N, int K, float alpha,
                                  float temperature, void* stream);

int cogaim_focal_colsum_rows() { return CS_ROWS; }

// out16[b*N+n][:] = tok[b][n][:] * scale[b][n] + pe[n][:]    (tok rows are D apart, images tok_stride_b apart)
int cogaim_focal_prep(const float* tok, long long tok_stride_b, const float* scale, const float* pe, void* out, int B,
                      int N, int D, int dtype, void* stream) {
  if (B <= 0 || N <= 0 || (D & 3)) return COGAIM_EINVAL;
  if ((long long)B * N >= (1ll << 31)) return COGAIM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype != 0 && dtype != 1) return COGAIM_EINVAL;
  int g;                  // workgroups of four waves: a wave per (token, four images) or per (image, token)
#define COGAIM_PREP(NV_)                                                                                                     \
  {                                                                                                                         \
    if (dtype == 0) hipLaunchKernelGGL((focal_prep_kernel<__bf16, NV_>), dim3(g), dim3(256), 0, st, tok, tok_stride_b, scale, pe, (__bf16*)out, B, N, D); \
    else hipLaunchKernelGGL((focal_prep_kernel<_Float16, NV_>), dim3(g), dim3(256), 0, st, tok, tok_stride_b, scale, pe, (_Float16*)out, B, N, D); \
    return COGAIM_LAUNCH_CHECK();                                                                                            \
  }
  g = (N * ((B + 3) / 4) + 3) / 4;
  if (D == 768) COGAIM_PREP(3)
  if (D == 1024) COGAIM_PREP(4)
  g = (B * N + 3) / 4;
  COGAIM_PREP(0)
#undef COGAIM_PREP
}

// Row statistics of the log2-domain scores + per-row-block column sums of the softmax.
// part: [B][nblk][N] with nblk = ceil(N / cogaim_focal_colsum_rows()).
int cogaim_focal_softmax_stats_colsum(const float* S, int ldS, long long strideS, float* rmax, float* rinv, float* part,
                                      int B, int N, void* stream) {
  if (B <= 0 || N <= 0 || ldS < N) return COGAIM_EINVAL;
  const int nblk = (N + CS_ROWS - 1) / CS_ROWS;
  if (N <= 1536 && !(ldS & 3))       // the 518-pixel grid (N = 1369) and smaller: rows fit in registers
    hipLaunchKernelGGL(focal_stats_colsum_rows_kernel<6>, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, S, ldS,
                       strideS, rmax, rinv, part, N, nblk);
  else
    hipLaunchKernelGGL(focal_colsum_kernel<true>, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, S, ldS, strideS,
                       rmax, rinv, (const float*)nullptr, part, N, nblk);
  return COGAIM_LAUNCH_CHECK();
}

// part[b][blk][c] = sum_{n in blk} roww[b][n] * P[b][n][c]
int cogaim_focal_weighted_colsum(const float* S, int ldS, long long strideS, const float* rmax, const float* rinv,
                                 const float* roww, float* part, int B, int N, void* stream) {
  if (B <= 0 || N <= 0 || ldS < N) return COGAIM_EINVAL;
  const int nblk = (N + CS_ROWS - 1) / CS_ROWS;
  hipLaunchKernelGGL(focal_colsum_kernel<false>, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, S, ldS, strideS,
                     (float*)rmax, (float*)rinv, roww, part, N, nblk);
  return COGAIM_LAUNCH_CHECK();
}

// cand [3][B][N], stats fp64 [B][ceil(N / 256)][6]: the three candidate summaries of model.py:234-249 and their {sum, sum
// of squares} per image and 256-column block.  part [B, nblk, N] column-sum partials (nblk <= 0 = ceil(N / cogaim_focal_colsum_rows())); S may be
// NULL when no score matrix is kept in memory (the diagonal candidate then repeats the column mean).
int cogaim_focal_attention_candidates(const float* part, int nblk, const float* center_bias, const float* S, int ldS,
                                      long long strideS, const float* rmax, const float* rinv, float* cand, double* stats,
                                      int B, int N, void* stream) {
  if (B <= 0 || N <= 0 || (S && ldS < N)) return COGAIM_EINVAL;
  if (nblk <= 0) nblk = (N + CS_ROWS - 1) / CS_ROWS;
  // one launch: the reduction of the partials needs a wide grid (one thread per column; folded into a one-workgroup-
  // per-image kernel it took 40 us instead of 8: 43 dependent-latency loads per column on 16 CUs), so the candidates
  // and their statistics are formed on that grid too, the statistics as per-block partial sums
  hipLaunchKernelGGL(focal_attention_candidates_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, part,
                     nblk, center_bias, S, ldS, strideS, rmax, rinv, cand, stats, B, N);
  return COGAIM_LAUNCH_CHECK();
}

// a[b][:] from the candidates by the reference's variance cascade; flags[slot] = 1 when even the last deterministic
// candidate is degenerate (or the diagonal is needed and have_diag is 0), else 0;
// scale_out = scale_in * (1 + focus_strength * a) when scale_out != NULL (scale_in NULL = ones).
int cogaim_focal_attention_vector(const float* cand, const double* stats, int have_diag, float* a, const float* scale_in,
                                  float* scale_out, float focus_strength, int B, int N, int* flags, int slot, void* stream) {
  if (B <= 0 || N <= 0 || !flags || slot < 0) return COGAIM_EINVAL;
  hipLaunchKernelGGL(focal_attention_vector_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cand, stats, have_diag, a,
                     scale_in, scale_out, focus_strength, B, N, flags, slot);
  return COGAIM_LAUNCH_CHECK();
}

int cogaim_focal_reduce_u(const float* part, float* u, float* usum, int B, int N, int nblk, void* stream) {
  if (B <= 0 || N <= 0) return COGAIM_EINVAL;
  if (nblk <= 0) nblk = (N + CS_ROWS - 1) / CS_ROWS;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, part, nblk, u, N);
  hipLaunchKernelGGL(row_sum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)u, usum, N);
  return COGAIM_LAUNCH_CHECK();
}

// Row statistics from the per-strip partials of cogaim_focal_scores_rowstats: pmax / psum [B*N][parts] -> rmax / rinv [B*N]
int cogaim_focal_rowstats_combine(const float* pmax, const float* psum, int parts, float* rmax, float* rinv, int B, int N,
                                  void* stream) {
  if (B <= 0 || N <= 0 || parts <= 0) return COGAIM_EINVAL;
  const long long rows = (long long)B * N;
  hipLaunchKernelGGL(rowstats_combine_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pmax,
                     psum, parts, rmax, rinv, rows);
  return COGAIM_LAUNCH_CHECK();
}

int cogaim_weighted_tokens_rows() { return WT_ROWS; }

// workspace: fp32 [K * B * ceil(N / cogaim_weighted_tokens_rows()) * D]; w [K, B, N] -> out [K, B, D]
static int weighted_tokens_launch(const float* tok, long long tok_stride_b, const float* scale, const float* pe,
                                  const float* w, float* out, float* workspace, int B, int N, int D, int K,
                                  void* stream, const float* cpart = nullptr, int cnblk = 0, float* wsum_part = nullptr) {
  if (B <= 0 || N <= 0 || D <= 0 || K <= 0 || K > 65535 || (long long)K * B > 65535 || (D & 3) || (tok_stride_b & 3) ||
      !workspace || (!w && !cpart) || (cpart && (cnblk <= 0 || K != 1)))
    return COGAIM_EINVAL;
  const int nchunk = (N + WT_ROWS - 1) / WT_ROWS;
  hipLaunchKernelGGL(weighted_tokens_partial_kernel, dim3(nchunk, B, K), dim3(256), 0, (hipStream_t)stream, tok,
                     tok_stride_b, scale, pe, w, workspace, nchunk, N, D, cpart, cnblk, wsum_part);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((D + 255) / 256, K * B), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nchunk, out, D);
  return COGAIM_LAUNCH_CHECK();
}

int cogaim_weighted_tokens(const float* tok, long long tok_stride_b, const float* scale, const float* pe, const float* w,
                           float* out, float* workspace, int B, int N, int D, void* stream) {
  return weighted_tokens_launch(tok, tok_stride_b, scale, pe, w, out, workspace, B, N, D, 1, stream);
}

// cogaim_weighted_tokens with the weights given as column-sum partials: w[b][n] = sum_blk cpart[b][blk][n] (cnblk
// blocks; <= 0 = ceil(N / cogaim_focal_colsum_rows())), reduced inside the token pass instead of by launches of their
// own; wsum_part [B][ceil(N / cogaim_weighted_tokens_rows())] (may be NULL) receives the weights' sums per chunk of rows.
int cogaim_weighted_tokens_parts(const float* tok, long long tok_stride_b, const float* scale, const float* pe,
                                 const float* cpart, int cnblk, float* out, float* workspace, float* wsum_part, int B,
                                 int N, int D, void* stream) {
  if (!cpart) return COGAIM_EINVAL;
  if (cnblk <= 0) cnblk = (N + CS_ROWS - 1) / CS_ROWS;
  return weighted_tokens_launch(tok, tok_stride_b, scale, pe, nullptr, out, workspace, B, N, D, 1, stream, cpart, cnblk,
                                wsum_part);
}

// K weight vectors per image in one pass over the tokens (instruction sweep): w [K, B, N] -> out [K, B, D];
// slice k equals what cogaim_weighted_tokens returns for w[k], bit for bit.
int cogaim_weighted_tokens_multi(const float* tok, long long tok_stride_b, const float* w, float* out, float* workspace,
                                 int B, int N, int D, int K, void* stream) {
  return weighted_tokens_launch(tok, tok_stride_b, nullptr, nullptr, w, out, workspace, B, N, D, K, stream);
}

// cogaim_weighted_tokens without a workspace argument and without any allocation: the partial sums live in the
// library's own static device array (8 MiB per GPU).  The array's address on the current device is resolved on
// the first call there (hipGetSymbolAddress; make that call outside a stream capture) and kept.  Every call on
// one GPU shares the array: calls that overlap on a GPU (different streams or host threads) race.  Nothing in
// it has to start at zero -- the partial kernel writes every element the reduction reads.
int cogaim_weighted_tokens_static(const float* tok, long long tok_stride_b, const float* scale, const float* pe,
                                  const float* w, float* out, int B, int N, int D, void* stream) {
  constexpr int MAX_DEV = 64;
  static std::mutex mu;
  static float* dev_ptr[MAX_DEV] = {};
  if (B <= 0 || N <= 0 || D <= 0) return COGAIM_EINVAL;
  const long long need = (long long)B * ((N + WT_ROWS - 1) / WT_ROWS) * D;
  if (need > (long long)COGAIM_WT_STATIC_FLOATS) return COGAIM_EINVAL;     // beyond the fixed 8 MiB of partial sums
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_rc(e);
  if (dev < 0 || dev >= MAX_DEV) return COGAIM_EINVAL;
  float* ws;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!dev_ptr[dev]) {
      void* p = nullptr;
      e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !p) return e != hipSuccess ? hip_rc(e) : COGAIM_EINVAL;
      dev_ptr[dev] = static_cast<float*>(p);
    }
    ws = dev_ptr[dev];
  }
  return weighted_tokens_launch(tok, tok_stride_b, scale, pe, w, out, ws, B, N, D, 1, stream);
}

int cogaim_guided_attention(const float* a, const float* mask, float* out, int B, int N, float alpha, float temperature,
                            void* stream) {
  return cogaim_guided_attention_multi(a, mask, out, B, N, 1, alpha, temperature, stream);
}

// masks [K, N] against one base attention a [B, N] -> out [K, B, N]
int cogaim_guided_attention_multi(const float* a, const float* masks, float* out, int B, int N, int K, float alpha,
                                  float temperature, void* stream) {
  if (B <= 0 || N <= 0 || K <= 0 || K > 65535 || temperature <= 0.f) return COGAIM_EINVAL;
  hipLaunchKernelGGL(guided_attention_kernel, dim3(B, K), dim3(256), 0, (hipStream_t)stream, a, masks, out, N, alpha,
                     1.0f / temperature);
  return COGAIM_LAUNCH_CHECK();
}

}  // extern "C"
