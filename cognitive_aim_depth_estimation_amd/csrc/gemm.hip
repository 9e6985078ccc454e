// MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous,
// 16-bit bf16/fp16, fp32 accumulate) with the fused epilogues the DINOv2 blocks
// and the focal stream need.
//
// Tile 128x128x64, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 =
// 4x4 v_mfma_f32_16x16x32 tiles.  Operand tiles are staged global -> LDS with
// 16-byte global_load_lds (no VGPR round trip) into two buffers; the LDS image
// is [row][8 x 16-B chunk] with the chunk index XORed by (row>>1)&7, applied on
// the per-lane SOURCE address (the DMA destination is lane-linear) and again on
// the ds_read_b128 fragment reads, which makes those reads bank-conflict free.
// Rows past M / N are clamped on load and masked on store, so no padding of the
// caller's buffers is needed.  K must be a multiple of 64.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

struct GemmShape {
  const void* A; const void* W;
  int M, N, K, lda, ldw;
  long long strideA, strideW;   // per blockIdx.y batch, in elements
};

// ---- epilogues -------------------------------------------------------------
// Every epilogue is split into loads and stores so that a kernel can issue ALL loads of a
// tile (column constants, residual values) before the first store: a store to the output may
// alias a later load as far as the compiler can tell, so loads written after stores are
// serialised behind them, one memory round trip per fragment.
//   SWAP orientation (a lane owns 4 or 16 consecutive columns of one row):
//     load_cols(n0, b, l)            column constants of columns n0..n0+3 (bias, LayerScale)
//     load_res(z, m, n0)             per-element addend (kResidual epilogues only)
//     finish4n(z, m, n0, acc, b, l, r)
//     finish8x2n(z, m, n0, acc[4], b[4])     kWide epilogues: columns [n0, n0+8) from acc[0..1], [n0+32, n0+40) from acc[2..3]
//     store4n(z, m, n0, acc)         all of the above for one fragment (128x128 kernel, fix-up)
//   plain orientation (a lane owns 4 consecutive rows of one column): load_col1 / finish4m / store4m.

template <class T>
__device__ __forceinline__ void store_half8(T* p, const float (&y)[8]) {   // 8 contiguous 16-bit values, 16-B aligned
  alignas(16) T h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (T)y[i];
  // streaming store: a tile's 128 KB would otherwise be allocated in L2, where the 256 CUs' simultaneous
  // epilogues evict the operand panels; the consumers re-read these outputs from HBM either way
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(*reinterpret_cast<const u4*>(h), reinterpret_cast<u4*>(p));
}

// A row's two 8-column runs (16-bit, 16 B each, 64 B apart) written so that every store instruction covers whole
// 128-byte lines: lanes fr and fr ^ 8 of a 16-lane row exchange one run each (DPP row_ror:8), after which the lanes
// fr < 8 hold the first halves of rows fr and fr + 8 and the lanes fr >= 8 the second halves; the first
// instruction then writes rows 0..7 of the fragment complete, the second rows 8..15.  `row` points at this lane's
// own row and run 0; rows of a fragment are `ldc` elements apart.  Without the exchange each instruction leaves
// 16 half-written lines behind.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ void pack_half8x2_lines(const float (&y0)[8], const float (&y1)[8], u32x4& first, u32x4& second) {
  alignas(16) T h0[8], h1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { h0[i] = (T)y0[i]; h1[i] = (T)y1[i]; }
  const u32x4 r0 = *reinterpret_cast<const u32x4*>(h0), r1 = *reinterpret_cast<const u32x4*>(h1);
  const bool low = (threadIdx.x & 8) == 0;              // fr < 8 (fr = lane & 15)
  u32x4 give, got;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    give[i] = low ? r1[i] : r0[i];
    got[i] = (unsigned)__builtin_amdgcn_mov_dpp((int)give[i], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { first[i] = low ? r0[i] : got[i]; second[i] = low ? got[i] : r1[i]; }
}
// where the two payloads of pack_half8x2_lines go: lane fr < 8: rows fr, fr + 8, bytes 0..63;  lane fr >= 8: rows
// fr - 8, fr, bytes 64..127 (`row` = this lane's own row and run 0)
template <class T>
__device__ __forceinline__ void line_targets(T* row, long long ldc, T*& p_first, T*& p_second) {
  const bool low = (threadIdx.x & 8) == 0;
  p_first = low ? row : row - 8 * ldc + 32;
  p_second = low ? row + 8 * ldc : row + 32;
}
template <class T>
__device__ __forceinline__ void store_half8x2_lines(T* row, long long ldc, const float (&y0)[8], const float (&y1)[8]) {
  u32x4 first, second;
  pack_half8x2_lines<T>(y0, y1, first, second);
  T *p_first, *p_second;
  line_targets<T>(row, ldc, p_first, p_second);
  __builtin_nontemporal_store(first, reinterpret_cast<u32x4*>(p_first));
  __builtin_nontemporal_store(second, reinterpret_cast<u32x4*>(p_second));
}

// fp16 range guard.  Outputs stored as fp16 become inf above 65504; the epilogues that store 16-bit values keep a
// per-lane running maximum of |y| (one v_max3_f32 per two outputs, operands are results of vector instructions,
// never raw MFMA results) and raise *range_flag when a tile has seen a value fp16 cannot hold.  bf16 has fp32's
// exponent range: nothing is tracked there.
template <class T> struct TracksRange { static constexpr bool value = false; };
template <> struct TracksRange<_Float16> { static constexpr bool value = true; };
template <class T, int NV>
__device__ __forceinline__ void track_peak(float& peak, const float (&y)[NV]) {
  if constexpr (TracksRange<T>::value) {
#pragma unroll
    for (int i = 0; i + 1 < NV; i += 2)
      asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(peak) : "v"(y[i]), "v"(y[i + 1]));
  }
}
template <class T>
__device__ __forceinline__ void flush_peak_to(int* range_flag, float peak) {
  if constexpr (TracksRange<T>::value) {
    if (range_flag && peak > 65504.0f) atomicOr(range_flag, 1);
  }
}

// Guarded 16-byte load without a branch: the address is moved to element 0 of the array when the guard
// fails and the value zeroed afterwards, so that consecutive loads of an epilogue issue back to back
// (a branch per load makes the compiler wait for each load before the next one is issued).
__device__ __forceinline__ float4 ldg4(const float* base, size_t off, bool ok) {
  float4 v = *reinterpret_cast<const float4*>(base + (ok ? off : 0));
  if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
  return v;
}

template <class T> struct EpiBiasHalf {           // out16 = (acc + bias) [* scale for n < scale_cols]
  static constexpr int kReduce = 0;   // 1 / 2: the tile is reduced in registers instead of stored (EpiRowStats / EpiColSum)
  static constexpr bool kWide = true, kResidual = false;
  static constexpr int kColVecs = 1;   // per-column vectors load_cols fills: 0 none, 1 bias, 2 bias and scale
  T* out; const float* bias; int ldc; int scale_cols; float scale; int M, N; int* range_flag;
  __device__ __forceinline__ void load_cols(int n0, float4& b, float4&) const { b = ldg4(bias, n0, n0 < N); }
  __device__ __forceinline__ float4 load_res(int, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void flush_peak(float peak) const { flush_peak_to<T>(range_flag, peak); }
  __device__ __forceinline__ void finish4n(int, int m, int n0, f32x4 v, float4 b, float4, float4, float& peak) const {
    if (m >= M || n0 >= N) return;
    const float s = n0 < scale_cols ? scale : 1.0f;
    const float y[4] = {(v[0] + b.x) * s, (v[1] + b.y) * s, (v[2] + b.z) * s, (v[3] + b.w) * s};
    track_peak<T, 4>(peak, y);
    *reinterpret_cast<Half4<T>*>(out + (size_t)m * ldc + n0) = to_half4<T>(y[0], y[1], y[2], y[3]);
  }
  __device__ __forceinline__ void finish8x2n(int, int m, int n0, const f32x4 (&v)[4], const float4 (&b)[4],
                                             float& peak) const {
    if (m >= M) return;
    if (n0 + 40 > N || (ldc & 7) || (scale_cols & 7)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) finish4n(0, m, n0 + 4 * (q & 1) + 32 * (q >> 1), v[q], b[q], b[q], b[q], peak);
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float s = n0 + 32 * h < scale_cols ? scale : 1.0f;
      float y[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        y[4 * q] = (v[2 * h + q][0] + b[2 * h + q].x) * s; y[4 * q + 1] = (v[2 * h + q][1] + b[2 * h + q].y) * s;
        y[4 * q + 2] = (v[2 * h + q][2] + b[2 * h + q].z) * s; y[4 * q + 3] = (v[2 * h + q][3] + b[2 * h + q].w) * s;
      }
      track_peak<T, 8>(peak, y);
      store_half8<T>(out + (size_t)m * ldc + n0 + 32 * h, y);
    }
  }
  // the same for a tile that lies wholly inside the matrix (every lane of the wave takes part): whole-line stores
  __device__ __forceinline__ bool lines_ok() const { return !(ldc & 7) && !(scale_cols & 7); }
  __device__ __forceinline__ void finish8x2n_lines(int, int m, int n0, const f32x4 (&v)[4], const float4 (&b)[4],
                                                   float& peak) const {
    float y[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float s = n0 + 32 * h < scale_cols ? scale : 1.0f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        y[h][4 * q] = (v[2 * h + q][0] + b[2 * h + q].x) * s; y[h][4 * q + 1] = (v[2 * h + q][1] + b[2 * h + q].y) * s;
        y[h][4 * q + 2] = (v[2 * h + q][2] + b[2 * h + q].z) * s; y[h][4 * q + 3] = (v[2 * h + q][3] + b[2 * h + q].w) * s;
      }
    }
    track_peak<T, 8>(peak, y[0]); track_peak<T, 8>(peak, y[1]);
    store_half8x2_lines<T>(out + (size_t)m * ldc + n0, ldc, y[0], y[1]);
  }
  // the two 16-byte payloads finish8x2n_lines stores, left in registers (kernels that issue the stores later)
  __device__ __forceinline__ void pack8x2n_lines(int n0, const f32x4 (&v)[4], const float4 (&b)[4], float& peak,
                                                 u32x4& first, u32x4& second) const {
    float y[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float s = n0 + 32 * h < scale_cols ? scale : 1.0f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        y[h][4 * q] = (v[2 * h + q][0] + b[2 * h + q].x) * s; y[h][4 * q + 1] = (v[2 * h + q][1] + b[2 * h + q].y) * s;
        y[h][4 * q + 2] = (v[2 * h + q][2] + b[2 * h + q].z) * s; y[h][4 * q + 3] = (v[2 * h + q][3] + b[2 * h + q].w) * s;
      }
    }
    track_peak<T, 8>(peak, y[0]); track_peak<T, 8>(peak, y[1]);
    pack_half8x2_lines<T>(y[0], y[1], first, second);
  }
  __device__ __forceinline__ void store4n(int, int m, int n0, f32x4 v) const {
    float4 b, l; load_cols(n0, b, l);
    float peak = 0.f; finish4n(0, m, n0, v, b, b, b, peak); flush_peak(peak);
  }
};

// Exact-erf GELU (hidden_act="gelu"): gelu(x) = max(x, 0) - |x| Phi(-|x|), which covers both signs
// without a select and has no cancellation in the tail.  log2 Phi(-a) is smooth and close to a
// parabola, so a degree-6 polynomial on [0, 6] (weighted least squares on the error of a Phi(-a))
// followed by one exp2 gives |abs error| <= 5e-7 in fp32 -- far below the 16-bit rounding of the
// output -- for one transcendental instead of the two (rcp, exp) of the Abramowitz-Stegun form;
// a transcendental costs four plain VALU slots on this chip.  Beyond a = 6, a Phi(-a) < 6e-9 a.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float a = fminf(ax, 6.0f);
  float p = 3.275438741e-05f;
  p = __builtin_fmaf(p, a, -7.609388156e-04f);
  p = __builtin_fmaf(p, a, 8.037637648e-03f);
  p = __builtin_fmaf(p, a, -5.332307151e-02f);
  p = __builtin_fmaf(p, a, -4.588527004e-01f);
  p = __builtin_fmaf(p, a, -1.151170468e+00f);
  p = __builtin_fmaf(p, a, -9.999967080e-01f);
  return __builtin_fmaf(-ax, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}

// The same function on eight values at once, as four pairs walked in lock step: the Horner steps become packed
// v_pk_fma_f32 (two elements per issue slot; the scalar form above issues one v_fmaak per element and step, each
// waiting for the one before it) and four independent chains hide the latency of the dependent steps.  Same
// operations in the same order per element, so the results equal gelu_erf bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_erf_x8(float (&y)[8]) {
  f32x2 x[4], ax[4], p[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    x[q] = f32x2{y[2 * q], y[2 * q + 1]};
    ax[q] = f32x2{fabsf(x[q].x), fabsf(x[q].y)};
    p[q] = f32x2{3.275438741e-05f, 3.275438741e-05f};
  }
  f32x2 a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = f32x2{fminf(ax[q].x, 6.0f), fminf(ax[q].y, 6.0f)};
  constexpr float C[6] = {-7.609388156e-04f, 8.037637648e-03f, -5.332307151e-02f, -4.588527004e-01f, -1.151170468e+00f,
                          -9.999967080e-01f};
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = __builtin_elementwise_fma(p[q], a[q], f32x2{C[k], C[k]});
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(p[q].x), __builtin_amdgcn_exp2f(p[q].y)};
    const f32x2 r = __builtin_elementwise_fma(-ax[q], e, f32x2{fmaxf(x[q].x, 0.f), fmaxf(x[q].y, 0.f)});
    y[2 * q] = r.x; y[2 * q + 1] = r.y;
  }
}

template <class T> struct EpiBiasGeluHalf {
  static constexpr int kReduce = 0;   // 1 / 2: the tile is reduced in registers instead of stored (EpiRowStats / EpiColSum)
  static constexpr bool kWide = true, kResidual = false;
  static constexpr int kColVecs = 1;   // per-column vectors load_cols fills: 0 none, 1 bias, 2 bias and scale
  T* out; const float* bias; int ldc; int M, N; int* range_flag;
  __device__ __forceinline__ void load_cols(int n0, float4& b, float4&) const { b = ldg4(bias, n0, n0 < N); }
  __device__ __forceinline__ float4 load_res(int, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void flush_peak(float peak) const { flush_peak_to<T>(range_flag, peak); }
  __device__ __forceinline__ void finish4n(int, int m, int n0, f32x4 v, float4 b, float4, float4, float& peak) const {
    if (m >= M || n0 >= N) return;
    const float y[4] = {gelu_erf(v[0] + b.x), gelu_erf(v[1] + b.y), gelu_erf(v[2] + b.z), gelu_erf(v[3] + b.w)};
    track_peak<T, 4>(peak, y);
    *reinterpret_cast<Half4<T>*>(out + (size_t)m * ldc + n0) = to_half4<T>(y[0], y[1], y[2], y[3]);
  }
  __device__ __forceinline__ void finish8x2n(int, int m, int n0, const f32x4 (&v)[4], const float4 (&b)[4],
                                             float& peak) const {
    if (m >= M) return;
    if (n0 + 40 > N || (ldc & 7)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) finish4n(0, m, n0 + 4 * (q & 1) + 32 * (q >> 1), v[q], b[q], b[q], b[q], peak);
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float y[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        y[4 * q] = v[2 * h + q][0] + b[2 * h + q].x; y[4 * q + 1] = v[2 * h + q][1] + b[2 * h + q].y;
        y[4 * q + 2] = v[2 * h + q][2] + b[2 * h + q].z; y[4 * q + 3] = v[2 * h + q][3] + b[2 * h + q].w;
      }
      gelu_erf_x8(y);
      track_peak<T, 8>(peak, y);
      store_half8<T>(out + (size_t)m * ldc + n0 + 32 * h, y);
    }
  }
  __device__ __forceinline__ bool lines_ok() const { return !(ldc & 7); }
  __device__ __forceinline__ void finish8x2n_lines(int, int m, int n0, const f32x4 (&v)[4], const float4 (&b)[4],
                                                   float& peak) const {
    float y[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        y[h][4 * q] = v[2 * h + q][0] + b[2 * h + q].x; y[h][4 * q + 1] = v[2 * h + q][1] + b[2 * h + q].y;
        y[h][4 * q + 2] = v[2 * h + q][2] + b[2 * h + q].z; y[h][4 * q + 3] = v[2 * h + q][3] + b[2 * h + q].w;
      }
      gelu_erf_x8(y[h]);
    }
    track_peak<T, 8>(peak, y[0]); track_peak<T, 8>(peak, y[1]);
    store_half8x2_lines<T>(out + (size_t)m * ldc + n0, ldc, y[0], y[1]);
  }
  // the two 16-byte payloads finish8x2n_lines stores, left in registers (kernels that issue the stores later)
  __device__ __forceinline__ void pack8x2n_lines(int, const f32x4 (&v)[4], const float4 (&b)[4], float& peak,
                                                 u32x4& first, u32x4& second) const {
    float y[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        y[h][4 * q] = v[2 * h + q][0] + b[2 * h + q].x; y[h][4 * q + 1] = v[2 * h + q][1] + b[2 * h + q].y;
        y[h][4 * q + 2] = v[2 * h + q][2] + b[2 * h + q].z; y[h][4 * q + 3] = v[2 * h + q][3] + b[2 * h + q].w;
      }
      gelu_erf_x8(y[h]);
    }
    track_peak<T, 8>(peak, y[0]); track_peak<T, 8>(peak, y[1]);
    pack_half8x2_lines<T>(y[0], y[1], first, second);
  }
  __device__ __forceinline__ void store4n(int, int m, int n0, f32x4 v) const {
    float4 b, l; load_cols(n0, b, l);
    float peak = 0.f; finish4n(0, m, n0, v, b, b, b, peak); flush_peak(peak);
  }
};

struct EpiResidualF32 {                            // x += (acc + bias) * layerscale   (in place)
  static constexpr int kReduce = 0;   // 1 / 2: the tile is reduced in registers instead of stored (EpiRowStats / EpiColSum)
    static constexpr bool kWide = false, kResidual = true;
  static constexpr int kColVecs = 2;   // per-column vectors load_cols fills: 0 none, 1 bias, 2 bias and scale
  float* x; const float* bias; const float* ls; int ldc; int M, N;
  __device__ __forceinline__ void load_cols(int n0, float4& b, float4& l) const {
    b = ldg4(bias, n0, n0 < N); l = ldg4(ls, n0, n0 < N);
  }
  __device__ __forceinline__ float4 load_res(int, int m, int n0) const {
    return ldg4(x, (size_t)m * ldc + n0, m < M && n0 < N);
  }
  __device__ __forceinline__ void flush_peak(float) const {}
  __device__ __forceinline__ void finish4n(int, int m, int n0, f32x4 v, float4 b, float4 l, float4 r, float&) const {
    if (m >= M || n0 >= N) return;
    r.x += (v[0] + b.x) * l.x; r.y += (v[1] + b.y) * l.y;
    r.z += (v[2] + b.z) * l.z; r.w += (v[3] + b.w) * l.w;
    *reinterpret_cast<float4*>(x + (size_t)m * ldc + n0) = r;
  }
  __device__ __forceinline__ void store4n(int, int m, int n0, f32x4 v) const {
    float4 b, l; load_cols(n0, b, l); float pk = 0.f; finish4n(0, m, n0, v, b, l, load_res(0, m, n0), pk);
  }
};

// epilogues that read-modify-write an fp32 matrix in place through plain pointers (gemm256f_kernel pipelines those)
template <class E> struct InPlaceRMW { static constexpr bool value = false; };
template <> struct InPlaceRMW<EpiResidualF32> { static constexpr bool value = true; };

struct EpiStoreF32 {                               // batched plain fp32 store (focal scores)
  static constexpr int kReduce = 0;   // 1 / 2: the tile is reduced in registers instead of stored (EpiRowStats / EpiColSum)
  static constexpr bool kWide = false, kResidual = false;
  static constexpr int kColVecs = 0;   // per-column vectors load_cols fills: 0 none, 1 bias, 2 bias and scale
  float* out; int ldc; long long strideC; int M, N;
  __device__ __forceinline__ void load_cols(int, float4&, float4&) const {}
  __device__ __forceinline__ float4 load_res(int, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void flush_peak(float) const {}
  __device__ __forceinline__ void finish4n(int z, int m, int n0, f32x4 v, float4, float4, float4, float&) const {
    if (m >= M || n0 >= N) return;
    *reinterpret_cast<float4*>(out + z * strideC + (size_t)m * ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ __forceinline__ void store4n(int z, int m, int n0, f32x4 v) const {
    const float4 o = make_float4(0.f, 0.f, 0.f, 0.f); float pk = 0.f; finish4n(z, m, n0, v, o, o, o, pk);
  }
};

struct EpiPatchEmbed {                             // token row b*T+1+p = acc + bias + pos[1+p]
  static constexpr int kReduce = 0;   // 1 / 2: the tile is reduced in registers instead of stored (EpiRowStats / EpiColSum)
  static constexpr bool kWide = false, kResidual = true;
  static constexpr int kColVecs = 1;   // per-column vectors load_cols fills: 0 none, 1 bias, 2 bias and scale
  float* x; const float* bias; const float* pos; int D; int P; int M;   // P patches per image
  __device__ __forceinline__ void load_cols(int n0, float4& b, float4&) const { b = ldg4(bias, n0, n0 < D); }
  __device__ __forceinline__ float4 load_res(int, int m, int n0) const {
    const int p = m % P;
    return ldg4(pos, (size_t)(1 + p) * D + n0, m < M && n0 < D);
  }
  __device__ __forceinline__ void flush_peak(float) const {}
  __device__ __forceinline__ void finish4n(int, int m, int n0, f32x4 v, float4 bi, float4, float4 po, float&) const {
    if (m >= M || n0 >= D) return;
    const int b = m / P, p = m - b * P;
    *reinterpret_cast<float4*>(x + ((size_t)b * (P + 1) + 1 + p) * D + n0) =
        make_float4(v[0] + bi.x + po.x, v[1] + bi.y + po.y, v[2] + bi.z + po.z, v[3] + bi.w + po.w);
  }
  __device__ __forceinline__ void store4n(int, int m, int n0, f32x4 v) const {
    float4 b, l; load_cols(n0, b, l); float pk = 0.f; finish4n(0, m, n0, v, b, b, load_res(0, m, n0), pk);
  }
};

// Focal scores that are never stored (large grids).  Both epilogues see a whole 256x256 tile of
// S = Y X~^T (log2 domain) in the accumulators of the persistent kernel and reduce it there:
//   EpiRowStats: per row and per 64-column wave strip, max and sum of exp2(S - max)   -> pmax / psum [z][row][part]
//   EpiColSum:   per column, sum over the tile half's 128 rows of w_n exp2(S - m_n) / l_n -> part [z][block][col]
// The element-wise hooks below exist only because the shared kernels name them; these epilogues run on
// whole tiles of the persistent kernel only (no K-split, no quadrant items, no 128x128 kernel).
struct EpiRowStats {
  static constexpr int kReduce = 1;
  static constexpr bool kWide = false, kResidual = false;
  static constexpr int kColVecs = 0;
  float* pmax; float* psum; int parts; int M, N;      // parts = 4 * column tiles; arrays [z][M][parts]
  __device__ __forceinline__ void load_cols(int, float4&, float4&) const {}
  __device__ __forceinline__ float4 load_res(int, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void flush_peak(float) const {}
  __device__ __forceinline__ void finish4n(int, int, int, f32x4, float4, float4, float4, float&) const {}
  __device__ __forceinline__ void store4n(int, int, int, f32x4) const {}
  // acc: the wave's 128 x 64 accumulators; rows m_base + 16 i + fr, columns n_base + 16 jj + 4 fq + e
  __device__ __forceinline__ void reduce_tile(int z, int m_base, int n_base, int part_idx, const f32x4 (&acc)[8][4],
                                              int fr, int fq) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v[16];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[4 * jj + e] = (n_base + 16 * jj + 4 * fq + e < N) ? acc[i][jj][e] : -INFINITY;
      float mx = v[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sm = 0.f;
      if (mx > -INFINITY) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sm += __builtin_amdgcn_exp2f(v[k] - mx);
      }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const int m = m_base + 16 * i + fr;
      if (fq == 0 && m < M) {
        const size_t o = ((size_t)z * M + m) * parts + part_idx;
        pmax[o] = mx; psum[o] = sm;
      }
    }
  }
};

struct EpiColSum {
  static constexpr int kReduce = 2;
  static constexpr bool kWide = false, kResidual = false;
  static constexpr int kColVecs = 0;
  const float* rmax; const float* rinv; const float* roww; float* part; int nblk; int M, N;   // part [z][nblk][N]
  __device__ __forceinline__ void load_cols(int, float4&, float4&) const {}
  __device__ __forceinline__ float4 load_res(int, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void flush_peak(float) const {}
  __device__ __forceinline__ void finish4n(int, int, int, f32x4, float4, float4, float4, float&) const {}
  __device__ __forceinline__ void store4n(int, int, int, f32x4) const {}
  __device__ __forceinline__ void reduce_tile(int z, int m_base, int n_base, int block, const f32x4 (&acc)[8][4],
                                              int fr, int fq) const {
    float mrow[8], srow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m_base + 16 * i + fr;
      const bool ok = m < M;
      const size_t o = (size_t)z * M + (ok ? m : 0);
      mrow[i] = ok ? rmax[o] : INFINITY;          // rows past M: exp2(S - inf) = 0, never inf * 0
      float sc = rinv[o];
      if (roww) sc *= roww[o];
      srow[i] = ok ? sc : 0.f;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += __builtin_amdgcn_exp2f(acc[i][jj][e] - mrow[i]) * srow[i];   // rows ascending: fixed order
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o, 64);      // the 16 rows (lanes fr) of a fragment
        c[e] = a;
      }
      const int n0 = n_base + 16 * jj + 4 * fq;
      if (fr == 0) {
        float* dst = part + ((size_t)z * nblk + block) * N + n0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + e < N) dst[e] = c[e];
      }
    }
  }
};

// bijective XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous tile range
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <class T, class Epi, bool SWAP>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmShape g, Epi epi) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // [stage][A|W]
  using H = Half<T>;
  using v8 = typename H::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int z = blockIdx.y;
  const T* A = reinterpret_cast<const T*>(g.A) + z * g.strideA;
  const T* W = reinterpret_cast<const T*>(g.W) + z * g.strideW;

  // per-lane source pointers for the 4 DMA instructions per operand per stage
  const T* srcA[4]; const T* srcW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int ra = tm * BM + r; ra = ra < g.M ? ra : g.M - 1;
    int rw = tn * BN + r; rw = rw < g.N ? rw : g.N - 1;
    srcA[i] = A + (size_t)ra * g.lda + c * 8;
    srcW[i] = W + (size_t)rw * g.ldw + c * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* a = smem + buf * 2 * TILE_BYTES + wave * 32 * 128;
    char* w = a + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(srcA[i] + k0, a + i * 8 * 128);
      glds16(srcW[i] + k0, w + i * 8 * 128);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int rowA = wm * 64 + fr, rowW = wn * 64 + fr;       // + 16*i
  const int swA = (rowA >> 1) & 7, swW = (rowW >> 1) & 7;   // unchanged by + 16*i

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) stage((t + 1) & 1, (t + 1) * BK);
    const char* as = smem + (t & 1) * 2 * TILE_BYTES;
    const char* ws = as + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const v8*>(as + (rowA + 16 * i) * 128 + (((4 * s + fq) ^ swA) << 4));
        wf[i] = *reinterpret_cast<const v8*>(ws + (rowW + 16 * i) * 128 + (((4 * s + fq) ^ swW) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = SWAP ? H::mfma16(wf[j], af[i], acc[i][j]) : H::mfma16(af[i], wf[j], acc[i][j]);
    }
    __syncthreads();
  }

  const int m_base = tm * BM + wm * 64, n_base = tn * BN + wn * 64;
  if constexpr (SWAP) {
    float peak = 0.f;
    float4 cb[4], cl[4], xr[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) epi.load_cols(n_base + 16 * j + 4 * fq, cb[j], cl[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xr[i][j] = Epi::kResidual ? epi.load_res(z, m_base + 16 * i + fr, n_base + 16 * j + 4 * fq) : cb[j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        epi.finish4n(z, m_base + 16 * i + fr, n_base + 16 * j + 4 * fq, acc[i][j], cb[j], cl[j], xr[i][j], peak);
    epi.flush_peak(peak);
  } else {
    float cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = epi.load_col1(n_base + 16 * j + fr);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) epi.finish4m(z, m_base + 16 * i + 4 * fq, n_base + 16 * j + fr, acc[i][j], cb[j]);
  }
}

constexpr int BM2 = 256, BN2 = 256;
constexpr int OP_BYTES = 256 * BK * 2;        // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;     // 64 KiB

#ifdef COGAIM_STAMPS
__device__ long long g_stamp_buf[256 * 12 * 4];
__device__ long long g_clock_buf[4 * 256 * 4];  // [problem class][workgroup]: shader clock and 100-MHz clock at entry and exit
#endif
int g_gemm_slice_cap = 0;   // changed only by builds with -DCOGAIM_TOOLS (tools/gemm_bench.py)
int g_gemm_quads = 1;       // 0: K-split + fix-up launch for every partial last round (ditto)
int g_gemm_band = -1;       // tile numbering of the persistent kernel: -1 = chosen per shape (ditto)

int g_gemm_prefetch = 1;    // whole rounds of large problems on gemm256f_kernel (0: gemm256p_kernel for everything;
                            // changed only by -DCOGAIM_TOOLS builds)
int g_gemm_fast_tail = 3;    // partial last rounds: 3 / 2 as items of gemm256f_kernel's first workgroups ahead of / behind
                             // their whole tiles, 1 on gemm256q_kernel behind it, 0 on gemm256p_kernel behind it
                             // (changed only by -DCOGAIM_TOOLS builds)
template <class T, class Epi> void launch_tail(const GemmShape& g, const Epi& epi, const struct PersistSched& tail, hipStream_t st);
int g_gemm_skip_tail = 0;    // timing-only: leave the partial last round of tiles out (wrong results); tools builds only
int g_gemm_wide = 0;       // whole rounds on the four-wave gemm256w_kernel instead (tools builds: knob 13 / 12)
int g_wide_diag = 0;
template <class T, class Epi> void launch_wide(const GemmShape& g, const Epi& epi, const struct PersistSched& sc, hipStream_t st, int diag);
int g_prefetch_diag = 0;    // timing-only variant of gemm256f_kernel (bit mask, see gemm_prefetch.h); tools builds only
template <class T, class Epi> void launch_prefetch(const GemmShape& g, const Epi& epi, const struct PersistSched& sc, hipStream_t st);
inline bool prefetch_eligible(const GemmShape& g, long long tiles);
#include "gemm_persistent.h"
#include "gemm_tail.h"
#include "gemm_prefetch.h"
#if defined(COGAIM_TOOLS) && defined(COGAIM_WIDE)
#include "gemm_wide.h"     // four-wave 128x128-per-wave kernel: a measured experiment (tools/wt_build.sh with COGAIM_EXTRA_FLAGS=-DCOGAIM_WIDE)
#else
template <class T, class Epi> void launch_wide(const GemmShape&, const Epi&, const struct PersistSched&, hipStream_t, int) {}
#endif
#ifdef COGAIM_TOOLS
#include "gemm_stream.h"      // 128x256 streaming kernel: a measured experiment, not part of the shipped library
#endif

template <class T, class Epi>
void launch_prefetch(const GemmShape& g, const Epi& epi, const PersistSched& sc, hipStream_t st) {
#ifdef COGAIM_TOOLS
#define F_DIAG_CASE(d_) \
  case d_: hipLaunchKernelGGL((gemm256f_kernel<T, Epi, d_>), dim3(256), dim3(512), 0, st, g, epi, sc); break;
  switch (g_prefetch_diag) {
    F_DIAG_CASE(1) F_DIAG_CASE(2) F_DIAG_CASE(4) F_DIAG_CASE(8) F_DIAG_CASE(16) F_DIAG_CASE(7)
    F_DIAG_CASE(9) F_DIAG_CASE(10) F_DIAG_CASE(12) F_DIAG_CASE(24) F_DIAG_CASE(13) F_DIAG_CASE(15) F_DIAG_CASE(29)
    default: hipLaunchKernelGGL((gemm256f_kernel<T, Epi, 0>), dim3(256), dim3(512), 0, st, g, epi, sc);
  }
#undef F_DIAG_CASE
#else
  hipLaunchKernelGGL((gemm256f_kernel<T, Epi, 0>), dim3(256), dim3(512), 0, st, g, epi, sc);
#endif
}

template <class T, class Epi>
void launch_tail(const GemmShape& g, const Epi& epi, const PersistSched& tail, hipStream_t st) {
  if constexpr (Epi::kReduce == 0)
    hipLaunchKernelGGL((gemm256q_kernel<T, Epi>), dim3(tail.n_split * tail.slices * 4), dim3(512), 0, st, g, epi, tail);
}

int g_stream_diag = 0; // timing-only variant of the streaming kernel (bit mask, see gemm_stream.h); tools builds only
int g_gemm_diag = 0;   // timing-only diagnostic variant (wrong results): reachable only in -DCOGAIM_TOOLS builds
int g_gemm_algo = 2;   // 0: 128x128 kernel everywhere; 1: 256x256 ping-pong kernel, one tile per workgroup;
                       // 2: persistent 256x256 ping-pong kernel (large problems only for 1 and 2); 3: as 2, and the
                       // 128x256 streaming kernel for the 16-bit-output epilogues; changed only by -DCOGAIM_TOOLS builds

#ifdef COGAIM_TOOLS
extern "C" int cogaim_set_gemm_algo(int algo);
// COGAIM_GEMM_KNOBS="5,6": codes of cogaim_set_gemm_algo applied once, for programs that have no call of it (bench.py)
static void tools_env_once() {
  static const bool done = [] {
    if (const char* e = getenv("COGAIM_GEMM_KNOBS"))
      for (const char* p = e; *p;) {
        cogaim_set_gemm_algo(atoi(p));
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
    return true;
  }();
  (void)done;
}
#endif

template <class T, class Epi, bool SWAP>
int launch(const GemmShape& g, const Epi& epi, int batch, hipStream_t st, bool ragged_n = false,
           void* workspace = nullptr, long long workspace_bytes = 0) {
#ifdef COGAIM_TOOLS
  tools_env_once();
#endif
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK) != 0 || (!ragged_n && (g.N & 3) != 0) ||
      (g.lda & 7) != 0 || (g.ldw & 7) != 0 || g.lda < g.K || g.ldw < g.K ||
      ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) != 0 || !g.A || !g.W)
    return COGAIM_EINVAL;
  const int tm2 = (g.M + BM2 - 1) / BM2, tn2 = (g.N + BN2 - 1) / BN2;
  const long long tiles2 = (long long)tm2 * tn2;
  if constexpr (Epi::kReduce != 0) {       // in-register reductions exist in the persistent kernel only
    if (tiles2 * batch >= (1 << 30)) return COGAIM_EINVAL;
    return launch_persistent<T, Epi, SWAP>(g, epi, batch, st, nullptr, 0, true, 0);
  }
#ifdef COGAIM_TOOLS
  if constexpr (Epi::kWide && SWAP) {
    if (g_gemm_algo == 3 && stream_eligible<Epi>(g, batch)) return launch_stream<T, Epi>(g, epi, st, g_stream_diag);
  }
#endif
  if (g_gemm_algo >= 1 && tiles2 * batch >= 128 && tiles2 * batch < (1 << 30))
    return launch_persistent<T, Epi, SWAP>(g, epi, batch, st, workspace, workspace_bytes, g_gemm_algo >= 2,
                                           g_gemm_diag);
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_kernel<T, Epi, SWAP>), dim3(tiles, batch), dim3(256), 0, st, g, epi);
  return COGAIM_LAUNCH_CHECK();
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F16) \
  ((dtype) == 0 ? (CALL_BF16) : (dtype) == 1 ? (CALL_F16) : COGAIM_EINVAL)

extern "C" {

// out16[M,N] = (A * W^T + bias), columns [0, scale_cols) additionally multiplied by scale.
static bool bad_out(const void* out, int ldc, int N, int align) {
  return !out || (ldc & 3) || ldc < N || (reinterpret_cast<uintptr_t>(out) & (align - 1));
}

int cogaim_gemm_bias_half(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                          int lda, int ldw, int ldc, int scale_cols, float scale, int dtype, void* workspace,
                          long long workspace_bytes, int* range_flag, void* stream) {
  GemmShape g{A, W, M, N, K, lda, ldw, 0, 0};
  if (bad_out(out, ldc, N, 16) || (scale_cols & 3) || scale_cols < 0 || !bias) return COGAIM_EINVAL;
  return DISPATCH_T(dtype,
      (launch<__bf16, EpiBiasHalf<__bf16>, true>(g, {(__bf16*)out, bias, ldc, scale_cols, scale, M, N, range_flag}, 1, (hipStream_t)stream, false, workspace, workspace_bytes)),
      (launch<_Float16, EpiBiasHalf<_Float16>, true>(g, {(_Float16*)out, bias, ldc, scale_cols, scale, M, N, range_flag}, 1, (hipStream_t)stream, false, workspace, workspace_bytes)));
}

// out16[M,N] = gelu_erf(A * W^T + bias)
int cogaim_gemm_bias_gelu_half(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                               int lda, int ldw, int ldc, int dtype, void* workspace, long long workspace_bytes,
                               int* range_flag, void* stream) {
  GemmShape g{A, W, M, N, K, lda, ldw, 0, 0};
  if (bad_out(out, ldc, N, 16) || !bias) return COGAIM_EINVAL;
  return DISPATCH_T(dtype,
      (launch<__bf16, EpiBiasGeluHalf<__bf16>, true>(g, {(__bf16*)out, bias, ldc, M, N, range_flag}, 1, (hipStream_t)stream, false, workspace, workspace_bytes)),
      (launch<_Float16, EpiBiasGeluHalf<_Float16>, true>(g, {(_Float16*)out, bias, ldc, M, N, range_flag}, 1, (hipStream_t)stream, false, workspace, workspace_bytes)));
}

// x[M,N] += (A * W^T + bias) * layerscale      (fp32 residual stream, in place)
int cogaim_gemm_residual_f32(const void* A, const void* W, const float* bias, const float* layerscale, float* x,
                             int M, int N, int K, int lda, int ldw, int ldc, int dtype, void* workspace,
                             long long workspace_bytes, void* stream) {
  GemmShape g{A, W, M, N, K, lda, ldw, 0, 0};
  if (bad_out(x, ldc, N, 16) || !bias || !layerscale) return COGAIM_EINVAL;
  EpiResidualF32 e{x, bias, layerscale, ldc, M, N};
  return DISPATCH_T(dtype, (launch<__bf16, EpiResidualF32, true>(g, e, 1, (hipStream_t)stream, false, workspace, workspace_bytes)),
                    (launch<_Float16, EpiResidualF32, true>(g, e, 1, (hipStream_t)stream, false, workspace, workspace_bytes)));
}

// out32[z][M,N] = A[z] * W[z]^T for z < batch (strides in elements)
int cogaim_gemm_batched_f32(const void* A, const void* W, float* out, int M, int N, int K, int lda, int ldw, int ldc,
                            long long strideA, long long strideW, long long strideC, int batch, int dtype,
                            void* stream) {
  GemmShape g{A, W, M, N, K, lda, ldw, strideA, strideW};
  // N need not be a multiple of 4: the last 4-wide store of a row may spill into the row padding
  if ((ldc & 3) || ldc < ((N + 3) & ~3) || (strideC & 3) || (strideA & 7) || (strideW & 7) || batch <= 0 || !out ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return COGAIM_EINVAL;
  EpiStoreF32 e{out, ldc, strideC, M, N};
  return DISPATCH_T(dtype, (launch<__bf16, EpiStoreF32, true>(g, e, batch, (hipStream_t)stream, true)),
                    (launch<_Float16, EpiStoreF32, true>(g, e, batch, (hipStream_t)stream, true)));
}

// ---- focal scores without a score matrix in memory (src/model.py:197-204, 234 for large grids) -----------------
int cogaim_focal_rowstats_parts(int N) { return ((N + BN2 - 1) / BN2) * 4; }
int cogaim_focal_scores_colsum_blocks(int N) { return ((N + BM2 - 1) / BM2) * 2; }

// S[z] = Y[z] X[z]^T is formed tile by tile and only its row statistics leave the chip:
// pmax / psum [batch][N][cogaim_focal_rowstats_parts(N)] = max and sum of exp2(S - max) over each 64-column strip.
int cogaim_focal_scores_rowstats(const void* Y, const void* X, float* pmax, float* psum, int N, int D, int ld,
                                 long long stride, int batch, int dtype, void* stream) {
  GemmShape g{Y, X, N, N, D, ld, ld, stride, stride};
  if (!pmax || !psum || batch <= 0 || (stride & 7)) return COGAIM_EINVAL;
  EpiRowStats e{pmax, psum, cogaim_focal_rowstats_parts(N), N, N};
  return DISPATCH_T(dtype, (launch<__bf16, EpiRowStats, true>(g, e, batch, (hipStream_t)stream, true)),
                    (launch<_Float16, EpiRowStats, true>(g, e, batch, (hipStream_t)stream, true)));
}

// The tiles again, now normalised with the finished row statistics and summed over rows:
// part[z][blk][c] = sum over the 128 rows n of block blk of roww[z][n] exp2(S[n][c] - rmax[z][n]) rinv[z][n]
// (roww NULL = 1), part [batch][cogaim_focal_scores_colsum_blocks(N)][N].
int cogaim_focal_scores_colsum(const void* Y, const void* X, const float* rmax, const float* rinv, const float* roww,
                               float* part, int N, int D, int ld, long long stride, int batch, int dtype, void* stream) {
  GemmShape g{Y, X, N, N, D, ld, ld, stride, stride};
  if (!rmax || !rinv || !part || batch <= 0 || (stride & 7)) return COGAIM_EINVAL;
  EpiColSum e{rmax, rinv, roww, part, cogaim_focal_scores_colsum_blocks(N), N, N};
  return DISPATCH_T(dtype, (launch<__bf16, EpiColSum, true>(g, e, batch, (hipStream_t)stream, true)),
                    (launch<_Float16, EpiColSum, true>(g, e, batch, (hipStream_t)stream, true)));
}

// Patch-embedding GEMM: rows are im2col'ed patches (B*P rows, K padded to a multiple of 64);
// writes token rows 1..P of x[B, P+1, D] = acc + bias + pos[1+p].
int cogaim_gemm_patch_embed(const void* cols, const void* W, const float* bias, const float* pos, float* x,
                            int B, int P, int D, int K, int dtype, void* workspace, long long workspace_bytes,
                            void* stream) {
  GemmShape g{cols, W, B * P, D, K, K, K, 0, 0};
  EpiPatchEmbed e{x, bias, pos, D, P, B * P};
  return DISPATCH_T(dtype, (launch<__bf16, EpiPatchEmbed, true>(g, e, 1, (hipStream_t)stream, false, workspace, workspace_bytes)),
                    (launch<_Float16, EpiPatchEmbed, true>(g, e, 1, (hipStream_t)stream, false, workspace, workspace_bytes)));
}

#ifdef COGAIM_STAMPS
int cogaim_debug_stamps(void* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamp_buf), sizeof(long long) * 256 * 12 * 4) == hipSuccess ? 0 : -1;
}
int cogaim_debug_clock(void* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_clock_buf), sizeof(long long) * 4 * 256 * 4) == hipSuccess ? 0 : -1;
}
#endif
#ifdef COGAIM_TOOLS
// Experiment knobs of tools/gemm_bench.py; process-wide and unsynchronised, which is why the shipped library does
// not export them.  0 = 128x128 kernel for every problem; 1 = 256x256 kernel, one tile per workgroup; 2 (default) =
// persistent 256x256 kernel (1 and 2 apply to large problems only); the larger codes set the other g_gemm_* values.
int cogaim_set_gemm_algo(int algo) {
  if (algo >= 1000) { g_gemm_slice_cap = algo - 1000; return COGAIM_OK; }
#ifdef COGAIM_WIDE
  if (algo >= 600 && algo < 632) { g_wide_diag = algo - 600; g_gemm_wide = 1; g_gemm_prefetch = 1; g_gemm_algo = 2; g_gemm_diag = 0; return COGAIM_OK; }
  if (algo == 12 || algo == 13) { g_gemm_wide = algo - 12; g_wide_diag = 0; return COGAIM_OK; }
#endif
  if (algo >= 500 && algo < 532) { g_prefetch_diag = algo - 500; g_gemm_prefetch = 1; g_gemm_algo = 2; g_gemm_diag = 0; return COGAIM_OK; }
  if (algo >= 400 && algo < 432) { g_stream_diag = algo - 400; g_gemm_algo = 3; g_gemm_diag = 0; return COGAIM_OK; }
  if (algo >= 8 && algo <= 11) { g_gemm_fast_tail = algo - 8; return COGAIM_OK; }       // 11 / 10: tails inside gemm256f ahead of / behind the whole tiles, 9: gemm256q, 8: gemm256p
  if (algo == 6 || algo == 7) { g_gemm_skip_tail = algo == 6; return COGAIM_OK; }       // 6: timing-only, no partial last rounds
  if (algo == 4 || algo == 5) { g_gemm_prefetch = algo - 4; g_prefetch_diag = 0; return COGAIM_OK; }   // 5: on, 4: off
  if (algo == 300 || algo == 301) { g_gemm_quads = algo - 300; return COGAIM_OK; }
  if (algo >= 200) { g_gemm_band = algo == 299 ? -1 : algo - 200; return COGAIM_OK; }
  if (algo >= 100) { g_gemm_diag = 1; g_gemm_algo = algo - 100; return COGAIM_OK; }
  if (algo < 0 || algo > 3) return COGAIM_EINVAL;
  g_prefetch_diag = 0;
  g_gemm_algo = algo;
  g_gemm_diag = 0;
  g_stream_diag = 0;
  return COGAIM_OK;
}
#endif

// Bytes of split-K workspace that let every GEMM entry point remove its partial last round of tiles.
long long cogaim_gemm_workspace_bytes() { return 256LL * 32 * 512 * 16; }   // 256 partial tiles of 256 KiB

}  // extern "C"
