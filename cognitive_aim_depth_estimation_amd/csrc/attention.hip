// Fused multi-head self-attention (flash style) for the DINOv2 blocks on gfx950.
// Head dim is fixed at 64.  Computes ctx = softmax(Q K^T) V per (image, head)
// with the score matrix never leaving registers.
//
// Layout choices (all for the 64-lane v_mfma_f32_32x32x16 fragment maps):
//  * scores are produced TRANSPOSED, S^T = K Q^T, so a lane owns one query
//    column: the running max / sum / rescale factors are lane-local scalars
//    and the only cross-lane step per tile is one v_permlane32_swap;
//  * the S^T accumulator (rows = keys) is converted to 16-bit in place and fed
//    straight back as the B operand of O^T += V^T P^T (no LDS round trip);
//  * V stays row-major ([token][d], straight out of the fused QKV projection): its
//    tiles are staged [key][d] in LDS and read with ds_read_b64_tr_b16, which hands
//    each lane 4 consecutive keys of its d column -- the A operand of O^T += V^T P^T
//    without a transposed copy in memory;
//  * K and V tiles reach LDS by 16-byte LDS-DMA (global_load_lds) into the buffer the previous
//    iteration read, one tile ahead of the math: no staging registers, no ds_write.
// Q is expected pre-multiplied by dh^-0.5 * log2(e): softmax runs on exp2.
#include "common.h"
#include <type_traits>

namespace {

constexpr int KV = 64;            // keys per tile
constexpr int QB = 128;           // queries per block (4 waves x 32)
constexpr int K_TILE = KV * 128;  // K and V tiles: 64 keys x 64 d x 2 B, chunk-swizzled
constexpr float RESCALE_THR = 6.0f;   // log2 units

template <class T>
__global__ __launch_bounds__(256, 2) void mhsa_qkv_kernel(const T* __restrict__ qk, T* __restrict__ ctx, int Tn, int H, int ldqk,
                                                       int D) {
  __shared__ __attribute__((aligned(16))) char smem[4 * K_TILE];     // K[2] | V[2], 64 rows x 128 B each
  using Hf = Half<T>;
  using v8 = typename Hf::v8;
  typedef T v4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  // Workgroups are dealt round-robin over the 8 XCDs (private L2 each): remap the linear id so that the
  // query blocks of one (image, head) -- which all stream the same K/V -- are neighbours on ONE XCD.
  const int nqb = (Tn + QB - 1) / QB;
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int bh = lid / nqb, b = bh / H, h = bh - b * H;
  const int q0 = (lid - bh * nqb) * QB + wave * 32;
  const size_t row0 = (size_t)b * Tn;

  // Q fragments: B operand, lane (q = ql, half hh) holds Q[q][16 s + 8 hh .. +7]
  v8 qf[4];
  {
    int q = q0 + ql; q = q < Tn ? q : Tn - 1;
    const T* qp = qk + (row0 + q) * ldqk + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const v8*>(qp + 16 * s);
  }

  // Staging: K and V tiles go global -> LDS by 16-byte LDS-DMA, no register round trip.  A DMA instruction of a wave
  // fills 1 KiB = 8 tile rows (lane l: row l >> 3, 16-byte position l & 7); the chunk swizzles of the two tile
  // layouts are applied on the per-lane SOURCE address.  Wave w brings rows 16 w .. 16 w + 15 of both tiles.
  const T* kbase = qk + D + h * 64;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int sr0 = 16 * uwave + (lane >> 3), sr1 = sr0 + 8, sp = lane & 7;
  const int kc0 = (sp ^ ((sr0 >> 1) & 7)) * 8, kc1 = (sp ^ ((sr1 >> 1) & 7)) * 8;                   // element offsets
  const int vc0 = D + (sp ^ (((sr0 >> 1) & 1) << 2)) * 8, vc1 = D + (sp ^ (((sr1 >> 1) & 1) << 2)) * 8;
#define LOAD_TILE(kv0_, buf_)                                                                    \
  {                                                                                              \
    int key0 = (kv0_) + sr0; key0 = key0 < Tn ? key0 : Tn - 1;                                   \
    int key1 = (kv0_) + sr1; key1 = key1 < Tn ? key1 : Tn - 1;                                   \
    const T* r0_ = kbase + (row0 + key0) * ldqk;                                                 \
    const T* r1_ = kbase + (row0 + key1) * ldqk;                                                 \
    char* ks_ = smem + (buf_) * K_TILE + 2048 * uwave;                                           \
    char* vs_ = smem + (2 + (buf_)) * K_TILE + 2048 * uwave;                                     \
    glds16(r0_ + kc0, ks_); glds16(r1_ + kc1, ks_ + 1024);                                       \
    glds16(r0_ + vc0, vs_); glds16(r1_ + vc1, vs_ + 1024);                                       \
  }
  // the same for a tile that lies wholly inside the sequence: a wave-uniform base that advances by one tile and four
  // fixed 32-bit lane offsets (scalar-base addressing: no vector arithmetic per tile; the clamping form above
  // costs ~20 vector instructions a tile)
  unsigned kb_lo, kb_hi;       // forced into scalar registers (the block indices come out of vector divisions)
  {
    const unsigned long long p = reinterpret_cast<unsigned long long>(kbase + row0 * ldqk);
    kb_lo = __builtin_amdgcn_readfirstlane((unsigned)p); kb_hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  }
  const unsigned ok0 = (unsigned)(sr0 * ldqk + kc0) * 2u, ok1 = (unsigned)(sr1 * ldqk + kc1) * 2u;
  const unsigned ov0 = (unsigned)(sr0 * ldqk + vc0) * 2u, ov1 = (unsigned)(sr1 * ldqk + vc1) * 2u;
  const unsigned long long tile_step = (unsigned long long)KV * ldqk * 2;
  // (the base passes through an empty asm: otherwise the loop optimiser turns base + offset back into four
  // per-lane 64-bit pointers that it increments with vector instructions)
#define LOAD_TILE_INSIDE(buf_)                                                                   \
  {                                                                                              \
    char* ks_ = smem + (buf_) * K_TILE + 2048 * uwave;                                           \
    char* vs_ = smem + (2 + (buf_)) * K_TILE + 2048 * uwave;                                     \
    asm volatile("" : "+s"(kb_lo), "+s"(kb_hi));                                                 \
    const unsigned long long kb64_ = ((unsigned long long)kb_hi << 32) | kb_lo;                  \
    const char* kb_u_ = reinterpret_cast<const char*>(kb64_);                                    \
    unsigned a_ = ok0, b_ = ok1, c_ = ov0, d_ = ov1;     /* opaque too: a hoisted 64-bit copy loses the scalar-base form */ \
    asm volatile("" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_));                                   \
    glds16(kb_u_ + a_, ks_); glds16(kb_u_ + b_, ks_ + 1024);                                     \
    glds16(kb_u_ + c_, vs_); glds16(kb_u_ + d_, vs_ + 1024);                                     \
    const unsigned long long nx_ = kb64_ + tile_step;                                            \
    kb_lo = (unsigned)nx_; kb_hi = (unsigned)(nx_ >> 32);                                        \
  }
#define TILE_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  // transposed-read addresses inside a V tile: inside its group of 16 lanes, lane 4q + p points at row (key)
  // 4 hh + q, columns (d) 16 dhalf + 4p .. 4p+3 of the d block; rows whose bit 1 is set keep their two 64-B
  // halves exchanged (conflict-free over the 32-lane halves of the instruction)
  typedef short s4 __attribute__((ext_vector_type(4)));
  int vtr0, vtr1;
  {
    const int q = (lane & 15) >> 2, p = lane & 3, dhalf = (lane >> 4) & 1;
    const int chunk = (2 * dhalf + (p >> 1)) ^ (((q >> 1) & 1) << 2);
    vtr0 = (4 * hh + q) * 128 + chunk * 16 + 8 * (p & 1);
    vtr1 = vtr0 ^ 64;
  }
  f32x16 o[2], negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; negm[i] = 0.f; }
  float m_run = 0.f, l_run = 0.f;

  const int nt = (Tn + KV - 1) / KV;
  if (nt > 1) LOAD_TILE_INSIDE(0) else LOAD_TILE(0, 0);
  TILE_LANDED();
  __syncthreads();
  // One key tile.  The buffer index is a compile-time constant (the loop below alternates two copies of this body), so
  // every LDS address of the tile is a loop-invariant register plus an immediate offset.
  // A wave whose 32 queries all lie past the sequence end (the last query block's upper waves) has nothing to compute:
  // it stages its share of every tile, meets the barriers, and leaves its SIMD to the co-resident workgroups.
  if (__builtin_amdgcn_readfirstlane(q0) >= Tn) {
    for (int t = 0; t + 1 < nt; ++t) {
      const int nb = (t & 1) ^ 1;
      if (t + 2 < nt) LOAD_TILE_INSIDE(nb) else LOAD_TILE((t + 1) * KV, nb);
      TILE_LANDED();
      __syncthreads();
    }
    return;
  }
  // LAST: the sequence's last tile (keys past the end are masked, nothing is staged behind it, no barrier after it);
  // NKB: how many of its two 32-key halves hold keys at all.  A half of nothing but padding contributes exact
  // zeros to the row sums and to O, so leaving it out changes no bit of the result.
  auto tile = [&](const int t, auto buf_c, auto last_c, auto nkb_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value, NKB = decltype(nkb_c)::value;
    constexpr bool LAST = decltype(last_c)::value != 0;
    const int kv0 = t * KV;
    // the next tile goes straight into the other buffer (last read one iteration ago, before the barrier)
    if constexpr (!LAST) {
      if (t + 2 < nt) LOAD_TILE_INSIDE(BUF ^ 1)     // only the last tile can reach past the sequence end
      else LOAD_TILE(kv0 + KV, BUF ^ 1);
    }
    const char* ks = smem + BUF * K_TILE;
    const char* vs = smem + (2 + BUF) * K_TILE;

    // S^T = K Q^T - m_run : s[kb][i] = S^T[key kb*32 + (i&3) + 8(i>>2) + 4hh][query ql], already relative
    // to the running maximum (the accumulator starts at -m_run, so no subtraction pass is needed)
    // All eight K fragments are requested before the first MFMA and the two accumulator chains alternate: left to
    // itself the compiler reads one fragment, waits for it, and issues one MFMA that depends on the previous one.
    f32x16 s[NKB];
    v8 kf[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int r = kb * 32 + ql;
      const int sw = (r >> 1) & 7;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        kf[kb][s4] = *reinterpret_cast<const v8*>(ks + r * 128 + (((2 * s4 + hh) ^ sw) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) s[kb] = Hf::mfma32(kf[kb][0], qf[0], negm);     // negm: 16 registers holding -m_run
#pragma unroll
    for (int s4 = 1; s4 < 4; ++s4)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) s[kb] = Hf::mfma32(kf[kb][s4], qf[s4], s[kb]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LAST) {
      if (kv0 + KV > Tn) {   // mask keys past the sequence end
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (key >= Tn) s[kb][i] = -INFINITY;
          }
      }
    }
    // tile maximum per query column (lane-local + one half swap), relative to m_run
    float mx;
    {   // four independent chains (a single 31-deep dependent chain of v_max would serialise the wave)
      float m4[4];
      if constexpr (NKB == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) m4[c] = fmaxf(s[0][c], s[1][c]);
#pragma unroll
        for (int i = 4; i < 16; ++i) m4[i & 3] = fmaxf(fmaxf(m4[i & 3], s[0][i]), s[1][i]);      // v_max3_f32
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) m4[c] = s[0][c];
#pragma unroll
        for (int i = 4; i < 16; ++i) m4[i & 3] = fmaxf(m4[i & 3], s[0][i]);
      }
      mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    }
    {
      const unsigned u = __float_as_uint(mx);
      auto sw2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      mx = fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
    }
    // Deferred rescale: the running maximum only moves when some query of the wave sees a score more
    // than RESCALE_THR (log2 units) above it; until then exp2 runs on the accumulator as it is and the
    // probabilities are bounded by 2^RESCALE_THR, exact in 16-bit floating point.  The first tile always
    // sets the maximum (m_run starts at 0 there).
    const bool rescale = (t == 0) || !__all(mx <= RESCALE_THR);
    if (rescale) {
      const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      m_run += d;
      l_run *= alpha;
      const float nm = -m_run;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha; o[1][i] *= alpha;
        s[0][i] -= d;
        if constexpr (NKB == 2) s[1][i] -= d;
        // written in place (tied operand): as a plain assignment the new values get registers of their own and the
        // common path, which does not rescale, pays sixteen copies per tile to follow them
        asm volatile("v_mov_b32 %0, %1" : "+v"(negm[i]) : "v"(nm));
      }
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 l2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};        // two v_pk_add_f32 chains: 16 adds for the 32 values
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        s[kb][i] = __builtin_amdgcn_exp2f(s[kb][i]);
        s[kb][i + 1] = __builtin_amdgcn_exp2f(s[kb][i + 1]);
        l2[(i >> 1) & 1] += f32x2{s[kb][i], s[kb][i + 1]};
      }
    // l2[0] = {sum of elements 0 mod 4, 1 mod 4}, l2[1] = {2 mod 4, 3 mod 4}: combined in the association of four
    // scalar chains, (c0 + c1) + (c2 + c3).  The 16-bit roundings downstream amplify a last-place change of this
    // sum into a different draw of the whole forward's rounding error, so the order is part of the pinned numerics.
    l_run += (l2[0][0] + l2[0][1]) + (l2[1][0] + l2[1][1]);

    // O^T += V^T P^T : A = V^T fragment (rows = d), B = P^T fragment (the S^T registers)
    // A operand = V^T fragment: rows d = dt*32 + ql, element j = key 16 g + 4 hh + 8 (j >> 2) + (j & 3), g = 2 kb + ks2.
    // V sits row-major ([key][d]) in LDS; ds_read_b64_tr_b16 hands each lane 4 consecutive keys of its d.
    // The transposed reads are written as asm with hand-counted waits: through the builtin the compiler puts a
    // vmcnt(0) in front of every one of them while the next tile's LDS-DMA is in flight (it cannot tell that the
    // DMA fills the other buffer), which would expose the DMA's latency in the middle of every tile.  Two groups of
    // four reads stay in flight ahead of the MFMAs that use them (LDS operations return in order).
    const unsigned vs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)vs;
    const unsigned va0 = vs_lds + vtr0, va1 = vs_lds + vtr1;
    s4 vlo[4][2], vhi[4][2];
#define V_READ(g_)                                                                                                    \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"                     \
                 "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"                          \
                 : "=&v"(vlo[g_][0]), "=&v"(vhi[g_][0]), "=&v"(vlo[g_][1]), "=&v"(vhi[g_][1])                           \
                 : "v"(va0), "v"(va1), "n"((g_) * 2048), "n"((g_) * 2048 + 1024) : "memory")
#define V_WAIT(g_, left_)                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #left_ ")"                                                                       \
                 : "+v"(vlo[g_][0]), "+v"(vhi[g_][0]), "+v"(vlo[g_][1]), "+v"(vhi[g_][1]))
#define PV(g_)                                                                                                         \
    {                                                                                                                  \
      v8 pf;                                                                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) pf[j] = (T)s[(g_) >> 1][8 * ((g_) & 1) + j];                       \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                               \
        const v8 vf = __builtin_bit_cast(v8, __builtin_shufflevector(vlo[g_][dt], vhi[g_][dt], 0, 1, 2, 3, 4, 5, 6, 7)); \
        o[dt] = Hf::mfma32(vf, pf, o[dt]);                                                                             \
      }                                                                                                                \
    }
    V_READ(0); V_READ(1);
    V_WAIT(0, 4); PV(0);
    if constexpr (NKB == 2) {
      V_READ(2);
      V_WAIT(1, 4); PV(1);
      V_READ(3);
      V_WAIT(2, 4); PV(2);
      V_WAIT(3, 0); PV(3);
    } else {
      V_WAIT(1, 0); PV(1);
    }
#undef V_READ
#undef V_WAIT
#undef PV

    if constexpr (!LAST) {
      TILE_LANDED();
      __syncthreads();
    }
  };
  using std::integral_constant;
  for (int t = 0; t + 1 < nt; t += 2) {
    tile(t, integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 2>{});
    if (t + 2 < nt) tile(t + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 2>{});
  }
  {
    const int t = nt - 1;
    const bool one_half = t * KV + 32 >= Tn;            // the tile's second 32 keys are all past the end
    if (t & 1) {
      if (one_half) tile(t, integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{});
      else tile(t, integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{});
    } else {
      if (one_half) tile(t, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{});
      else tile(t, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{});
    }
  }

  // combine the two lane halves' partial sums, normalise, store ctx[b*T+q][h*64 + d]
  {
    const unsigned u = __float_as_uint(l_run);
    auto sw2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    l_run = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
  }
  const float inv = 1.0f / l_run;
  const int q = q0 + ql;
  if (q < Tn) {
    T* op = ctx + (row0 + q) * D + h * 64 + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<Half4<T>*>(op + dt * 32 + 8 * g) =
            to_half4<T>(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
  }
}

#undef LOAD_TILE
#undef LOAD_TILE_INSIDE
#undef TILE_LANDED

}  // namespace

extern "C" {

// qkv 16-bit [B*T, ld] with Q (pre-scaled by 64^-0.5 * log2 e) in
// columns [0, D), K in [D, 2D), V in [2D, 3D), all row-major; ctx 16-bit [B*T, D].  No transposed copy of V:
// the kernel reads its V tiles through the LDS transpose read.
int cogaim_mhsa_qkv_fwd(const void* qkv, void* ctx, int B, int T, int H, int ld, int dtype, void* stream) {
  if (B <= 0 || T <= 0 || H <= 0 || (ld & 7) || ld < 3 * 64 * H) return COGAIM_EINVAL;
  const int D = 64 * H;
  dim3 grid(((T + QB - 1) / QB) * B * H);
  if (dtype == 0)
    hipLaunchKernelGGL(mhsa_qkv_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)qkv,
                       (__bf16*)ctx, T, H, ld, D);
  else if (dtype == 1)
    hipLaunchKernelGGL(mhsa_qkv_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)qkv,
                       (_Float16*)ctx, T, H, ld, D);
  else
    return COGAIM_EINVAL;
  return COGAIM_LAUNCH_CHECK();
}

}  // extern "C"
