// Persistent 128x256x64 GEMM for the epilogues that store 16-bit outputs (included by gemm.hip, inside its
// anonymous namespace, after gemm_persistent.h).
//
// The 256x256 ping-pong kernel keeps a 256x256 fp32 tile in registers (128 per lane), which leaves room neither
// for a second set of operand fragments nor for a finished tile: LDS reads and MFMAs alternate in barrier-delimited
// intervals, and after the last K-tile the matrix pipe idles while 128 KB per CU go through the store path
// (~85 cycles per 1-KB store instruction, a third of a tile's time at K = 768).  Here a workgroup's tile is 128x256,
// a wave's 64x64 (64 accumulator registers), and the freed registers buy
//   * operand fragments read from LDS one phase AHEAD of the MFMAs that use them (second fragment set): no
//     load/math alternation, one workgroup barrier per K-tile instead of eight;
//   * a finished tile parked as 32 registers of packed 16-bit values whose eight 16-byte stores are issued one per
//     K-tile under the MFMAs of the NEXT tile.
// Price: the W panel is staged once per 128 rows instead of once per 256 (1.5x the LDS-DMA bytes per flop) and a
// wave reads 16 KB of fragments per 32 MFMAs instead of 24 KB per 64.
//
// K-tiles of consecutive tiles form one stream through a ring of three 48-KB LDS stages (A 128x64, W 256x64,
// swizzled as in the 256x256 kernel).  Per K-tile f a wave runs
//   phase a: read W fragments (columns 32-63) of f | DMA: last 3 pieces of K-tile f+2 | 16 MFMAs (columns 0-31)
//            wait: own pieces of K-tile f+1 landed, own LDS reads of f done | workgroup barrier
//   phase b: read A and W (columns 0-31) fragments of f+1 | DMA: first 3 pieces of K-tile f+3 into the stage of f
//            | 16 MFMAs (columns 32-63) | at most one parked store (or the tile's bias row) behind them
// Vector-memory operations retire in order, so "all but the newest N" is the only wait there is: N = 6 (the pieces
// issued since those of f+1) plus the one store or bias piece issued since, when there was one.  A wait that names
// fewer than were issued is only slower; one that names more reads a stage before it has landed.
// Results are bit-identical to the 256x256 kernel's: the same k order per output element and the same epilogue
// arithmetic.
#pragma once
#include <type_traits>

constexpr int S_BM = 128, S_BN = 256;
constexpr int S_A_BYTES = S_BM * BK * 2;          // 16 KiB
constexpr int S_W_BYTES = S_BN * BK * 2;          // 32 KiB
constexpr int S_STAGE = S_A_BYTES + S_W_BYTES;    // 48 KiB
constexpr int S_RING = 3 * S_STAGE;               // 144 KiB
constexpr int S_LDS = S_RING + 2 * 1024;          // + two 1-KiB rows of column constants (tile parity)
constexpr int S_MIN_NK = 12;                      // K-tiles 0..7 carry the parked stores, K-tile 8 the bias row

struct StreamSched { int tiles_m, tiles_n, band, n_tiles, nwg; };

// DIAG (timing-only variants of tools builds, wrong results): 1 no DMA, 2 no workgroup barriers, 4 no LDS fragment
// reads, 8 no epilogue arithmetic and no stores, 16 no MFMAs.
template <class T, class Epi, int DIAG>
__global__ __launch_bounds__(512, 2) void gemm128s_kernel(GemmShape g, Epi epi, StreamSched sc) {
  __shared__ __attribute__((aligned(16))) char smem[S_LDS];
  using H = Half<T>;
  using v8 = typename H::v8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K / BK;
  const int first = xcd_remap(blockIdx.x, sc.nwg);
  if (first >= sc.n_tiles) return;
  const int n_items = (sc.n_tiles - first + sc.nwg - 1) / sc.nwg;
  const T* const A0 = reinterpret_cast<const T*>(g.A);
  const T* const W0 = reinterpret_cast<const T*>(g.W);
  PersistSched ps{};
  ps.tiles_m = sc.tiles_m; ps.tiles_n = sc.tiles_n; ps.band = sc.band;

  // ---- DMA side ------------------------------------------------------------------------------------------
  // A piece is 1 KiB = 8 rows of a stage; wave w brings A pieces 2w, 2w+1 and W pieces 4w .. 4w+3 of every K-tile.
  // Addressing is scalar: a buffer resource per operand that starts at the tile's first row (rows past the matrix
  // fall outside it and read as zero), a fixed 32-bit lane offset per piece, the K-tile's byte offset as soffset.
  const int l3 = lane >> 3, lc = lane & 7;
  unsigned offA[2], offW[4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (2 * wave + u) * 8 + l3;
    offA[u] = (unsigned)r * (unsigned)(g.lda * 2) + (unsigned)((lc ^ swz_a(r)) << 4);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = (4 * wave + u) * 8 + l3;
    offW[u] = (unsigned)r * (unsigned)(g.ldw * 2) + (unsigned)((lc ^ swz_w(r)) << 4);
  }
  // The cursor (next K-tile to issue: its place in its tile, its tile, its stage) never branches: past the last
  // tile the resources have zero records, so the hardware drops the loads while they still count in vmcnt, and the
  // resource words of the cursor's tile and of the tile after it are wave-uniform scalars switched by a select
  // when the cursor leaves a tile (the words of the tile after are recomputed once per tile by the math side).
  int d_kt = 0, d_st = 0;
  auto tile_origin = [&](int item, int& row0, int& col0) {
    int tm, tn;
    tile_coords(ps, item * sc.nwg + first, tm, tn);
    row0 = tm * S_BM; col0 = tn * S_BN;
  };
  // resource words (base address, bytes) of a tile's A rows and W rows; zero bytes past the last tile
  auto item_rs = [&](int item, unsigned& a_lo, unsigned& a_hi, unsigned& a_n, unsigned& w_lo, unsigned& w_hi,
                     unsigned& w_n) {
    int row0 = 0, col0 = 0;
    const bool ok = item < n_items;
    if (ok) tile_origin(item, row0, col0);
    const unsigned long long pa = reinterpret_cast<unsigned long long>(A0 + (size_t)row0 * g.lda);
    const unsigned long long pw = reinterpret_cast<unsigned long long>(W0 + (size_t)col0 * g.ldw);
    a_lo = __builtin_amdgcn_readfirstlane((unsigned)pa); a_hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
    w_lo = __builtin_amdgcn_readfirstlane((unsigned)pw); w_hi = __builtin_amdgcn_readfirstlane((unsigned)(pw >> 32));
    a_n = __builtin_amdgcn_readfirstlane(ok ? (unsigned)(g.M - row0) * (unsigned)(g.lda * 2) : 0u);
    w_n = __builtin_amdgcn_readfirstlane(ok ? (unsigned)(g.N - col0) * (unsigned)(g.ldw * 2) : 0u);
  };
  auto rsrc_of = [&](unsigned lo, unsigned hi, unsigned n) {
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)n, 0x00020000);
  };
  unsigned ca_lo, ca_hi, ca_n, cw_lo, cw_hi, cw_n;      // the cursor's tile
  unsigned na_lo, na_hi, na_n, nw_lo, nw_hi, nw_n;      // the tile after it
  item_rs(0, ca_lo, ca_hi, ca_n, cw_lo, cw_hi, cw_n);
  item_rs(1, na_lo, na_hi, na_n, nw_lo, nw_hi, nw_n);
  auto dma_first = [&]() {                              // A pieces and the first W piece of the cursor's K-tile
    char* dst = smem + d_st;
    const int ko = d_kt * (BK * 2);
    const __amdgpu_buffer_rsrc_t rA = rsrc_of(ca_lo, ca_hi, ca_n), rW = rsrc_of(cw_lo, cw_hi, cw_n);
    if constexpr (!(DIAG & 1)) {
      bufload_lds16(rA, offA[0], ko, dst + (2 * wave) * 1024);
      bufload_lds16(rA, offA[1], ko, dst + (2 * wave + 1) * 1024);
      bufload_lds16(rW, offW[0], ko, dst + S_A_BYTES + (4 * wave) * 1024);
    }
  };
  auto dma_second = [&]() {                             // the other three W pieces; the cursor moves on
    char* dst = smem + d_st + S_A_BYTES;
    const int ko = d_kt * (BK * 2);
    const __amdgpu_buffer_rsrc_t rW = rsrc_of(cw_lo, cw_hi, cw_n);
    if constexpr (!(DIAG & 1)) {
      bufload_lds16(rW, offW[1], ko, dst + (4 * wave + 1) * 1024);
      bufload_lds16(rW, offW[2], ko, dst + (4 * wave + 2) * 1024);
      bufload_lds16(rW, offW[3], ko, dst + (4 * wave + 3) * 1024);
    }
    const bool last = ++d_kt == nk;
    d_kt = last ? 0 : d_kt;
    ca_lo = last ? na_lo : ca_lo; ca_hi = last ? na_hi : ca_hi; ca_n = last ? na_n : ca_n;
    cw_lo = last ? nw_lo : cw_lo; cw_hi = last ? nw_hi : cw_hi; cw_n = last ? nw_n : cw_n;
    d_st = d_st == 2 * S_STAGE ? 0 : d_st + S_STAGE;
  };

  // ---- math side -----------------------------------------------------------------------------------------
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = (fq ^ swz_a(fr)) << 4;                                 // A k-chunk of k32 step 0; step 1 is c0 ^ 64
  const int wrow = 8 * (fr >> 2) + (fr & 3);                            // interleaved W rows (see gemm_persistent.h)
  const int cw0 = (fq ^ swz_w(wrow)) << 4;
  const char* const a_rd0 = smem + (wr * 64 + fr) * 128 + c0;           // + stage + i * 2048
  const char* const a_rd1 = smem + (wr * 64 + fr) * 128 + (c0 ^ 64);
  const char* const w_rd0 = smem + S_A_BYTES + (wc * 64 + wrow) * 128 + cw0;   // + stage + nh * 4096 + j * 512
  const char* const w_rd1 = smem + S_A_BYTES + (wc * 64 + wrow) * 128 + (cw0 ^ 64);

#define S_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ") lgkmcnt(0)" ::: "memory")
#define S_BARRIER()                                \
  do {                                             \
    __builtin_amdgcn_sched_barrier(0);             \
    if constexpr (!(DIAG & 2)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);             \
  } while (0)
#define S_LOAD_A(dst_, st_)                                                                          \
  if constexpr (!(DIAG & 4)) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                 \
    dst_[0][i_] = *reinterpret_cast<const v8*>(a_rd0 + (st_) + i_ * 2048);                           \
    dst_[1][i_] = *reinterpret_cast<const v8*>(a_rd1 + (st_) + i_ * 2048);                           \
  }
#define S_LOAD_W(dst_, st_, nh_)                                                                     \
  if constexpr (!(DIAG & 4)) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                 \
    dst_[0][j_] = *reinterpret_cast<const v8*>(w_rd0 + (st_) + (nh_) * 4096 + j_ * 512);             \
    dst_[1][j_] = *reinterpret_cast<const v8*>(w_rd1 + (st_) + (nh_) * 4096 + j_ * 512);             \
  }
// n_ times { m_ MFMAs, then one instruction of class mask_ } in the emitted order (mask_ 0: MFMAs only)
#define S_MIX(mask_, n_, m_)                                                                         \
  _Pragma("unroll") for (int k_ = 0; k_ < ((mask_) ? (n_) : 1); ++k_) {                              \
    __builtin_amdgcn_sched_group_barrier(0x008, (m_), 0);                                            \
    if (mask_) __builtin_amdgcn_sched_group_barrier((mask_), 1, 0);                                  \
  }
#define S_MATH(af_, wf_, nh_)                                                                        \
  if constexpr (!(DIAG & 16)) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                   \
    acc[i_][(nh_) * 2 + j_] = H::mfma16(wf_[s_][j_], af_[s_][i_], acc[i_][(nh_) * 2 + j_])

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  v8 af[2][2][4], wf0[2][2], wf1[2][2];       // A fragments of this and the next K-tile, W fragments per column half
  if constexpr ((DIAG & 4) != 0) {             // never loaded in this variant: defined, and opaque to the optimiser
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
      for (int i_ = 0; i_ < 4; ++i_) {
        af[0][s_][i_] = v8{}; af[1][s_][i_] = v8{};
        asm volatile("" : "+v"(af[0][s_][i_]), "+v"(af[1][s_][i_]));
      }
#pragma unroll
      for (int j_ = 0; j_ < 2; ++j_) {
        wf0[s_][j_] = v8{}; wf1[s_][j_] = v8{};
        asm volatile("" : "+v"(wf0[s_][j_]), "+v"(wf1[s_][j_]));
      }
    }
  }
  u32x4 pk[8];                                 // the parked tile: two 16-byte payloads per 16-row fragment
  T* p_first = nullptr; T* p_second = nullptr; // where pk[0] and pk[1] go; fragment i is 16 i rows further down
  bool pending = false;                        // wave-uniform: pk holds a tile whose stores are still to be issued
  float peak = 0.f;                            // fp16 range guard (see track_peak)
  int st_cur = 0, st_nxt = S_STAGE;            // stages of K-tile f and f + 1
  __amdgpu_buffer_rsrc_t rB;                   // bias row of the tile being computed

  // ---- prologue: K-tiles 0 and 1 whole, the first half of K-tile 2 ------------------------------------------
  dma_first(); dma_second();
  dma_first(); dma_second();
  dma_first();
  S_WAIT(9);                                   // K-tile 0 (15 pieces issued)
  S_BARRIER();
  S_LOAD_A(af[0], 0);
  S_LOAD_W(wf0, 0, 0);

  // One K-tile.  PAR: which fragment set holds its A fragments.  SLOT: what phase b issues behind its MFMAs
  // (-1 nothing, 0..7 that parked store, 8 the bias row).  PREV: what the previous K-tile's phase b issued
  // (0 nothing, 1 a parked store if there was a parked tile, 2 the bias row).
  auto ktile = [&](auto par_c, auto slot_c, auto prev_c, const int item) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value, SLOT = decltype(slot_c)::value, PREV = decltype(prev_c)::value;
    // ---- phase a
    S_LOAD_W(wf1, st_cur, 1);
    dma_second();
    S_MATH(af[PAR], wf0, 0);
    // one scheduling region: the reads and the DMA go between the MFMAs, not in front of them
    S_MIX(0x100, 4, 1); S_MIX(0x010, 3, 2); S_MIX(0x000, 0, 6);
    __builtin_amdgcn_sched_barrier(0);
    if (PREV == 2 || (PREV == 1 && pending)) S_WAIT(7); else S_WAIT(6);
    S_BARRIER();
    // ---- phase b
    S_LOAD_A(af[PAR ^ 1], st_nxt);
    S_LOAD_W(wf0, st_nxt, 0);
    dma_first();
    S_MATH(af[PAR], wf1, 1);
    S_MIX(0x100, 12, 1); S_MIX(0x010, 3, 1); S_MIX(0x000, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SLOT >= 0 && SLOT < 8) {
      if (pending) {
        T* p = ((SLOT & 1) ? p_second : p_first) + (size_t)(SLOT >> 1) * 16 * epi.ldc;
        __builtin_nontemporal_store(pk[SLOT], reinterpret_cast<u32x4*>(p));
      }
    } else if constexpr (SLOT == 8) {
      bufload_lds16(rB, lane * 16, 0, smem + S_RING + (item & 1) * 1024);
    }
    st_cur = st_nxt;
    st_nxt = st_nxt == 2 * S_STAGE ? 0 : st_nxt + S_STAGE;
  };
  using std::integral_constant;
#define S_KT(par_, slot_, prev_) \
  ktile(integral_constant<int, par_>{}, integral_constant<int, slot_>{}, integral_constant<int, prev_>{}, j)

  for (int j = 0; j < n_items; ++j) {
    int row0, col0;
    tile_origin(j, row0, col0);
    rB = make_rsrc(epi.bias + col0, (unsigned)(g.N - col0) * 4u);
    if (j) item_rs(j + 1, na_lo, na_hi, na_n, nw_lo, nw_hi, nw_n);   // the cursor is inside tile j until this tile's last K-tiles
    S_KT(0, 0, 0); S_KT(1, 1, 1); S_KT(0, 2, 1); S_KT(1, 3, 1); S_KT(0, 4, 1); S_KT(1, 5, 1);
    S_KT(0, 6, 1); S_KT(1, 7, 1); S_KT(0, 8, 1); S_KT(1, -1, 2);
    for (int kt = 10; kt < nk; kt += 2) { S_KT(0, -1, 0); S_KT(1, -1, 0); }

    // ---- the tile is complete: bias (+ GELU), 16-bit, parked for the next tile's K-tiles ----------------------
    const int m_base = row0 + wr * 64, n_base = col0 + wc * 64;
    const float4* cbl = reinterpret_cast<const float4*>(smem + S_RING + (j & 1) * 1024) + wc * 16 + 2 * fq;
    float4 cb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cb[q] = cbl[(q & 1) + 8 * (q >> 1)];
    T* row = epi.out + (size_t)(m_base + fr) * epi.ldc + n_base + 8 * fq;
    if constexpr ((DIAG & 8) != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) asm volatile("" ::"v"(acc[i][jj]));
      pending = false;
    } else if (row0 + S_BM <= g.M && col0 + S_BN <= g.N && epi.lines_ok()) {     // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i) epi.pack8x2n_lines(n_base + 8 * fq, acc[i], cb, peak, pk[2 * i], pk[2 * i + 1]);
      line_targets<T>(row, epi.ldc, p_first, p_second);
      pending = true;
    } else {                                                             // edge tiles store at once, guarded per lane
#pragma unroll
      for (int i = 0; i < 4; ++i) epi.finish8x2n(0, m_base + 16 * i + fr, n_base + 8 * fq, acc[i], cb, peak);
      pending = false;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (pending && !(DIAG & 8)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      T* p = ((u & 1) ? p_second : p_first) + (size_t)(u >> 1) * 16 * epi.ldc;
      __builtin_nontemporal_store(pk[u], reinterpret_cast<u32x4*>(p));
    }
  }
  S_WAIT(0);                                   // dropped loads past the last tile included
  epi.flush_peak(peak);
#undef S_KT
#undef S_MATH
#undef S_MIX
#undef S_LOAD_W
#undef S_LOAD_A
#undef S_BARRIER
#undef S_WAIT
}

// Whether a problem can run on gemm128s_kernel: 16-bit-output epilogue, no batch, K of at least S_MIN_NK K-tiles
// and an even number of them, operands addressable with 32-bit byte offsets, enough tiles to fill the chip.
template <class Epi>
bool stream_eligible(const GemmShape& g, int batch) {
  if (!Epi::kWide || batch != 1) return false;
  const int nk = g.K / BK;
  if (nk < S_MIN_NK || (nk & 1)) return false;
  if ((long long)g.M * g.lda * 2 >= (1LL << 31) || (long long)g.N * g.ldw * 2 >= (1LL << 31)) return false;
  const long long tiles = (long long)((g.M + S_BM - 1) / S_BM) * ((g.N + S_BN - 1) / S_BN);
  return tiles >= 256 && tiles < (1 << 30);
}

template <class T, class Epi>
int launch_stream(const GemmShape& g, const Epi& epi, hipStream_t st, int diag = 0) {
  constexpr int CUS = 256;
  StreamSched sc;
  sc.tiles_m = (g.M + S_BM - 1) / S_BM;
  sc.tiles_n = (g.N + S_BN - 1) / S_BN;
  sc.band = (sc.tiles_n % 4 == 0 && sc.tiles_n >= 8) ? 8 : 1;
  sc.n_tiles = sc.tiles_m * sc.tiles_n;
  sc.nwg = sc.n_tiles < CUS ? sc.n_tiles : CUS;
#ifdef COGAIM_TOOLS
#define S_DIAG_CASE(d_) \
  case d_: hipLaunchKernelGGL((gemm128s_kernel<T, Epi, d_>), dim3(sc.nwg), dim3(512), 0, st, g, epi, sc); break;
  switch (diag) {
    S_DIAG_CASE(1) S_DIAG_CASE(2) S_DIAG_CASE(4) S_DIAG_CASE(8) S_DIAG_CASE(16) S_DIAG_CASE(5) S_DIAG_CASE(7) S_DIAG_CASE(3)
    default: hipLaunchKernelGGL((gemm128s_kernel<T, Epi, 0>), dim3(sc.nwg), dim3(512), 0, st, g, epi, sc);
  }
#undef S_DIAG_CASE
#else
  (void)diag;
  hipLaunchKernelGGL((gemm128s_kernel<T, Epi, 0>), dim3(sc.nwg), dim3(512), 0, st, g, epi, sc);
#endif
  return COGAIM_LAUNCH_CHECK();
}
