// 256x256x64 GEMM main loop on FOUR waves of 128x128 (included by gemm.hip, inside its anonymous namespace, after
// gemm_prefetch.h): the whole rounds of tiles of a large problem, like gemm256f_kernel, with half its waves.
//
// Why: a K-tile of the eight-wave kernels moves 256 KB through the CU's LDS port -- 8 waves x (16 KB of A + 8 KB of W
// fragments) read, 64 KB written by the LDS-DMA -- which at 128 bytes per clock is as long as the K-tile's 2048 MFMA
// cycles per SIMD; timing-only variants show the loop at the MFMA rate with either the reads or the DMA left out
// and 1.6x slower with both.  A wave that owns 128x128 reads 16 KB of A and 16 KB of W per K-tile: 128 KB for the
// workgroup, 192 KB with the DMA, three quarters of the MFMA time.  The price is the register file: 256 accumulator
// registers per lane (the AGPR half of a 512-register wave), so one wave per SIMD and nobody to fill its waits.
//
// Same LDS image, same fragment maps, same k order per output element and the same epilogue arithmetic as
// gemm256f_kernel / gemm256p_kernel: bit-identical results.  A K-tile is eight steps of 16 MFMAs, one 64x64 quadrant
// of the wave's tile at one k32 half, ordered as in gemm256f_kernel (a step changes ONE operand and reads the four
// fragments the next step changes between its own MFMAs; 64 fragment registers):
//   step  quadrant (rows, cols)  k32   reads issued meanwhile
//    1    (0-63,   0-63)   0           W cols 64-127, k32 0
//    2    (0-63,   64-127) 0           A rows 64-127, k32 0
//    3    (64-127, 64-127) 0           A rows 64-127, k32 1
//    4    (64-127, 0-63)   0           W cols 0-63, k32 1        then: own reads done | BARRIER (regions Al, We free)
//    5    (64-127, 0-63)   1           W cols 64-127, k32 1
//    6    (64-127, 64-127) 1           A rows 0-63, k32 1
//    7    (0-63,   64-127) 1           -            then: all but the six youngest DMA pieces landed, own reads done | BARRIER
//    8    (0-63,   0-63)   1           A rows 0-63 and W cols 0-63, k32 0, of the NEXT K-tile
// A wave brings 16 of a K-tile's 64 1-KiB pieces, two per step, in the order the regions of the other stage fall free.
#pragma once
#include <type_traits>

template <class T, class Epi, int DIAG>
__global__ __launch_bounds__(256) void gemm256w_kernel(GemmShape g, Epi epi, PersistSched sc) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  using H = Half<T>;
  using v8 = typename H::v8;
  constexpr bool WIDE = Epi::kWide;
  constexpr int CUS = 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K / BK;
  const int first = xcd_remap(blockIdx.x, CUS);
  const int n_items = sc.n_full / CUS;                  // whole rounds only: every workgroup the same count
  const T* const A0 = reinterpret_cast<const T*>(g.A);
  const T* const W0 = reinterpret_cast<const T*>(g.W);

  // ---- DMA side -----------------------------------------------------------------------------------------------
  // A piece is 1 KiB = 8 stage rows.  Each wave brings four pieces of each of the four regions of a stage, named
  // after the fragments read from them: Ae = A rows 0-63 of every wave (row mod 128 < 64), Al = the other A rows,
  // We = W rows of column halves 0-63 (row mod 128 < 64), Wl = the other W rows.  Cursor order of a K-tile's sixteen
  // pieces: Al x 4, We x 4, Ae x 4, Wl x 4 (the order in which the regions of a stage fall free).
  const int l3 = lane >> 3, lc = lane & 7;
  unsigned off[16];                                     // lane offsets (bytes from the tile's first A / W row)
  int pdst[16];                                         // LDS byte offsets inside a stage (wave-uniform)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = 4 * wave + u;                         // 0..15
    const int pe = e < 8 ? e : e + 8, pl = pe + 8;      // piece (of 32) in the low-numbered / high-numbered region
    const int ra_l = pl * 8 + l3, ra_e = pe * 8 + l3;
    off[u] = (unsigned)ra_l * (unsigned)(g.lda * 2) + (unsigned)((lc ^ swz_a(ra_l)) << 4);                       // Al
    off[4 + u] = (unsigned)ra_e * (unsigned)(g.ldw * 2) + (unsigned)((lc ^ (WIDE ? swz_w(ra_e) : swz_a(ra_e))) << 4);   // We
    off[8 + u] = (unsigned)ra_e * (unsigned)(g.lda * 2) + (unsigned)((lc ^ swz_a(ra_e)) << 4);                   // Ae
    off[12 + u] = (unsigned)ra_l * (unsigned)(g.ldw * 2) + (unsigned)((lc ^ (WIDE ? swz_w(ra_l) : swz_a(ra_l))) << 4);  // Wl
    pdst[u] = pl * 1024;
    pdst[4 + u] = OP_BYTES + pe * 1024;
    pdst[8 + u] = pe * 1024;
    pdst[12 + u] = OP_BYTES + pl * 1024;
  }
  auto tile_origin = [&](int item, int& z, int& row0, int& col0) {
    const int tile = item * CUS + first;
    z = tile / sc.tiles_per_z;
    int tm, tn;
    tile_coords(sc, tile - z * sc.tiles_per_z, tm, tn);
    row0 = tm * BM2; col0 = tn * BN2;
  };
  // resource words (base address, bytes) of a tile's A rows and W rows; zero bytes past the last tile, so that the
  // hardware drops the loads of a cursor that has run off the end while they still count in vmcnt
  auto item_rs = [&](int item, unsigned& a_lo, unsigned& a_hi, unsigned& a_n, unsigned& w_lo, unsigned& w_hi,
                     unsigned& w_n) {
    int z = 0, row0 = 0, col0 = 0;
    const bool ok = item < n_items;
    if (ok) tile_origin(item, z, row0, col0);
    const unsigned long long pa = reinterpret_cast<unsigned long long>(A0 + z * g.strideA + (size_t)row0 * g.lda);
    const unsigned long long pw = reinterpret_cast<unsigned long long>(W0 + z * g.strideW + (size_t)col0 * g.ldw);
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
  int d_kt = 0, d_st = 0;                               // cursor: next K-tile to issue (place in its tile), its stage
  auto dma_piece = [&](const int q) {                   // piece q (cursor order) of the cursor's K-tile
    if constexpr (!(DIAG & 1)) {
      const int ko = d_kt * (BK * 2);
      const bool is_a = q < 4 || (q >= 8 && q < 12);
      bufload_lds16(is_a ? rsrc_of(ca_lo, ca_hi, ca_n) : rsrc_of(cw_lo, cw_hi, cw_n), off[q], ko, smem + d_st + pdst[q]);
    }
  };
  auto dma_next = [&]() {                               // the cursor moves to the next K-tile of the stream
    const bool last = ++d_kt == nk;
    d_kt = last ? 0 : d_kt;
    ca_lo = last ? na_lo : ca_lo; ca_hi = last ? na_hi : ca_hi; ca_n = last ? na_n : ca_n;
    cw_lo = last ? nw_lo : cw_lo; cw_hi = last ? nw_hi : cw_hi; cw_n = last ? nw_n : cw_n;
    d_st ^= STAGE_BYTES;
  };

  // ---- math side ------------------------------------------------------------------------------------------------
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = (fq ^ swz_a(fr)) << 4;
  const int wrow = WIDE ? 8 * (fr >> 2) + (fr & 3) : fr;
  constexpr int W_NH_STRIDE = 32 * 128;
  constexpr int W_NJ_STRIDE = WIDE ? 4 * 128 : 16 * 128;
  const int cw0 = (fq ^ (WIDE ? swz_w(wrow) : swz_a(wrow))) << 4;
  const char* const a_rd0 = smem + (wr * 128 + fr) * 128 + c0;                  // + stage + (mh * 4 + i) * 2048
  const char* const a_rd1 = smem + (wr * 128 + fr) * 128 + (c0 ^ 64);
  const char* const w_rd0 = smem + OP_BYTES + (wc * 128 + wrow) * 128 + cw0;    // + stage + nh * W_NH_STRIDE + j * W_NJ_STRIDE
  const char* const w_rd1 = smem + OP_BYTES + (wc * 128 + wrow) * 128 + (cw0 ^ 64);

#define G_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ") lgkmcnt(0)" ::: "memory")
#define G_BARRIER()                                          \
  do {                                                       \
    __builtin_amdgcn_sched_barrier(0);                       \
    if constexpr (!(DIAG & 2)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);                       \
  } while (0)
#define G_LOAD_A(dst_, st_, mh_, s_)                                                                 \
  if constexpr (!(DIAG & 4)) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                        \
    dst_[i_] = *reinterpret_cast<const v8*>(((s_) ? a_rd1 : a_rd0) + (st_) + ((mh_) * 4 + i_) * 2048)
// column half nq_ (64 columns) = fragments (nh, j) = (2 nq_ + (jj >> 1), jj & 1), jj = 0..3
#define G_LOAD_W(dst_, st_, nq_, s_)                                                                 \
  if constexpr (!(DIAG & 4)) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                        \
    dst_[j_] = *reinterpret_cast<const v8*>(((s_) ? w_rd1 : w_rd0) + (st_) + (2 * (nq_) + (j_ >> 1)) * W_NH_STRIDE + (j_ & 1) * W_NJ_STRIDE)
// the emitted order of a step: nd_ times { MFMA, LDS read }, nv_ times { MFMA, pv_ DMA pieces }, the other MFMAs
#define G_MIX(nd_, nv_, pv_)                                                                            \
  do {                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < (nd_); ++k_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
    }                                                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < (nv_); ++k_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
      __builtin_amdgcn_sched_group_barrier(0x010, (pv_), 0);                                         \
    }                                                                                                \
    if ((nd_) + (nv_) < 16) __builtin_amdgcn_sched_group_barrier(0x008, 16 - (nd_) - (nv_), 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#define G_MATH(af_, wf_, mh_, nq_)                                                                   \
  if constexpr (!(DIAG & 16)) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                       \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                   \
    acc[nq_][(mh_) * 4 + i_][j_] = H::mfma16(wf_[j_], af_[i_], acc[nq_][(mh_) * 4 + i_][j_])
#define G_PIN4(v_) asm volatile("" : "+v"((v_).x), "+v"((v_).y), "+v"((v_).z), "+v"((v_).w))

  // acc[h]: the wave's columns 64 h .. 64 h + 63, laid out as the 128x64 wave tile of the eight-wave kernels
  f32x4 acc[2][8][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[h][i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  v8 ax[4], ay[4], wa[4], wb[4];               // two A buffers, two W buffers, one k32 half of a 64-wide quadrant side each
  if constexpr ((DIAG & 4) != 0) {              // never loaded in this variant: defined, and opaque to the optimiser
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_) {
      ax[i_] = v8{}; ay[i_] = v8{}; wa[i_] = v8{}; wb[i_] = v8{};
      asm volatile("" : "+v"(ax[i_]), "+v"(ay[i_]), "+v"(wa[i_]), "+v"(wb[i_]));
    }
  }
  int st_cur = 0;                               // stage of the K-tile being computed; the next one is in the other
  int first_allow = 6;                          // wave-uniform: what the next tile's first barrier may leave in flight: its own 6
                                                // pieces plus the stores the last epilogue is known to have issued behind its last wait

  // ---- prologue: K-tiles 0 and 1 whole (the first K-tile of a tile expects its successor issued) ----------------
#pragma unroll
  for (int q = 0; q < 16; ++q) dma_piece(q);
  dma_next();
#pragma unroll
  for (int q = 0; q < 16; ++q) dma_piece(q);
  dma_next();
  G_WAIT(16);
  G_BARRIER();
  G_LOAD_A(ax, 0, 0, 0);
  G_LOAD_W(wa, 0, 0, 0);

  // One K-tile.  FIRST: first of a tile (the previous tile's last K-tile issued this cursor K-tile's last ten
  // pieces ahead of its epilogue, and the vector-memory operations issued since are the epilogue's).  LAST: last of
  // a tile.  Pieces: steps 5, 6, 7 issue pieces 0-5 of the cursor's K-tile (regions Al and We of this K-tile's
  // stage, free behind the barrier before step 5), step 8 and steps 1-4 of the next K-tile pieces 6-15.
  auto ktile = [&](auto first_c, auto last_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value != 0, LAST = decltype(last_c)::value != 0;
    constexpr int NV = FIRST ? 0 : 2;
    const int st_nxt = st_cur ^ STAGE_BYTES;
    __builtin_amdgcn_sched_barrier(0);
    G_LOAD_W(wb, st_cur, 1, 0);                 // step 1
    if constexpr (!FIRST) { dma_piece(8); dma_piece(9); }
    G_MATH(ax, wa, 0, 0);
    G_MIX(4, NV, 1);
    G_LOAD_A(ay, st_cur, 1, 0);                 // step 2
    if constexpr (!FIRST) { dma_piece(10); dma_piece(11); }
    G_MATH(ax, wb, 0, 1);
    G_MIX(4, NV, 1);
    G_LOAD_A(ax, st_cur, 1, 1);                 // step 3
    if constexpr (!FIRST) { dma_piece(12); dma_piece(13); }
    G_MATH(ay, wb, 1, 1);
    G_MIX(4, NV, 1);
    G_LOAD_W(wb, st_cur, 0, 1);                 // step 4
    if constexpr (!FIRST) { dma_piece(14); dma_piece(15); dma_next(); }
    G_MATH(ay, wa, 1, 0);
    G_MIX(4, NV, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own reads of regions Al and We done
    G_BARRIER();
    G_LOAD_W(wa, st_cur, 1, 1);                 // step 5
    dma_piece(0); dma_piece(1);
    G_MATH(ax, wb, 1, 0);
    G_MIX(4, 2, 1);
    G_LOAD_A(ay, st_cur, 0, 1);                 // step 6
    dma_piece(2); dma_piece(3);
    G_MATH(ax, wa, 1, 1);
    G_MIX(4, 2, 1);
    dma_piece(4); dma_piece(5);                 // step 7
    G_MATH(ay, wa, 0, 1);
    G_MIX(0, 2, 1);
    if (FIRST && first_allow == 38) G_WAIT(38);              // all but pieces 0-5 just issued (and the epilogue's stores)
    else G_WAIT(6);
    G_BARRIER();
    G_LOAD_A(ax, st_nxt, 0, 0);                 // step 8
    G_LOAD_W(wa, st_nxt, 0, 0);
    dma_piece(6); dma_piece(7);
    if constexpr (LAST) {
#pragma unroll
      for (int q = 8; q < 16; ++q) dma_piece(q);
      dma_next();
    }
    G_MATH(ay, wb, 0, 0);
    G_MIX(8, LAST ? 5 : 2, LAST ? 2 : 1);
    st_cur = st_nxt;
  };
  using std::integral_constant;
#define G_KT(first_, last_) ktile(integral_constant<int, first_>{}, integral_constant<int, last_>{})

  float peak = 0.f;           // fp16 range guard of the 16-bit epilogues (see track_peak)
  for (int j = 0; j < n_items; ++j) {
    if (j) item_rs(j + 1, na_lo, na_hi, na_n, nw_lo, nw_hi, nw_n);   // the cursor is inside tile j until its last K-tiles
    G_KT(1, 0);
    for (int kt = 2; kt < nk; ++kt) G_KT(0, 0);
    G_KT(0, 1);

    // ---- epilogue (that of gemm256f_kernel, once per 64-column half of the wave's tile) ---------------------------
    int z, row0, col0;
    tile_origin(j, z, row0, col0);
    first_allow = 6;
    if constexpr ((DIAG & 8) != 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) asm volatile("" ::"v"(acc[h][i][jj]));
    } else {
      const int m_base = row0 + wr * 128;
      float4 cbw[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n_base = col0 + wc * 128 + 64 * h;
        if constexpr (Epi::kReduce == 1) {
          epi.reduce_tile(z, m_base, n_base, (col0 / BN2) * 4 + wc * 2 + h, acc[h], fr, fq);
        } else if constexpr (Epi::kReduce == 2) {
          epi.reduce_tile(z, m_base, n_base, (row0 / BM2) * 2 + wr, acc[h], fr, fq);
        } else if constexpr (WIDE) {
          if (h == 0) {     // the column vectors of both halves ahead of the first store: a later load's wait would also wait for the stores
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float4 cl;
                epi.load_cols(n_base + 64 * hh + 8 * fq + 4 * (q & 1) + 32 * (q >> 1), cbw[hh][q], cl);
              }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) { G_PIN4(cbw[hh][0]); G_PIN4(cbw[hh][1]); G_PIN4(cbw[hh][2]); G_PIN4(cbw[hh][3]); }
          }
          if (row0 + BM2 <= g.M && col0 + BN2 <= g.N && epi.lines_ok()) {   // wave-uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) epi.finish8x2n_lines(z, m_base + 16 * i + fr, n_base + 8 * fq, acc[h][i], cbw[h], peak);
            if (h == 1) first_allow = 38;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) epi.finish8x2n(z, m_base + 16 * i + fr, n_base + 8 * fq, acc[h][i], cbw[h], peak);
          }
        } else {
          float4 cb[4], cl[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) epi.load_cols(n_base + 16 * jj + 4 * fq, cb[jj], cl[jj]);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            if constexpr (Epi::kColVecs >= 1) G_PIN4(cb[jj]);
            if constexpr (Epi::kColVecs >= 2) G_PIN4(cl[jj]);
          }
#pragma unroll
          for (int part = 0; part < 8; ++part) {        // 4 residual loads in flight per lane, then their 4 stores
            float4 xr[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              xr[jj] = Epi::kResidual ? epi.load_res(z, m_base + 16 * part + fr, n_base + 16 * jj + 4 * fq) : cb[jj];
            if constexpr (Epi::kResidual) {
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) G_PIN4(xr[jj]);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              epi.finish4n(z, m_base + 16 * part + fr, n_base + 16 * jj + 4 * fq, acc[h][part][jj], cb[jj], cl[jj], xr[jj], peak);
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[h][i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  G_WAIT(0);                                    // dropped loads past the last tile included
  epi.flush_peak(peak);
#undef G_KT
#undef G_PIN4
#undef G_MATH
#undef G_MIX
#undef G_LOAD_W
#undef G_LOAD_A
#undef G_BARRIER
#undef G_WAIT
}

template <class T, class Epi>
void launch_wide(const GemmShape& g, const Epi& epi, const PersistSched& sc, hipStream_t st, int diag) {
#ifdef COGAIM_TOOLS
#define G_DIAG_CASE(d_) \
  case d_: hipLaunchKernelGGL((gemm256w_kernel<T, Epi, d_>), dim3(256), dim3(256), 0, st, g, epi, sc); break;
  switch (diag) {
    G_DIAG_CASE(8) G_DIAG_CASE(9) G_DIAG_CASE(12) G_DIAG_CASE(24) G_DIAG_CASE(13)
    default: hipLaunchKernelGGL((gemm256w_kernel<T, Epi, 0>), dim3(256), dim3(256), 0, st, g, epi, sc);
  }
#undef G_DIAG_CASE
#else
  (void)diag;
  hipLaunchKernelGGL((gemm256w_kernel<T, Epi, 0>), dim3(256), dim3(256), 0, st, g, epi, sc);
#endif
}
