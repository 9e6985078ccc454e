// Persistent 256x256x64 ping-pong GEMM kernel (included by gemm.hip, inside its anonymous namespace).
//
// One 512-thread workgroup per CU walks a list of work items: whole 256x256 output tiles
// first, then (optionally) one K-slice of a tile of the partial last round.  The K-tiles of
// consecutive items form ONE stream: the LDS-DMA of the next K-tile is always in flight, also
// across an item boundary, so only the very first K-tile of a workgroup pays the load latency
// and the epilogue stores of one tile drain behind the math of the next.
//
// Per K-tile (64 deep) a wave runs four phases = four 64x32 quadrants of its 128x64 output,
// each phase two barrier-delimited intervals:
//     LOAD : ds_read this quadrant's fragments, issue 2 DMA pieces of the NEXT K-tile,
//            wait for the LDS reads and for all but the 4 newest DMA pieces, barrier
//     MATH : 16 x v_mfma_f32_16x16x32, barrier
// Waves 4-7 run one interval behind waves 0-3 (one extra barrier at entry); each SIMD hosts
// one wave of either group, so LDS/DMA traffic of one group overlaps the MFMAs of the other.
//
// DMA order within a K-tile, 2 pieces (1 KiB each) per wave per phase:
//   phase 0: A rows first needed by phase 0 ("A-early", row mod 128 < 64)
//   phase 1: W rows of phases 0 and 3      ("W-early", row mod 64 < 32)
//   phase 2: W rows of phases 1 and 2      ("W-late")
//   phase 3: A rows of phases 2 and 3      ("A-late")
// A piece is read two phases after the wait that retired it at the earliest, always behind a
// barrier all waves passed after their own wait:
//   wait at the end of LOAD 3 (all but W-late, A-late)   -> next tile's phase 0 reads A-early, W-early
//   wait at the end of LOAD 0 (all but A-late, new A-early) -> phase 1 reads W-late
//   wait at the end of LOAD 1                             -> phase 2 reads A-late
// and a stage is overwritten only by DMA issued after the barrier that follows its last read
// (LDS reads are waited for before that barrier).
#pragma once

// LDS chunk swizzles (row -> XOR on the 16-B chunk index, 0..7).  A fragments read rows fr, fr+16, ...;
// interleaved W fragments read rows 8*(fr>>2) + (fr&3) + 4*(ni&1) + 32*(ni>>1) (so that a lane ends up
// owning two runs of 8 adjacent output columns), which needs a different function to stay bank-conflict
// free; it ignores the two row bits that ni moves.
__device__ __forceinline__ int swz_a(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

struct PersistSched {
  int tiles_m;        // row tiles of one problem
  int tiles_n;        // column tiles of one problem
  int band;           // tiles are numbered band by band (band = `band` row tiles), rows fastest inside a band
                      // when band > 1, so that the 32 consecutive tiles an XCD works on at a time form a
                      // band x (32 / band) block: fewer distinct A / W panels per L2 than a 1 x N strip
  int tiles_per_z;    // tiles of one problem (batched GEMM: tile index = z * tiles_per_z + local tile)
  int n_full;         // tiles [0, n_full) are processed whole by workgroups [0, nblk_full), n_full % nblk_full == 0
  int nblk_full;
  int n_split;        // tiles [n_full, n_full + n_split) are cut into `slices` K-slices (1 = whole tile) ...
  int slices;
  int slice_block0;   // ... handled by workgroups [slice_block0, slice_block0 + n_split*slices), after their whole tiles
  float* ws;          // [split tile][slice][32 fragments][512 threads] float4 (slices >= 2 only)
  int tail_items;     // gemm256f_kernel only: workgroups [0, tail_items) run one item (tile, K-slice, quadrant) of the
                      // partial last round behind their whole tiles (gemm_tail.h), 0: the partial round is launched apart
  int quads;          // 1: tiles [n_full, n_full + n_split) are cut into four QUADRANT items instead (one of the
                      // four 64x32 quadrants of every wave's 128x64, i.e. a quarter of the tile's outputs over
                      // the whole K): no partial sums, no fix-up launch, results bit-identical to a whole tile
};

__device__ __forceinline__ void tile_coords(const PersistSched& sc, int lt, int& tm, int& tn) {
  if (sc.band <= 1) { tm = lt / sc.tiles_n; tn = lt - tm * sc.tiles_n; return; }
  const int per_band = sc.band * sc.tiles_n;
  const int b = lt / per_band, r = lt - b * per_band;
  const int m0 = b * sc.band, h = min(sc.band, sc.tiles_m - m0);
  tn = r / h; tm = m0 + (r - tn * h);
}

template <class T, class Epi, bool SWAP, int DIAG>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmShape g, Epi epi, PersistSched sc) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  using H = Half<T>;
  using v8 = typename H::v8;
  // WIDE: W fragment rows are interleaved so that a lane owns output columns [8 fq, 8 fq + 8) and
  // [32 + 8 fq, 32 + 8 fq + 8) of its wave's 64 (16-bit outputs: two 16-B stores per row, each of which
  // the four lanes of a row make 64 contiguous bytes).  fp32 outputs keep 4 columns per lane and
  // fragment, which makes every store instruction cover 64 contiguous bytes per row as well.
  constexpr bool WIDE = SWAP && Epi::kWide;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K / BK;
  const int bid = blockIdx.x;
  const int full_per_block = bid < sc.nblk_full ? sc.n_full / sc.nblk_full : 0;
  const int sb = bid - sc.slice_block0;
  const bool has_slice = !sc.quads && sb >= 0 && sb < sc.n_split * sc.slices;
  const int n_items = full_per_block + (has_slice ? 1 : 0);
  if (n_items == 0 && !sc.quads) return;       // (a launch of the partial last round alone: nblk_full == 0)
#ifdef COGAIM_STAMPS
  // problem class of the diagnostic clock stamps: 0 = N 3072 (fc1), 1 = N 2304 (QKV), 2 = K 3072 (fc2), 3 = the rest
  const int cclass = g.N == 3072 ? 0 : (g.N == 2304 ? 1 : (g.K == 3072 ? 2 : 3));
  if (tid == 0) {
    g_clock_buf[(cclass * 256 + bid) * 4 + 0] = __builtin_amdgcn_s_memtime();
    g_clock_buf[(cclass * 256 + bid) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const T* const A0 = reinterpret_cast<const T*>(g.A);
  const T* const W0 = reinterpret_cast<const T*>(g.W);

  // ---- work items ----------------------------------------------------------------------------
  struct Item { int tile, kt0, kt1, slice; };
  auto item_at = [&](int j) -> Item {
    if (j < full_per_block) return Item{j * sc.nblk_full + xcd_remap(bid, sc.nblk_full), 0, nk, -1};
    const int t = sc.n_full + sb / sc.slices, s = sb % sc.slices;
    return Item{t, (int)((long long)nk * s / sc.slices), (int)((long long)nk * (s + 1) / sc.slices),
                sc.slices == 1 ? -1 : s};
  };

  // ---- DMA side: a cursor over the flat K-tile stream -----------------------------------------
  // piece ids per wave: A-early / A-late / W-early / W-late, two each
  int pAe[2], pAl[2], pWe[2], pWl[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = 2 * wave + u;                         // 0..15
    pAe[u] = e < 8 ? e : e + 8;                         // rows (p*8) mod 128 < 64
    pAl[u] = pAe[u] + 8;
    pWe[u] = (e >> 2) * 8 + (e & 3);                    // rows (p*8) mod 64 < 32
    pWl[u] = pWe[u] + 4;
  }
  const T* sAe[2]; const T* sAl[2]; const T* sWe[2]; const T* sWl[2];
  int d_item = 0, d_kt = 0, d_kt1 = 0;                  // DMA cursor
  bool d_valid = true;
  auto dma_set_tile = [&](int tile, int lane) {        // `lane`: see the quadrant items for why it is a parameter
    const int z = tile / sc.tiles_per_z, lt = tile - z * sc.tiles_per_z;
    int tm, tn; tile_coords(sc, lt, tm, tn);
    const T* A = A0 + z * g.strideA;
    const T* W = W0 + z * g.strideW;
    auto src = [&](const T* base, int ld, int row0, int limit, int piece, bool w_op) {
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((w_op && WIDE) ? swz_w(r) : swz_a(r));
      int gr = row0 + r; gr = gr < limit ? gr : limit - 1;
      return base + (size_t)gr * ld + c * 8;
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      sAe[u] = src(A, g.lda, tm * BM2, g.M, pAe[u], false);
      sAl[u] = src(A, g.lda, tm * BM2, g.M, pAl[u], false);
      sWe[u] = src(W, g.ldw, tn * BN2, g.N, pWe[u], true);
      sWl[u] = src(W, g.ldw, tn * BN2, g.N, pWl[u], true);
    }
  };
  auto dma_set_item = [&](int j) {
    const Item it = item_at(j);
    d_kt = it.kt0; d_kt1 = it.kt1;
    dma_set_tile(it.tile, lane);
  };
  auto dma_advance = [&]() {        // after the 4 phases of one K-tile have issued its pieces
    if (++d_kt >= d_kt1) {
      if (++d_item < n_items) dma_set_item(d_item); else d_valid = false;
    }
  };
#define DMA2(srcs_, pieces_, op_off_, stage_)                                                        \
  do {                                                                                               \
    glds16(srcs_[0] + (size_t)d_kt * BK, smem + (stage_) * STAGE_BYTES + (op_off_) + pieces_[0] * 1024); \
    glds16(srcs_[1] + (size_t)d_kt * BK, smem + (stage_) * STAGE_BYTES + (op_off_) + pieces_[1] * 1024); \
  } while (0)

  // ---- math side --------------------------------------------------------------------------------
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = (fq ^ swz_a(fr)) << 4;                               // A k-chunk of k32 step 0; step 1 is c0 ^ 64
  const int wrow = WIDE ? 8 * (fr >> 2) + (fr & 3) : fr;                 // W row of this lane within the wave's 64
  constexpr int W_NH_STRIDE = 32 * 128;                                  // bytes between the rows of fragments ni, ni+2
  constexpr int W_NJ_STRIDE = WIDE ? 4 * 128 : 16 * 128;                 // ... and of fragments ni, ni+1
  const int cw0 = (fq ^ (WIDE ? swz_w(wrow) : swz_a(wrow))) << 4;
  const char* const a_base = smem + (wr * 128 + fr) * 128;               // + mi*2048
  const char* const w_base = smem + OP_BYTES + (wc * 64 + wrow) * 128;   // + ni*W_NI_STRIDE

  // Re-defines a loaded float4 through an empty asm.  With an LDS-DMA in flight the compiler turns every
  // wait for an ordinary load into vmcnt(0), and it repeats that wait in every later basic block that
  // uses the value: behind a store that makes the block wait for the store's write acknowledgement.  One
  // wait here, after which the values no longer count as loaded, keeps the epilogue's stores streaming.
#define PIN_LOADED4(v_) asm volatile("" : "+v"((v_).x), "+v"((v_).y), "+v"((v_).z), "+v"((v_).w))
#define WAIT_LOAD(n_) asm volatile("s_waitcnt vmcnt(" #n_ ") lgkmcnt(0)" ::: "memory")
#define BARRIER()                                  \
  do {                                             \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    __builtin_amdgcn_sched_barrier(0);             \
  } while (0)
#define LOAD_A(dst_, mh_)                                                                            \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
    dst_[s_][i_] = *reinterpret_cast<const v8*>(a_base + boff + ((mh_) * 4 + i_) * 2048 + (s_ ? (c0 ^ 64) : c0))
#define LOAD_W(dst_, nh_)                                                                            \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                   \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                   \
    dst_[s_][j_] = *reinterpret_cast<const v8*>(w_base + boff + (nh_) * W_NH_STRIDE + j_ * W_NJ_STRIDE + (s_ ? (cw0 ^ 64) : cw0))
#define MATH(af_, wf_, mh_, nh_)                                                                     \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                 \
      acc[(mh_) * 4 + i_][(nh_) * 2 + j_] = SWAP                                                     \
          ? H::mfma16(wf_[s_][j_], af_[s_][i_], acc[(mh_) * 4 + i_][(nh_) * 2 + j_])                 \
          : H::mfma16(af_[s_][i_], wf_[s_][j_], acc[(mh_) * 4 + i_][(nh_) * 2 + j_]);                \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)

  if (n_items > 0) {
  // ---- prologue: the first K-tile of the stream into stage 0 --------------------------------------
  dma_set_item(0);
  DMA2(sAe, pAe, 0, 0);
  DMA2(sWe, pWe, OP_BYTES, 0);
  DMA2(sWl, pWl, OP_BYTES, 0);
  DMA2(sAl, pAl, 0, 0);
  dma_advance();
  WAIT_LOAD(4);
  BARRIER();
  if (wr == 1) BARRIER();          // waves 4-7 trail by one interval

  int flat = 0;                    // K-tiles consumed so far: stage = flat & 1
  for (int j = 0; j < n_items; ++j) {
    const Item it = item_at(j);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef COGAIM_STAMPS
    if (tid == 0 && j < 12) g_stamp_buf[(bid * 12 + j) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
    for (int kt = it.kt0; kt < it.kt1; ++kt, ++flat) {
#ifdef COGAIM_STAMPS
      if (tid == 0 && j < 12 && kt == it.kt0 + 1) g_stamp_buf[(bid * 12 + j) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
      const int st = flat & 1;
      const int boff = st * STAGE_BYTES;
      const bool more = (DIAG == 1) ? false : d_valid;       // wave-uniform
      v8 af[2][4], wf0[2][2], wf1[2][2];
      // ---- phase 0: quadrant (m 0-63, n 0-31)
      LOAD_A(af, 0);
      LOAD_W(wf0, 0);
      if (more) { DMA2(sAe, pAe, 0, st ^ 1); WAIT_LOAD(4); } else { WAIT_LOAD(2); }
      BARRIER();
      MATH(af, wf0, 0, 0);
      BARRIER();
      // ---- phase 1: (m 0-63, n 32-63)
      LOAD_W(wf1, 1);
      if (more) { DMA2(sWe, pWe, OP_BYTES, st ^ 1); WAIT_LOAD(4); } else { WAIT_LOAD(0); }
      BARRIER();
      MATH(af, wf1, 0, 1);
      BARRIER();
      // ---- phase 2: (m 64-127, n 32-63)
      LOAD_A(af, 1);
      if (more) { DMA2(sWl, pWl, OP_BYTES, st ^ 1); WAIT_LOAD(4); } else { WAIT_LOAD(0); }
      BARRIER();
      MATH(af, wf1, 1, 1);
      BARRIER();
      // ---- phase 3: (m 64-127, n 0-31)
      if (more) { DMA2(sAl, pAl, 0, st ^ 1); dma_advance(); WAIT_LOAD(4); }
      BARRIER();
      MATH(af, wf0, 1, 0);
      BARRIER();
    }

#ifdef COGAIM_STAMPS
    if (tid == 0 && j < 12) g_stamp_buf[(bid * 12 + j) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    const int z = it.tile / sc.tiles_per_z, lt = it.tile - z * sc.tiles_per_z;
    int tm, tn; tile_coords(sc, lt, tm, tn);
    if (it.slice < 0) {
      const int m_base = tm * BM2 + wr * 128, n_base = tn * BN2 + wc * 64;
      float peak = 0.f;           // fp16 range guard of the 16-bit epilogues (see track_peak)
      // all loads of the tile first, then the stores (see the note above the epilogues in gemm.hip)
      if constexpr (Epi::kReduce == 1) {
        epi.reduce_tile(z, m_base, n_base, tn * 4 + wc, acc, fr, fq);
      } else if constexpr (Epi::kReduce == 2) {
        epi.reduce_tile(z, m_base, n_base, tm * 2 + wr, acc, fr, fq);
      } else if constexpr (WIDE) {
        float4 cb[4], cl;
#pragma unroll
        for (int q = 0; q < 4; ++q) epi.load_cols(n_base + 8 * fq + 4 * (q & 1) + 32 * (q >> 1), cb[q], cl);
        PIN_LOADED4(cb[0]); PIN_LOADED4(cb[1]); PIN_LOADED4(cb[2]); PIN_LOADED4(cb[3]);
        if ((tm + 1) * BM2 <= g.M && (tn + 1) * BN2 <= g.N && epi.lines_ok()) {   // wave-uniform
#pragma unroll
          for (int i = 0; i < 8; ++i) epi.finish8x2n_lines(z, m_base + 16 * i + fr, n_base + 8 * fq, acc[i], cb, peak);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) epi.finish8x2n(z, m_base + 16 * i + fr, n_base + 8 * fq, acc[i], cb, peak);
        }
      } else if constexpr (SWAP) {
        constexpr int RG = 1;   // fragment rows per group of residual loads (2 spills registers)
        float4 cb[4], cl[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) epi.load_cols(n_base + 16 * jj + 4 * fq, cb[jj], cl[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if constexpr (Epi::kColVecs >= 1) PIN_LOADED4(cb[jj]);
          if constexpr (Epi::kColVecs >= 2) PIN_LOADED4(cl[jj]);
        }
#pragma unroll
        for (int part = 0; part < 8 / RG; ++part) {   // 4 RG residual loads in flight per lane (register budget)
          float4 xr[RG][4];
#pragma unroll
          for (int i = 0; i < RG; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              xr[i][jj] = Epi::kResidual
                  ? epi.load_res(z, m_base + 16 * (RG * part + i) + fr, n_base + 16 * jj + 4 * fq) : cb[jj];
          if constexpr (Epi::kResidual) {             // one wait per group instead of one per store
#pragma unroll
            for (int i = 0; i < RG; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) PIN_LOADED4(xr[i][jj]);
          }
#pragma unroll
          for (int i = 0; i < RG; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              epi.finish4n(z, m_base + 16 * (RG * part + i) + fr, n_base + 16 * jj + 4 * fq, acc[RG * part + i][jj],
                           cb[jj], cl[jj], xr[i][jj], peak);
        }
      } else {
        float cb[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) cb[jj] = epi.load_col1(n_base + 16 * jj + fr);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            epi.finish4m(z, m_base + 16 * i + 4 * fq, n_base + 16 * jj + fr, acc[i][jj], cb[jj]);
      }
      epi.flush_peak(peak);
    } else {
      float4* out = reinterpret_cast<float4*>(sc.ws) +
                    ((size_t)((it.tile - sc.n_full) * sc.slices + it.slice) * 32) * 512 + tid;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          out[(size_t)(i * 4 + jj) * 512] = make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
    }
#ifdef COGAIM_STAMPS
    if (lane == 0 && j < 12)   // slowest wave
      atomicMax(reinterpret_cast<unsigned long long*>(&g_stamp_buf[(bid * 12 + j) * 4 + 3]),
                (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
  }
  if (wr == 0) BARRIER();          // match the extra barrier of the trailing group: both groups are level again
  }
#ifdef COGAIM_STAMPS
  if (tid == 0) {
    g_clock_buf[(cclass * 256 + bid) * 4 + 2] = __builtin_amdgcn_s_memtime();
    g_clock_buf[(cclass * 256 + bid) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // ---- quadrant items: the partial last round ------------------------------------------------------
  // Each leftover tile goes to four workgroups (on one XCD: they stream the same panels); a workgroup runs ONE
  // phase per K-tile -- every wave its quadrant (MH, NH) -- over the whole K, with only the half of the A rows
  // and the half of the W rows that quadrant reads staged (4 DMA pieces per wave and K-tile), two stages, two
  // barriers per K-tile: DMA of K-tile kt+1 | wait for kt | barrier | read, 16 MFMAs | barrier.
  if (sc.quads) {
    const int qt = ((bid >> 5) << 3) + (bid & 7), q = (bid >> 3) & 3;
    if (qt < sc.n_split) {
      const int tile = sc.n_full + qt;
      // The lane id goes through an empty asm first: otherwise the compiler computes this block's per-lane source
      // addresses ahead of the main loop (they depend on nothing it produces) and keeps them in registers across it,
      // which costs the main loop 4-7 spilled registers.
      int lane_q = lane;
      asm volatile("" : "+v"(lane_q));
      dma_set_tile(tile, lane_q);
      const int z = tile / sc.tiles_per_z, lt = tile - z * sc.tiles_per_z;
      int tm, tn; tile_coords(sc, lt, tm, tn);
      const int m_base = tm * BM2 + wr * 128, n_base = tn * BN2 + wc * 64;
#define QUAD_DMA(MH_, NH_, stage_)                                                                    \
  do {                                                                                               \
    if (MH_ == 0) DMA2(sAe, pAe, 0, stage_); else DMA2(sAl, pAl, 0, stage_);                          \
    if (NH_ == 0) DMA2(sWe, pWe, OP_BYTES, stage_); else DMA2(sWl, pWl, OP_BYTES, stage_);            \
  } while (0)
#define QUAD_ITEM(MH_, NH_)                                                                           \
  do {                                                                                               \
    f32x4 qa[4][2];                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) qa[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};          \
    d_kt = 0;                                                                                        \
    QUAD_DMA(MH_, NH_, 0);                                                                           \
    for (int kt = 0; kt < nk; ++kt) {                                                                \
      const int st = kt & 1;                                                                         \
      const int boff = st * STAGE_BYTES;                                                             \
      if (kt + 1 < nk) { d_kt = kt + 1; QUAD_DMA(MH_, NH_, st ^ 1); WAIT_LOAD(4); } else { WAIT_LOAD(0); } \
      BARRIER();                                                                                     \
      v8 af[2][4], wf[2][2];                                                                         \
      LOAD_A(af, MH_);                                                                               \
      LOAD_W(wf, NH_);                                                                               \
      __builtin_amdgcn_s_setprio(1);                                                                 \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                               \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                               \
        qa[i_][j_] = SWAP ? H::mfma16(wf[s_][j_], af[s_][i_], qa[i_][j_])                            \
                          : H::mfma16(af[s_][i_], wf[s_][j_], qa[i_][j_]);                           \
      __builtin_amdgcn_s_setprio(0);                                                                 \
      BARRIER();                                                                                     \
    }                                                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                               \
      const int i = MH_ * 4 + i_, jj = NH_ * 2 + j_;                                                 \
      if constexpr (WIDE) epi.store4n(z, m_base + 16 * i + fr, n_base + 8 * fq + 4 * (jj & 1) + 32 * (jj >> 1), qa[i_][j_]); \
      else if constexpr (SWAP) epi.store4n(z, m_base + 16 * i + fr, n_base + 16 * jj + 4 * fq, qa[i_][j_]); \
      else epi.store4m(z, m_base + 16 * i + 4 * fq, n_base + 16 * jj + fr, qa[i_][j_]);               \
    }                                                                                                \
  } while (0)
      switch (q) {                      // phase order of the main loop: (0,0) (0,1) (1,1) (1,0)
        case 0: QUAD_ITEM(0, 0); break;
        case 1: QUAD_ITEM(0, 1); break;
        case 2: QUAD_ITEM(1, 1); break;
        default: QUAD_ITEM(1, 0); break;
      }
#undef QUAD_ITEM
#undef QUAD_DMA
    }
  }

#undef DMA2
#undef WAIT_LOAD
#undef PIN_LOADED4
#undef BARRIER
#undef LOAD_A
#undef LOAD_W
#undef MATH
}

// Sums the K-slice partials of the split tiles in slice order and runs the epilogue
// (one accumulator fragment per thread: blockIdx.y = mi*4 + ni).
template <class Epi, bool SWAP>
__global__ __launch_bounds__(512) void gemm256p_fixup_kernel(Epi epi, PersistSched sc) {
  constexpr bool WIDE = SWAP && Epi::kWide;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st = blockIdx.x, i = blockIdx.y >> 2, jj = blockIdx.y & 3;
  const int tile = sc.n_full + st;
  const int z = tile / sc.tiles_per_z, lt = tile - z * sc.tiles_per_z;
  int tm, tn; tile_coords(sc, lt, tm, tn);
  const float4* in = reinterpret_cast<const float4*>(sc.ws) + ((size_t)st * sc.slices * 32 + blockIdx.y) * 512 + tid;
  float4 a = in[0];
  for (int s = 1; s < sc.slices; ++s) {
    const float4 b = in[(size_t)s * 32 * 512];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  const f32x4 v{a.x, a.y, a.z, a.w};
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
  const int m_base = tm * BM2 + wr * 128, n_base = tn * BN2 + wc * 64;
  if constexpr (WIDE) epi.store4n(z, m_base + 16 * i + fr, n_base + 8 * fq + 4 * (jj & 1) + 32 * (jj >> 1), v);
  else if constexpr (SWAP) epi.store4n(z, m_base + 16 * i + fr, n_base + 16 * jj + 4 * fq, v);
  else epi.store4m(z, m_base + 16 * i + 4 * fq, n_base + 16 * jj + fr, v);
}

// Host side.  persistent: at most one workgroup per CU, each walking whole rounds of tiles, the
// remainder K-split over the first workgroups.  Otherwise one workgroup per whole tile plus extra
// workgroups for the K-slices.  Without a workspace the remainder runs as whole tiles.
template <class T, class Epi, bool SWAP>
int launch_persistent(const GemmShape& g, const Epi& epi, int batch, hipStream_t st, void* workspace,
                      long long workspace_bytes, bool persistent, int diag) {
  constexpr int CUS = 256;
  const int tm2 = (g.M + BM2 - 1) / BM2, tn2 = (g.N + BN2 - 1) / BN2;
  const int tiles_per_z = tm2 * tn2, tiles = tiles_per_z * batch, nk = g.K / BK;
  const int lanes = tiles < CUS ? tiles : CUS;                 // concurrent workgroups
  // 8 x 4 blocks per XCD when the column tiles divide evenly into fours and there is more W than one L2 keeps
  const int band = (g_gemm_band >= 0) ? g_gemm_band : ((tn2 % 4 == 0 && tn2 >= 8) ? 8 : 1);
  PersistSched sc{tm2, tn2, band, tiles_per_z, (tiles / lanes) * lanes, 0, 0, 1, 0, (float*)workspace, 0, 0};
  const int rem = tiles - sc.n_full;
  if (rem > 0 && persistent && g_gemm_quads && rem * 4 <= lanes && lanes == CUS && nk <= 16 && Epi::kReduce == 0) {
    // a short last round of a short K: quadrant items on 4 x rem workgroups (every workgroup has whole tiles before
    // it).  Same-box A/B at M = 43840: K = 768 shapes 3-4 us faster than K-split + fix-up launch (fc1 226 -> 222,
    // QKV 150 -> 146), K = 3072 16 us slower (a quadrant item walks all 48 K-tiles at its latency-bound pace)
    sc.n_split = rem;
    sc.quads = 1;
  } else if (rem > 0) {
    sc.n_split = rem;
    int slices = lanes / rem;
    if (slices > nk) slices = nk;
    // at least two K-tiles per slice and at most 8 slices: a slice costs a 256 KB partial store and a pass of the
    // fix-up whatever its length (K = 768: 6 slices run 1% faster than 8 or 12)
    const int cap = g_gemm_slice_cap > 0 ? g_gemm_slice_cap : (nk / 2 < 8 ? (nk / 2 > 1 ? nk / 2 : 1) : 8);
    if (slices > cap) slices = cap;
    if (slices >= 2 && workspace && (long long)rem * slices * 32 * 512 * 16 <= workspace_bytes) sc.slices = slices;
  }
  sc.nblk_full = persistent ? lanes : sc.n_full;
  sc.slice_block0 = persistent ? 0 : sc.nblk_full;
  const int grid = persistent ? lanes : sc.nblk_full + sc.n_split * sc.slices;
  if (persistent && g_gemm_prefetch && SWAP && lanes == CUS && prefetch_eligible(g, tiles)) {
    // whole rounds on the fragment-prefetch kernel; the partial last round (same split as above, same results) as
    // quadrant items of that kernel's first workgroups, or behind it as a launch of its own
    const bool fast_tail = g_gemm_fast_tail && Epi::kReduce == 0 && rem > 0 && (sc.quads || sc.slices >= 2);
    if (fast_tail && g_gemm_fast_tail >= 2 && !g_gemm_skip_tail && rem * sc.slices * 4 <= CUS && !g_gemm_wide)
      sc.tail_items = (g_gemm_fast_tail == 3 ? -1 : 1) * rem * sc.slices * 4;
    if (g_gemm_wide) launch_wide<T, Epi>(g, epi, sc, st, g_wide_diag);
    else launch_prefetch<T, Epi>(g, epi, sc, st);
    if (rem > 0 && !g_gemm_skip_tail) {
      PersistSched tail = sc;
      tail.nblk_full = 0;           // no whole tiles for anybody: K-slice items or quadrant items only
      if (sc.tail_items) {
      } else if (fast_tail) {
        launch_tail<T, Epi>(g, epi, tail, st);     // quadrant items of every tile or K-slice, deep DMA prefetch
      } else {
        const int tgrid = tail.quads ? CUS : tail.n_split * tail.slices;
        hipLaunchKernelGGL((gemm256p_kernel<T, Epi, SWAP, 0>), dim3(tgrid), dim3(512), 0, st, g, epi, tail);
      }
      if (sc.slices >= 2)
        hipLaunchKernelGGL((gemm256p_fixup_kernel<Epi, SWAP>), dim3(sc.n_split, 32), dim3(512), 0, st, epi, sc);
    }
    return COGAIM_LAUNCH_CHECK();
  }
  if (diag == 1)
    hipLaunchKernelGGL((gemm256p_kernel<T, Epi, SWAP, 1>), dim3(grid), dim3(512), 0, st, g, epi, sc);
  else
    hipLaunchKernelGGL((gemm256p_kernel<T, Epi, SWAP, 0>), dim3(grid), dim3(512), 0, st, g, epi, sc);
  if (sc.n_split && sc.slices >= 2)
    hipLaunchKernelGGL((gemm256p_fixup_kernel<Epi, SWAP>), dim3(sc.n_split, 32), dim3(512), 0, st, epi, sc);
  return COGAIM_LAUNCH_CHECK();
}
