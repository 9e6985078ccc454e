// 256x256x64 GEMM main loop with operand fragments read one step ahead (included by gemm.hip, inside its anonymous
// namespace, after gemm_persistent.h and gemm_tail.h): the whole rounds of tiles of a large problem.  The partial last
// round runs inside the same launch as items of gemm_tail.h, one per workgroup for the first workgroups, ahead of
// (or behind) their whole tiles; K-sliced items leave partial sums for gemm256p_fixup_kernel.
//
// Same tile, same 8 waves (2 x 4, 128 x 64 each), same LDS image, same accumulator layout and same epilogues as
// gemm256p_kernel, and the same k order per output element, so the results are bit-identical.  What differs is how a
// K-tile is walked.  gemm256p alternates barrier-delimited LOAD and MATH intervals (eight barriers per K-tile; a
// wave's MFMAs wait for its own LDS reads, the partner wave group fills the gap).  Here a K-tile is eight steps of
// 8 MFMAs, one quadrant (64 rows x 32 columns) at one k32 half each, ordered so that a step changes ONE operand, and
// every step issues the LDS reads of the operand the next step changes between its own MFMAs.  Two A buffers and two
// W buffers of one k32 half (48 registers) are enough for that:
//   step  quadrant (rows, cols) k32   reads issued meanwhile
//    1    (0-63,   0-31)  0           W cols 32-63, k32 0
//    2    (0-63,   32-63) 0           A rows 64-127, k32 0
//    3    (64-127, 32-63) 0           A rows 64-127, k32 1
//    4    (64-127, 0-31)  0           W cols 0-31, k32 1       then: own reads done | BARRIER (regions Al, We free)
//    5    (64-127, 0-31)  1           W cols 32-63, k32 1
//    6    (64-127, 32-63) 1           A rows 0-63, k32 1
//    7    (0-63,   32-63) 1           -            then: all but the three youngest DMA pieces landed, own reads done | BARRIER
//    8    (0-63,   0-31)  1           A rows 0-63 and W cols 0-31, k32 0, of the NEXT K-tile
// Two workgroup barriers per K-tile.  Two 64-KB stages: the stage of K-tile f is refilled (K-tile f+2) one 1-KiB
// piece per wave and step, in the order its regions fall free -- pieces 0-2 (regions Al, We) in steps 5-7 of K-tile
// f, pieces 3-7 from step 8 through step 4 of K-tile f+1 -- so the vector-memory path sees an even trickle instead
// of a burst, and every piece has three steps or more to land.  The last K-tile of a tile issues pieces 3-7 together
// in its step 8, ahead of the epilogue's stores: vector-memory operations retire in order, and the next tile's
// barriers must not wait for the stores (first_allow: how many younger operations that wait may leave in flight).
// DMA addressing is scalar (a buffer resource per operand and tile, 32-bit lane offsets; rows past the matrix read
// as zero; past the last tile a zero-size resource drops the loads while they still count in vmcnt).
#pragma once
#include <type_traits>

template <class T, class Epi, int DIAG>
__global__ __launch_bounds__(512, 2) void gemm256f_kernel(GemmShape g, Epi epi, PersistSched sc) {
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

  // ---- the partial last round AHEAD of the whole tiles (tail_items < 0): its items then start on an idle memory
  // pipe instead of behind the last tile's epilogue stores, which a wait for any younger DMA piece also waits for
  if constexpr (Epi::kReduce == 0 && !DIAG) {
    if ((int)blockIdx.x < -sc.tail_items) {
      tail_item<T, Epi>(g, epi, sc, smem, blockIdx.x, tid);
      __builtin_amdgcn_s_barrier();               // the item's LDS reads are done in every wave
    }
  }

  // ---- DMA side -----------------------------------------------------------------------------------------------
  // A piece is 1 KiB = 8 stage rows.  Each wave brings two pieces of each of the four regions of a stage, named
  // after the fragments read from them: Ae = A rows 0-63 of every wave group (row mod 128 < 64), Al = the other A
  // rows, We = W rows of column halves 0-31 (row mod 64 < 32), Wl = the other W rows.  Cursor order of a K-tile's
  // eight pieces: Al, Al, We, We, Ae, Ae, Wl, Wl (the order in which the regions of a stage fall free, below).
  const int l3 = lane >> 3, lc = lane & 7;
  unsigned off[8];                                      // lane offsets (bytes from the tile's first A / W row)
  int pdst[8];                                          // LDS byte offsets inside a stage (wave-uniform)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = 2 * wave + u;                         // 0..15
    const int pAe = e < 8 ? e : e + 8, pAl = pAe + 8;
    const int pWe = (e >> 2) * 8 + (e & 3), pWl = pWe + 4;
    const int pa[2] = {pAl, pAe}, pw[2] = {pWe, pWl};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ra = pa[k] * 8 + l3, rw = pw[k] * 8 + l3;
      const int qa = (k == 0 ? 0 : 4) + u, qw = (k == 0 ? 2 : 6) + u;
      off[qa] = (unsigned)ra * (unsigned)(g.lda * 2) + (unsigned)((lc ^ swz_a(ra)) << 4);
      off[qw] = (unsigned)rw * (unsigned)(g.ldw * 2) + (unsigned)((lc ^ (WIDE ? swz_w(rw) : swz_a(rw))) << 4);
      pdst[qa] = pa[k] * 1024;
      pdst[qw] = OP_BYTES + pw[k] * 1024;
    }
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
      const bool is_a = q < 2 || (q >= 4 && q < 6);
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

  // ---- math side (fragment addressing of gemm256p_kernel) ------------------------------------------------------
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = (fq ^ swz_a(fr)) << 4;
  const int wrow = WIDE ? 8 * (fr >> 2) + (fr & 3) : fr;
  constexpr int W_NH_STRIDE = 32 * 128;
  constexpr int W_NJ_STRIDE = WIDE ? 4 * 128 : 16 * 128;
  const int cw0 = (fq ^ (WIDE ? swz_w(wrow) : swz_a(wrow))) << 4;
  const char* const a_rd0 = smem + (wr * 128 + fr) * 128 + c0;                  // + stage + (mh * 4 + i) * 2048
  const char* const a_rd1 = smem + (wr * 128 + fr) * 128 + (c0 ^ 64);
  const char* const w_rd0 = smem + OP_BYTES + (wc * 64 + wrow) * 128 + cw0;     // + stage + nh * W_NH_STRIDE + j * W_NJ_STRIDE
  const char* const w_rd1 = smem + OP_BYTES + (wc * 64 + wrow) * 128 + (cw0 ^ 64);

#define F_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ") lgkmcnt(0)" ::: "memory")
#define F_BARRIER()                                          \
  do {                                                       \
    __builtin_amdgcn_sched_barrier(0);                       \
    if constexpr (!(DIAG & 2)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);                       \
  } while (0)
#define F_LOAD_A(dst_, st_, mh_, s_)                                                                 \
  if constexpr (!(DIAG & 4)) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                        \
    dst_[i_] = *reinterpret_cast<const v8*>(((s_) ? a_rd1 : a_rd0) + (st_) + ((mh_) * 4 + i_) * 2048)
#define F_LOAD_W(dst_, st_, nh_, s_)                                                                 \
  if constexpr (!(DIAG & 4)) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                        \
    dst_[j_] = *reinterpret_cast<const v8*>(((s_) ? w_rd1 : w_rd0) + (st_) + (nh_) * W_NH_STRIDE + j_ * W_NJ_STRIDE)
// the emitted order of a step: nd_ times { MFMA, LDS read }, nv_ times { MFMA, pv_ DMA pieces }, the other MFMAs
#define F_MIX(nd_, nv_, pv_)                                                                         \
  do {                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < (nd_); ++k_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
    }                                                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < (nv_); ++k_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
      __builtin_amdgcn_sched_group_barrier(0x010, (pv_), 0);                                         \
    }                                                                                                \
    if ((nd_) + (nv_) < 8) __builtin_amdgcn_sched_group_barrier(0x008, 8 - (nd_) - (nv_), 0);        \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#define F_MATH(af_, wf_, mh_, nh_)                                                                   \
  if constexpr (!(DIAG & 16)) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                       \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                   \
    acc[(mh_) * 4 + i_][(nh_) * 2 + j_] = H::mfma16(wf_[j_], af_[i_], acc[(mh_) * 4 + i_][(nh_) * 2 + j_])
// Re-defines a loaded float4 through an empty asm (see gemm256p_kernel)
#define F_PIN4(v_) asm volatile("" : "+v"((v_).x), "+v"((v_).y), "+v"((v_).z), "+v"((v_).w))

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  v8 ax[4], ay[4], wa[2], wb[2];               // two A buffers, two W buffers, one k32 half each (see the table above)
  if constexpr ((DIAG & 4) != 0) {              // never loaded in this variant: defined, and opaque to the optimiser
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_) {
      ax[i_] = v8{}; ay[i_] = v8{};
      asm volatile("" : "+v"(ax[i_]), "+v"(ay[i_]));
    }
#pragma unroll
    for (int j_ = 0; j_ < 2; ++j_) {
      wa[j_] = v8{}; wb[j_] = v8{};
      asm volatile("" : "+v"(wa[j_]), "+v"(wb[j_]));
    }
  }
  int st_cur = 0;                               // stage of the K-tile being computed; the next one is in the other
  int first_allow = 3;                          // wave-uniform: what the next tile's first barrier may leave in flight: its own 3
                                                // pieces plus the stores the last epilogue is known to have issued behind its last wait

  // ---- prologue: K-tiles 0 and 1 whole (the first K-tile of a tile expects its successor issued) ----------------
#pragma unroll
  for (int q = 0; q < 8; ++q) dma_piece(q);
  dma_next();
#pragma unroll
  for (int q = 0; q < 8; ++q) dma_piece(q);
  dma_next();
  F_WAIT(8);
  F_BARRIER();
  F_LOAD_A(ax, 0, 0, 0);
  F_LOAD_W(wa, 0, 0, 0);

  // One K-tile.  FIRST: first of a tile (the previous tile's last K-tile issued this cursor K-tile's last five
  // pieces ahead of its epilogue, and the vector-memory operations issued since are the epilogue's).  LAST: last of
  // a tile.  Pieces: steps 5, 6, 7 issue pieces 0-2 of the cursor's K-tile (regions Al and We of this K-tile's
  // stage, last read in steps 3 and 4, free behind the barrier before step 5), step 8 and steps 1-4 of the next
  // K-tile pieces 3-7 (behind the barrier before step 8, after which nothing of this stage is read any more).  Every
  // piece has three steps or more until the barrier before the step 8 that first reads its K-tile, and that barrier
  // leaves the three youngest pieces in flight.
  auto ktile = [&](auto first_c, auto last_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value != 0, LAST = decltype(last_c)::value != 0;
    const int st_nxt = st_cur ^ STAGE_BYTES;
#define F_NV(k_) (FIRST ? 0 : 1)
    __builtin_amdgcn_sched_barrier(0);
    F_LOAD_W(wb, st_cur, 1, 0);                 // step 1
    if constexpr (!FIRST) dma_piece(4);
    F_MATH(ax, wa, 0, 0);
    F_MIX(2, F_NV(1), 1);
    F_LOAD_A(ay, st_cur, 1, 0);                 // step 2
    if constexpr (!FIRST) dma_piece(5);
    F_MATH(ax, wb, 0, 1);
    F_MIX(4, F_NV(2), 1);
    F_LOAD_A(ax, st_cur, 1, 1);                 // step 3
    if constexpr (!FIRST) dma_piece(6);
    F_MATH(ay, wb, 1, 1);
    F_MIX(4, F_NV(3), 1);
    F_LOAD_W(wb, st_cur, 0, 1);                 // step 4
    if constexpr (!FIRST) { dma_piece(7); dma_next(); }
    F_MATH(ay, wa, 1, 0);
    F_MIX(2, F_NV(4), 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own reads of regions Al and We done
    F_BARRIER();
    F_LOAD_W(wa, st_cur, 1, 1);                 // step 5
    dma_piece(0);
    F_MATH(ax, wb, 1, 0);
    F_MIX(2, 1, 1);
    F_LOAD_A(ay, st_cur, 0, 1);                 // step 6
    dma_piece(1);
    F_MATH(ax, wa, 1, 1);
    F_MIX(4, 1, 1);
    dma_piece(2);                               // step 7
    F_MATH(ay, wa, 0, 1);
    F_MIX(0, 1, 1);
    if (FIRST && first_allow == 19) F_WAIT(19);              // all but pieces 0-2 just issued (and the epilogue's last stores)
    else if (FIRST && first_allow == 11) F_WAIT(11);
    else F_WAIT(3);
    F_BARRIER();
    F_LOAD_A(ax, st_nxt, 0, 0);                 // step 8
    F_LOAD_W(wa, st_nxt, 0, 0);
    dma_piece(3);
    if constexpr (LAST) { dma_piece(4); dma_piece(5); dma_piece(6); dma_piece(7); dma_next(); }
    F_MATH(ay, wb, 0, 0);
    F_MIX(6, LAST ? 2 : 1, LAST ? 3 : 1);
    st_cur = st_nxt;
#undef F_NV
  };
  using std::integral_constant;
#define F_KT(first_, last_) ktile(integral_constant<int, first_>{}, integral_constant<int, last_>{})

  float peak = 0.f;           // fp16 range guard of the 16-bit epilogues (see track_peak)
  for (int j = 0; j < n_items; ++j) {
    if (j) item_rs(j + 1, na_lo, na_hi, na_n, nw_lo, nw_hi, nw_n);   // the cursor is inside tile j until its last K-tiles
    // (stamps builds: the 100-MHz clock at the start of a tile, after its first K-tile, after its last one and after
    // the epilogue was issued, taken by wave 0 -- tools/gemm_stamps.py)
#ifdef COGAIM_STAMPS
#define F_STAMP(k_) if (tid == 0 && j < 12) g_stamp_buf[(blockIdx.x * 12 + j) * 4 + (k_)] = __builtin_amdgcn_s_memrealtime()
#else
#define F_STAMP(k_)
#endif
    F_STAMP(0);
    F_KT(1, 0);
    F_STAMP(1);
    for (int kt = 2; kt < nk; ++kt) F_KT(0, 0);
    F_KT(0, 1);
    F_STAMP(2);

    // ---- epilogue (that of gemm256p_kernel) ------------------------------------------------------------------
    int z, row0, col0;
    tile_origin(j, z, row0, col0);
    first_allow = 3;
    if constexpr ((DIAG & 8) != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) asm volatile("" ::"v"(acc[i][jj]));
    } else {
      const int m_base = row0 + wr * 128, n_base = col0 + wc * 64;
      if constexpr (Epi::kReduce == 1) {
        epi.reduce_tile(z, m_base, n_base, (col0 / BN2) * 4 + wc, acc, fr, fq);
      } else if constexpr (Epi::kReduce == 2) {
        epi.reduce_tile(z, m_base, n_base, (row0 / BM2) * 2 + wr, acc, fr, fq);
      } else if constexpr (WIDE) {
        float4 cb[4], cl;
#pragma unroll
        for (int q = 0; q < 4; ++q) epi.load_cols(n_base + 8 * fq + 4 * (q & 1) + 32 * (q >> 1), cb[q], cl);
        F_PIN4(cb[0]); F_PIN4(cb[1]); F_PIN4(cb[2]); F_PIN4(cb[3]);
        if (row0 + BM2 <= g.M && col0 + BN2 <= g.N && epi.lines_ok()) {   // wave-uniform
#pragma unroll
          for (int i = 0; i < 8; ++i) epi.finish8x2n_lines(z, m_base + 16 * i + fr, n_base + 8 * fq, acc[i], cb, peak);
          first_allow = 19;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) epi.finish8x2n(z, m_base + 16 * i + fr, n_base + 8 * fq, acc[i], cb, peak);
        }
      } else {
        float4 cb[4], cl[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) epi.load_cols(n_base + 16 * jj + 4 * fq, cb[jj], cl[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if constexpr (Epi::kColVecs >= 1) F_PIN4(cb[jj]);
          if constexpr (Epi::kColVecs >= 2) F_PIN4(cl[jj]);
        }
        bool done = false;
        if constexpr (InPlaceRMW<Epi>::value) {
          if (row0 + BM2 <= g.M && col0 + BN2 <= g.N) {     // wave-uniform: every lane loads and stores in every part
            // Two sets of residual values: the loads of part p+1 are issued before the wait for part p, so a part
            // costs its arithmetic and store issue instead of a memory round trip.  The loads are asm statements
            // with counted waits: through plain loads the compiler answers every use with vmcnt(0), which also waits
            // for the previous part's stores.  Younger than the loads of part p at its wait: the 4 stores of part
            // p-1 and the 4 loads of part p+1.
            float* const xp = epi.x + (size_t)(m_base + fr) * epi.ldc + n_base + 4 * fq;
            const size_t pstep = (size_t)16 * epi.ldc;
            f32x4 xq[2][4];
#define F_RMW_LOAD(buf_, part_)                                                                      \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq[buf_][jj]) : "v"(xp + (part_) * pstep + 16 * jj) : "memory")
#define F_RMW_WAIT(n_, buf_)                                                                         \
  asm volatile("s_waitcnt vmcnt(" #n_ ")" : "+v"(xq[buf_][0]), "+v"(xq[buf_][1]), "+v"(xq[buf_][2]), "+v"(xq[buf_][3]) :: "memory")
            F_RMW_LOAD(0, 0);
#pragma unroll
            for (int part = 0; part < 8; ++part) {
              if (part < 7) {
                if (part & 1) { F_RMW_LOAD(0, part + 1); } else { F_RMW_LOAD(1, part + 1); }
              }
              if (part == 0 || part == 7) {
                if (part & 1) { F_RMW_WAIT(4, 1); } else { F_RMW_WAIT(4, 0); }
              } else {
                if (part & 1) { F_RMW_WAIT(8, 1); } else { F_RMW_WAIT(8, 0); }
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const f32x4 r = xq[part & 1][jj];
                const f32x4 a = acc[part][jj];
                float4 o;
                o.x = r[0] + (a[0] + cb[jj].x) * cl[jj].x; o.y = r[1] + (a[1] + cb[jj].y) * cl[jj].y;
                o.z = r[2] + (a[2] + cb[jj].z) * cl[jj].z; o.w = r[3] + (a[3] + cb[jj].w) * cl[jj].w;
                *reinterpret_cast<float4*>(xp + part * pstep + 16 * jj) = o;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
#undef F_RMW_LOAD
#undef F_RMW_WAIT
            first_allow = 11;                                 // the stores of the last two parts may still be in flight
            done = true;
          }
        }
        if (!done) {
#pragma unroll
        for (int part = 0; part < 8; ++part) {        // 4 residual loads in flight per lane, then their 4 stores
          float4 xr[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            xr[jj] = Epi::kResidual ? epi.load_res(z, m_base + 16 * part + fr, n_base + 16 * jj + 4 * fq) : cb[jj];
          if constexpr (Epi::kResidual) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) F_PIN4(xr[jj]);
          }
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            epi.finish4n(z, m_base + 16 * part + fr, n_base + 16 * jj + 4 * fq, acc[part][jj], cb[jj], cl[jj], xr[jj], peak);
        }
        }
      }
    }
    F_STAMP(3);
#undef F_STAMP
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  F_WAIT(0);                                    // dropped loads past the last tile included
  epi.flush_peak(peak);
  // ---- the partial last round behind the whole tiles: one item for each of the first workgroups (gemm_tail.h) ------
  if constexpr (Epi::kReduce == 0 && !DIAG) {
    if ((int)blockIdx.x < sc.tail_items) {
      int tid_q = tid;
      asm volatile("" : "+v"(tid_q));             // opaque: see tail_item
      __builtin_amdgcn_s_barrier();               // every wave is through with the stages
      tail_item<T, Epi>(g, epi, sc, smem, blockIdx.x, tid_q);
    }
  }
#undef F_KT
#undef F_PIN4
#undef F_MATH
#undef F_MIX
#undef F_LOAD_W
#undef F_LOAD_A
#undef F_BARRIER
#undef F_WAIT
}

// Whether gemm256f_kernel can take the whole rounds of a problem: at least one round on every CU, four K-tiles or
// more, operands addressable with 32-bit byte offsets.
inline bool prefetch_eligible(const GemmShape& g, long long tiles) {
  const int nk = g.K / BK;
  if (nk < 4 || tiles < 256) return false;   // (the DMA cursor must still be inside a tile when the math side starts it)
  return (long long)g.M * g.lda * 2 < (1LL << 31) && (long long)g.N * g.ldw * 2 < (1LL << 31);
}
