// The partial last round of tiles of a large GEMM (included by gemm.hip, inside its anonymous namespace, after
// gemm_persistent.h).  What is left after the whole rounds is 2 ... 64 tiles for 256 CUs, cut the way gemm256p cuts
// it -- four QUADRANT items per tile (each the same 64x32 quadrant of every wave's 128x64 over the whole K; results
// bit-identical to a whole tile) or, for long K, K-slices whose fp32 partial tiles gemm256p_fixup_kernel sums in slice
// order -- but every slice is cut into quadrants as well, and the walk over K is built for LATENCY instead of
// throughput: an item stages only the half of the A rows and the half of the W rows its quadrant reads (32 KB per
// K-tile), so four stages fit and the DMA runs three K-tiles ahead of the math, one barrier per K-tile.  In gemm256p a
// quadrant item walks K at one DMA round trip per K-tile (~1 us each, two stages); here a K-tile costs its LDS reads
// and 16 MFMAs.
#pragma once

// One item (tile, K-slice, quadrant) on a workgroup of 512 threads; `smem` is 128 KiB of LDS nobody else is using.
// `tid` is passed in so that a caller can make it opaque: everything here depends only on the thread index and the
// kernel arguments, and a compiler that sees that computes this function's per-lane addresses ahead of the caller's
// own main loop and keeps them in registers across it.
template <class T, class Epi>
__device__ __forceinline__ void tail_item(const GemmShape& g, const Epi& epi, const PersistSched& sc, char* smem,
                                          const int item, const int tid) {
  constexpr int Q_OP = 128 * BK * 2;                    // 16 KiB: half of an operand's rows
  constexpr int Q_STAGE = 2 * Q_OP;                     // 32 KiB
  using H = Half<T>;
  using v8 = typename H::v8;
  constexpr bool WIDE = Epi::kWide;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = g.K / BK;
  // item: (leftover tile, K-slice, quadrant)
  const int q = item & 3, rest = item >> 2;
  const int slice = rest % sc.slices, st = rest / sc.slices;
  if (st >= sc.n_split) return;
  const int MH = q >> 1, NH = q & 1;
  const int tile = sc.n_full + st;
  const int z = tile / sc.tiles_per_z;
  int tm, tn;
  tile_coords(sc, tile - z * sc.tiles_per_z, tm, tn);
  const int kt0 = (int)((long long)nk * slice / sc.slices), kt1 = (int)((long long)nk * (slice + 1) / sc.slices);
  const int nkk = kt1 - kt0;

  // ---- DMA: wave w brings pieces 2w, 2w+1 of the A half and of the W half (1 KiB = 8 compact rows each) ------------
  const int l3 = lane >> 3, lc = lane & 7;
  unsigned offA[2], offW[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int cr = (2 * wave + u) * 8 + l3;             // compact row 0..127
    const int ra = (cr >> 6) * 128 + MH * 64 + (cr & 63);           // row of the tile's 256 A rows
    const int rw = (cr >> 5) * 64 + NH * 32 + (cr & 31);            // row of the tile's 256 W rows
    offA[u] = (unsigned)ra * (unsigned)(g.lda * 2) + (unsigned)((lc ^ swz_a(ra)) << 4);
    offW[u] = (unsigned)rw * (unsigned)(g.ldw * 2) + (unsigned)((lc ^ (WIDE ? swz_w(rw) : swz_a(rw))) << 4);
  }
  const int row0 = tm * BM2, col0 = tn * BN2;
  const T* const Ab = reinterpret_cast<const T*>(g.A) + z * g.strideA + (size_t)row0 * g.lda;
  const T* const Wb = reinterpret_cast<const T*>(g.W) + z * g.strideW + (size_t)col0 * g.ldw;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(Ab, (unsigned)(g.M - row0) * (unsigned)(g.lda * 2));
  const __amdgpu_buffer_rsrc_t rW = make_rsrc(Wb, (unsigned)(g.N - col0) * (unsigned)(g.ldw * 2));
  const __amdgpu_buffer_rsrc_t rNone = make_rsrc(Ab, 0u);           // zero records: loads dropped, still counted
  auto issue = [&](int i) {                             // K-tile kt0 + i of the item into stage i & 3
    const bool ok = i < nkk;
    char* dst = smem + (i & 3) * Q_STAGE + (2 * wave) * 1024;
    const int ko = (kt0 + i) * (BK * 2);
    if (ok) {
      bufload_lds16(rA, offA[0], ko, dst); bufload_lds16(rA, offA[1], ko, dst + 1024);
      bufload_lds16(rW, offW[0], ko, dst + Q_OP); bufload_lds16(rW, offW[1], ko, dst + Q_OP + 1024);
    } else {
      bufload_lds16(rNone, offA[0], 0, dst); bufload_lds16(rNone, offA[1], 0, dst + 1024);
      bufload_lds16(rNone, offW[0], 0, dst + Q_OP); bufload_lds16(rNone, offW[1], 0, dst + Q_OP + 1024);
    }
  };

  // ---- fragments: compact rows wr*64 + 16 i + fr of the A half, wc*32 + wrow + (4 | 16) j of the W half ---------------
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = (fq ^ swz_a(fr)) << 4;
  const int wrow = WIDE ? 8 * (fr >> 2) + (fr & 3) : fr;
  constexpr int W_NJ_STRIDE = WIDE ? 4 * 128 : 16 * 128;
  const int cw0 = (fq ^ (WIDE ? swz_w(wrow) : swz_a(wrow))) << 4;
  const char* const a_rd0 = smem + (wr * 64 + fr) * 128 + c0;
  const char* const a_rd1 = smem + (wr * 64 + fr) * 128 + (c0 ^ 64);
  const char* const w_rd0 = smem + Q_OP + (wc * 32 + wrow) * 128 + cw0;
  const char* const w_rd1 = smem + Q_OP + (wc * 32 + wrow) * 128 + (cw0 ^ 64);

  f32x4 qa[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) qa[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0); issue(1); issue(2);
  for (int i = 0; i < nkk; ++i) {
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     // K-tile i landed (own pieces); own reads of i-1 done
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(i + 3);                                        // into the stage K-tile i-1 was read from
    const int so = (i & 3) * Q_STAGE;
    v8 af[2][4], wf[2][2];
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_) {
      af[0][i_] = *reinterpret_cast<const v8*>(a_rd0 + so + i_ * 2048);
      af[1][i_] = *reinterpret_cast<const v8*>(a_rd1 + so + i_ * 2048);
    }
#pragma unroll
    for (int j_ = 0; j_ < 2; ++j_) {
      wf[0][j_] = *reinterpret_cast<const v8*>(w_rd0 + so + j_ * W_NJ_STRIDE);
      wf[1][j_] = *reinterpret_cast<const v8*>(w_rd1 + so + j_ * W_NJ_STRIDE);
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
        for (int j_ = 0; j_ < 2; ++j_) qa[i_][j_] = H::mfma16(wf[s_][j_], af[s_][i_], qa[i_][j_]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dropped loads behind the last K-tile

  const int m_base = row0 + wr * 128, n_base = col0 + wc * 64;
  if (sc.slices == 1) {
    // all loads of the quadrant first (column constants of its two column groups, the eight residual values), then
    // the arithmetic and the stores: fragment by fragment, every store would wait for the load behind it
    float peak = 0.f;
    int n0[2];
    float4 cb[2] = {}, cl[2] = {}, xr[4][2];
#pragma unroll
    for (int j_ = 0; j_ < 2; ++j_) {
      const int jj = NH * 2 + j_;
      n0[j_] = WIDE ? n_base + 8 * fq + 4 * (jj & 1) + 32 * (jj >> 1) : n_base + 16 * jj + 4 * fq;
      epi.load_cols(n0[j_], cb[j_], cl[j_]);
    }
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
      for (int j_ = 0; j_ < 2; ++j_)
        xr[i_][j_] = Epi::kResidual ? epi.load_res(z, m_base + 16 * (MH * 4 + i_) + fr, n0[j_]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
      for (int j_ = 0; j_ < 2; ++j_)
        epi.finish4n(z, m_base + 16 * (MH * 4 + i_) + fr, n0[j_], qa[i_][j_], cb[j_], Epi::kColVecs >= 2 ? cl[j_] : cb[j_],
                     Epi::kResidual ? xr[i_][j_] : cb[j_], peak);
    epi.flush_peak(peak);
  } else {                                               // this quadrant's 8 of the slice's 32 partial fragments
    float4* out = reinterpret_cast<float4*>(sc.ws) + ((size_t)(st * sc.slices + slice) * 32) * 512 + tid;
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
      for (int j_ = 0; j_ < 2; ++j_) {
        const int i = MH * 4 + i_, jj = NH * 2 + j_;
        out[(size_t)(i * 4 + jj) * 512] = make_float4(qa[i_][j_][0], qa[i_][j_][1], qa[i_][j_][2], qa[i_][j_][3]);
      }
  }
}

template <class T, class Epi>
__global__ __launch_bounds__(512, 2) void gemm256q_kernel(GemmShape g, Epi epi, PersistSched sc) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  tail_item<T, Epi>(g, epi, sc, smem, blockIdx.x, threadIdx.x);
}
