// What limits the 16-bit GEMM epilogue's stores: the CU or the chip?  Every workgroup writes `reps` tiles of 128 KB the way
// the epilogue does (16-byte streaming stores, whole 128-byte lines per 8 lanes), from 1, 2, 4 or 8 waves, on 8 ... 256
// workgroups (one per CU at most: the kernel claims 128 KB of LDS).  Prints microseconds per 128-KB tile and workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ub/store_rate tools/ub/store_rate.hip && tools/ub/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(512) void store_tiles(u32x4* out, int reps, int waves, size_t tile_stride) {
  __shared__ char hold[128 * 1024];                     // one workgroup per CU
  if (threadIdx.x == 0) hold[0] = 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves) return;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (int r = 0; r < reps; ++r) {
    u32x4* base = out + ((size_t)blockIdx.x * reps + r) * tile_stride;
    // 128 KB = 8192 16-byte pieces; a wave-instruction writes 64 of them (1 KB, eight whole 128-byte lines)
    for (int i = wave; i < 128; i += waves) {
      u32x4* p = base + (size_t)i * 64 + lane;
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}

// the epilogue's own pattern: the workgroup's tile is 256 rows x 512 bytes of a matrix with `ldc_bytes` between rows;
// wave (wr, wc) of 8 owns rows 128 wr .. +127, bytes 128 wc .. +127; one instruction writes eight rows' 128-byte lines
__global__ __launch_bounds__(512) void store_tiles_rows(char* out, int reps, size_t ldc_bytes, int tiles_n) {
  __shared__ char hold[128 * 1024];
  if (threadIdx.x == 0) hold[0] = 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (int r = 0; r < reps; ++r) {
    const int tile = r * gridDim.x + blockIdx.x;                  // persistent order: tile = round * workgroups + id
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    char* base = out + (size_t)tm * 256 * ldc_bytes + (size_t)tn * 512 + (size_t)wr * 128 * ldc_bytes + wc * 128;
    for (int i = 0; i < 16; ++i) {
      char* p = base + (size_t)(8 * i + (lane >> 3)) * ldc_bytes + (lane & 7) * 16;
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    }
  }
}

static void run_rows(char* buf, int wgs, int reps, int n_cols) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t ldc = (size_t)n_cols * 2;
  store_tiles_rows<<<wgs, 512>>>(buf, reps, ldc, n_cols / 256);
  (void)hipEventRecord(e0);
  store_tiles_rows<<<wgs, 512>>>(buf, reps, ldc, n_cols / 256);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us_tile = ms * 1e3 / reps;
  printf("epilogue pattern, N = %4d  workgroups %3d : %6.2f us per 128-KB tile and workgroup, %6.2f TB/s over the chip\n", n_cols, wgs,
         us_tile, wgs * 131072.0 / us_tile / 1e6);
}

// 64 workgroups write tiles (epilogue pattern) while the other 192 stream operand panels into LDS the way the GEMM's K loop
// does (16-byte LDS-DMA, 64 KB per step from a 48-MB region, i.e. out of L2 / the Infinity Cache) or do nothing:
// does a CU's store drain depend on what the OTHER CUs read?  Writers time themselves (100-MHz clock, wait for all stores).
__global__ __launch_bounds__(512) void writers_and_readers(char* out, const char* src, long long* ticks, int reps, size_t ldc_bytes,
                                                           int tiles_n, int read_steps, int policy) {
  __shared__ __attribute__((aligned(16))) char stage[128 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool writer = ((blockIdx.x >> 3) & 3) == 0;
  if (!writer) {
    for (int s = 0; s < read_steps; ++s) {
      const size_t off = (((size_t)blockIdx.x * 7 + s) % 768) * 65536;               // 768 x 64 KB = 48 MB
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const char* p = src + off + (size_t)(wave * 8 + i) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(stage + ((s & 1) * 64 + wave * 8 + i) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int wr = wave >> 2, wc = wave & 3;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const int wid = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);                           // 0..63
  long long total = 0;
  for (int r = 0; r < reps; ++r) {
    const int tile = r * 64 + wid;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    char* base = out + (size_t)tm * 256 * ldc_bytes + (size_t)tn * 512 + (size_t)wr * 128 * ldc_bytes + wc * 128;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 16; ++i) {
      u32x4* p = reinterpret_cast<u32x4*>(base + (size_t)(8 * i + (lane >> 3)) * ldc_bytes + (lane & 7) * 16);
      if (policy == 0) __builtin_nontemporal_store(v, p); else *p = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    total += __builtin_amdgcn_s_memrealtime() - t0;
    __builtin_amdgcn_s_sleep(127);                                                    // a gap, as a K loop would leave
  }
  if (threadIdx.x == 0) ticks[wid] = total;
}

static void run_mixed(char* buf, const char* src, int read_steps, int policy) {
  const int reps = 16;
  long long* ticks;
  (void)hipMalloc(&ticks, 64 * sizeof(long long));
  for (int it = 0; it < 2; ++it)
    writers_and_readers<<<256, 512>>>(buf, src, ticks, reps, 6144, 12, read_steps, policy);
  (void)hipDeviceSynchronize();
  long long h[64];
  (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 64; ++i) sum += (double)h[i];
  printf("64 writers (%s stores), 192 %s: %6.2f us per 128-KB tile until its stores have retired\n", policy ? "plain" : "streaming",
         read_steps ? "CUs streaming panels into LDS" : "CUs idle", sum / 64 / reps / 100.0);
  (void)hipFree(ticks);
}

template <bool NT>
static void run(u32x4* buf, int wgs, int waves, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  store_tiles<NT><<<wgs, 512>>>(buf, reps, waves, 8192);
  (void)hipEventRecord(e0);
  store_tiles<NT><<<wgs, 512>>>(buf, reps, waves, 8192);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us_tile = ms * 1e3 / reps;
  printf("%s  workgroups %3d  waves %d : %6.2f us per 128-KB tile and workgroup, %6.2f TB/s over the chip\n", NT ? "streaming" : "plain    ",
         wgs, waves, us_tile, wgs * 131072.0 / us_tile / 1e6);
}

int main() {
  const int reps = 64;
  u32x4* buf;
  if (hipMalloc(&buf, (size_t)256 * reps * 131072) != hipSuccess) return 1;
  for (int wgs : {8, 32, 64, 128, 256})
    for (int waves : {1, 2, 4, 8}) run<true>(buf, wgs, waves, reps);
  for (int wgs : {8, 256})
    for (int waves : {4, 8}) run<false>(buf, wgs, waves, reps);
  for (int n : {3072, 2304})
    for (int wgs : {8, 32, 64, 128, 256}) run_rows(reinterpret_cast<char*>(buf), wgs, 16, n);
  char* src;
  if (hipMalloc(&src, (size_t)48 << 20) == hipSuccess) {
    (void)hipMemset(src, 1, (size_t)48 << 20);
    for (int policy : {0, 1})
      for (int steps : {0, 400}) run_mixed(reinterpret_cast<char*>(buf), src, steps, policy);
    (void)hipFree(src);
  }
  (void)hipFree(buf);
  return 0;
}
