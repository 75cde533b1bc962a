// bw_probe.hip - development microbenchmark: how fast can gfx950 READ a 1.34 GB tensor with the access
// patterns k_actor could use?  Not part of the product; build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CHK(x)                                                                 \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("%s: %s\n", #x, hipGetErrorString(e));                            \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// A: classic grid-stride, 16 B per lane, consecutive lanes consecutive addresses
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const v4f* __restrict__ p, size_t n4, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4f acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    v4f r[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) r[j] = NT ? __builtin_nontemporal_load(p + i + j * stride) : p[i + j * stride];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) acc += r[j];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// B: k_actor's pattern: each wave owns TILES consecutive 5-KiB tiles (NROW x 1 KiB), one tile's loads
// issued back to back, consumed, next tile...  PIPE: next tile's loads issued before consuming.
template <bool NT, int NROW, bool PIPE>
__global__ __launch_bounds__(256) void k_tiles(const v4f* __restrict__ p, long n_tiles_total, int tiles_per_wave,
                                               float* out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long t0 = wave * tiles_per_wave;
  if (t0 >= n_tiles_total) return;
  const v4f* g = p + t0 * NROW * 64 + lane;
  v4f acc = {0, 0, 0, 0};
  v4f r[NROW];
  if (PIPE) {
#pragma unroll
    for (int j = 0; j < NROW; ++j) r[j] = NT ? __builtin_nontemporal_load(g + j * 64) : g[j * 64];
  }
  for (int t = 0; t < tiles_per_wave; ++t) {
    if (PIPE) {
      v4f c[NROW];
#pragma unroll
      for (int j = 0; j < NROW; ++j) c[j] = r[j];
      g += NROW * 64;
      if (t + 1 < tiles_per_wave) {
#pragma unroll
        for (int j = 0; j < NROW; ++j) r[j] = NT ? __builtin_nontemporal_load(g + j * 64) : g[j * 64];
      }
#pragma unroll
      for (int j = 0; j < NROW; ++j) acc += c[j];
    } else {
#pragma unroll
      for (int j = 0; j < NROW; ++j) r[j] = NT ? __builtin_nontemporal_load(g + j * 64) : g[j * 64];
#pragma unroll
      for (int j = 0; j < NROW; ++j) acc += r[j];
      g += NROW * 64;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// C: LDS-free layout: lane l loads ITS OWN row (NROW x 16 B, row stride NROW*16 B) straight into
// registers; a wave's loads touch the same 5 KiB as in B, but each instruction is strided.
template <bool NT, int NROW, bool PIPE>
__global__ __launch_bounds__(256) void k_rows(const v4f* __restrict__ p, long n_tiles_total, int tiles_per_wave,
                                              float* out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long t0 = wave * tiles_per_wave;
  if (t0 >= n_tiles_total) return;
  const v4f* g = p + (t0 * 64 + lane) * NROW;
  v4f acc = {0, 0, 0, 0};
  v4f r[NROW];
  if (PIPE) {
#pragma unroll
    for (int j = 0; j < NROW; ++j) r[j] = NT ? __builtin_nontemporal_load(g + j) : g[j];
  }
  for (int t = 0; t < tiles_per_wave; ++t) {
    if (PIPE) {
      v4f c[NROW];
#pragma unroll
      for (int j = 0; j < NROW; ++j) c[j] = r[j];
      g += NROW * 64;
      if (t + 1 < tiles_per_wave) {
#pragma unroll
        for (int j = 0; j < NROW; ++j) r[j] = NT ? __builtin_nontemporal_load(g + j) : g[j];
      }
#pragma unroll
      for (int j = 0; j < NROW; ++j) acc += c[j];
    } else {
#pragma unroll
      for (int j = 0; j < NROW; ++j) r[j] = NT ? __builtin_nontemporal_load(g + j) : g[j];
#pragma unroll
      for (int j = 0; j < NROW; ++j) acc += r[j];
      g += NROW * 64;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}


// D: the production kernel's data path without its arithmetic: each wave owns `tiles_per_wave` consecutive tiles; a
// tile goes HBM -> LDS with NROW direct-to-LDS loads (global_load_lds_dwordx4 nt), every lane then pulls ITS row
// (NROW x 16 B) LDS -> registers, the next tile's loads are issued, the row is consumed.  REG: same, but the tile is
// loaded into registers and parked in LDS with ds_write_b128 (no direct-to-LDS path).
template <int NROW, bool REG>
__global__ __launch_bounds__(256) void k_tiles_lds(const v4f* __restrict__ p, long n_tiles_total, int tiles_per_wave,
                                                   float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * 4 + wv;
  const long t0 = wave * tiles_per_wave;
  if (t0 >= n_tiles_total) return;
  unsigned char* tile = smem + (size_t)wv * NROW * 1024;
  const unsigned char* g = reinterpret_cast<const unsigned char*>(p) + (size_t)t0 * NROW * 1024;
  const v4f* myrow = reinterpret_cast<const v4f*>(tile) + lane * NROW;
  v4f acc = {0, 0, 0, 0};
  v4f r[NROW];
  auto issue = [&](const unsigned char* gg) {
    if (REG) {
#pragma unroll
      for (int j = 0; j < NROW; ++j) r[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(gg) + j * 64 + lane);
    } else {
#pragma unroll
      for (int j = 0; j < NROW; ++j)
        __builtin_amdgcn_global_load_lds((glb_void*)(gg + j * 1024 + lane * 16), (lds_void*)(tile + j * 1024), 16, 0, 2);
    }
  };
  issue(g);
  for (int t = 0; t < tiles_per_wave; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (REG) {
#pragma unroll
      for (int j = 0; j < NROW; ++j) reinterpret_cast<v4f*>(tile)[j * 64 + lane] = r[j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    v4f c[NROW];
#pragma unroll
    for (int j = 0; j < NROW; ++j) c[j] = myrow[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    g += NROW * 1024;
    if (t + 1 < tiles_per_wave) issue(g);
#pragma unroll
    for (int j = 0; j < NROW; ++j) acc += c[j];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename F>
static double time_it(F launch, int reps = 20) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CHK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e-3;
}

int main(int argc, char** argv) {
  const long B = 65536, K = 256, R = 20;
  const size_t bytes = (size_t)B * K * R * 4;
  const size_t n4 = bytes / 16;
  v4f* d;
  float* out;
  CHK(hipMalloc(&d, bytes));
  CHK(hipMalloc(&out, 64));
  CHK(hipMemset(d, 0, bytes));
  printf("buffer %.3f GB\n", bytes / 1e9);
  if (argc > 1 && !strcmp(argv[1], "residency")) {
    // the production data path (LDS-DMA tiles) at different residencies (LDS request caps the blocks per CU) and
    // tiles per wave: is a low-residency launch limited by the bytes in flight?
    const long n_tiles = (long)B * K / 64;
    for (int lds_kb : {20, 36, 56}) {
      for (int tpw : {4, 32, 64}) {
        const unsigned tb = (unsigned)((n_tiles / tpw + 3) / 4);
        double best = 1e9, sum = 0;
        for (int win = 0; win < 5; ++win) {
          const double t = time_it([&] { hipLaunchKernelGGL((k_tiles_lds<5, false>), dim3(tb), dim3(256), lds_kb * 1024, 0, d, n_tiles, tpw, out); }, 50);
          best = t < best ? t : best;
          sum += t;
        }
        printf("LDS-DMA tiles  lds=%2d KB (%d blocks/CU) tiles/wave=%2d : mean %7.1f us %6.0f GB/s  best %7.1f us\n", lds_kb,
               160 / lds_kb, tpw, sum / 5 * 1e6, bytes / (sum / 5) / 1e9, best * 1e6);
      }
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "steady")) {
    // steady state: windows of 50 launches, 12 windows per variant (the first launches after the GPU wakes up run
    // in a clock transient; the bench's number is the long-run average)
    const long n_tiles = (long)B * K / 64;
    const unsigned tb = (unsigned)((n_tiles / 4 + 3) / 4);
    for (int v = 0; v < 5; ++v) {
      for (int win = 0; win < 6; ++win) {
        double t;
        if (v == 3)
          t = time_it([&] { hipLaunchKernelGGL((k_tiles_lds<5, false>), dim3(tb), dim3(256), 4 * 5 * 1024, 0, d, n_tiles, 4, out); }, 50);
        else if (v == 4)
          t = time_it([&] { hipLaunchKernelGGL((k_tiles_lds<5, true>), dim3(tb), dim3(256), 4 * 5 * 1024, 0, d, n_tiles, 4, out); }, 50);
        else if (v == 0)
          t = time_it([&] { hipLaunchKernelGGL((k_stream<true, 8>), dim3(8192), dim3(256), 0, 0, d, n4, out); }, 50);
        else if (v == 1)
          t = time_it([&] { hipLaunchKernelGGL((k_tiles<true, 5, true>), dim3(tb), dim3(256), 0, 0, d, n_tiles, 4, out); }, 50);
        else
          t = time_it([&] { hipLaunchKernelGGL((k_rows<true, 5, true>), dim3(tb), dim3(256), 0, 0, d, n_tiles, 4, out); }, 50);
        printf("%s window %2d : %7.1f us  %6.0f GB/s\n", v == 0 ? "stream nt unroll8" : v == 1 ? "tiles nt pipe tpw4 " : v == 2 ? "rows nt pipe tpw4  " : v == 3 ? "tiles LDS-DMA tpw4 " : "tiles reg->LDS tpw4",
               win, t * 1e6, bytes / t / 1e9);
      }
    }
    return 0;
  }
  for (int blocks : {2048, 4096, 8192, 16384}) {
    double t = time_it([&] { hipLaunchKernelGGL((k_stream<false, 4>), dim3(blocks), dim3(256), 0, 0, d, n4, out); });
    printf("stream   plain unroll4 blocks=%5d : %7.1f us  %6.0f GB/s\n", blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_stream<true, 4>), dim3(blocks), dim3(256), 0, 0, d, n4, out); });
    printf("stream   nt    unroll4 blocks=%5d : %7.1f us  %6.0f GB/s\n", blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_stream<true, 8>), dim3(blocks), dim3(256), 0, 0, d, n4, out); });
    printf("stream   nt    unroll8 blocks=%5d : %7.1f us  %6.0f GB/s\n", blocks, t * 1e6, bytes / t / 1e9);
  }
  const long n_tiles = (long)B * K / 64;
  for (int tpw : {4, 8, 16, 32, 64}) {
    const long waves = (n_tiles + tpw - 1) / tpw;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    double t = time_it([&] { hipLaunchKernelGGL((k_tiles<true, 5, false>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("tiles nt    nopipe tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_tiles<false, 5, false>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("tiles plain nopipe tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_tiles<true, 5, true>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("tiles nt    pipe   tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_tiles<false, 5, true>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("tiles plain pipe   tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_rows<true, 5, false>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("rows  nt    nopipe tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_rows<true, 5, true>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("rows  nt    pipe   tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((k_rows<false, 5, true>), dim3(blocks), dim3(256), 0, 0, d, n_tiles, tpw, out); });
    printf("rows  plain pipe   tpw=%2d blocks=%6u : %7.1f us  %6.0f GB/s\n", tpw, blocks, t * 1e6, bytes / t / 1e9);
  }
  return 0;
}
