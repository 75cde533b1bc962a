// valu_rate.hip - what one gfx950 SIMD really sustains on f32 vector arithmetic: v_fma_f32 against v_pk_fma_f32, by
// waves per SIMD.  Settles how the VALU-bound kernels of the path (generated-grid rollout, k_actor_opt, k_ticks) are
// priced and what "packing two candidates per instruction" can buy.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o build/valu_rate && build/valu_rate
//
// Every wave runs ITER iterations of 8 independent accumulator chains (so no instruction waits for its predecessor);
// the grid is 256 CUs x 4 SIMDs x W waves.  Printed: lane-FMAs per second (one v_fma_f32 = 64, one v_pk_fma_f32 = 128),
// as a fraction of 7.86e13 (256 x 4 x 32 lanes x 2.4 GHz), and cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 2048;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
  float acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk(float* out, float a, float b) {
  v2f acc[8];
  const v2f av = {a, a + 1e-3f}, bv = {b, b - 1e-3f};
  for (int i = 0; i < 8; ++i) acc[i] = v2f{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the mix the compiler makes of the generated-grid rollout today: per 64 instructions 36 packed, 19 moves, sin, cos
__global__ __launch_bounds__(256) void k_mix(float* out, float a, float b) {
  v2f acc[4];
  float sc[8];
  const v2f av = {a, a + 1e-3f}, bv = {b, b - 1e-3f};
  for (int i = 0; i < 4; ++i) acc[i] = v2f{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
  for (int i = 0; i < 8; ++i) sc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sc[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y;
  for (int i = 0; i < 8; ++i) s += sc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_trans(float* out, float a, float b) {
  float acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_sin_f32 %0, %0" : "+v"(acc[i]));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct V {
    const char* name;
    void (*k)(float*, float, float);
    double lane_ops_per_instr, instr_per_iter;
  } vs[] = {{"v_fma_f32", k_fma, 64, 32}, {"v_pk_fma_f32", k_pk, 128, 32}, {"mix 16 pk + 16 fma", k_mix, 96, 32},
            {"v_sin_f32", k_trans, 64, 32}};
  for (auto& v : vs)
    for (int W : {1, 2, 4, 8}) {  // waves per SIMD: W blocks of 256 threads per CU
      const dim3 grid(256 * W), block(256);
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(v.k, grid, block, 0, 0, d, 1.0001f, 0.5f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(v.k, grid, block, 0, 0, d, 1.0001f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double t = ms * 1e-3 / reps;
      const double instr_per_simd = (double)W * ITER * v.instr_per_iter;  // wave-instructions one SIMD issued
      const double lane = 256.0 * 4 * instr_per_simd * v.lane_ops_per_instr;
      printf("%-20s waves/SIMD %d: %.3e lane-FMA/s = %.2f of 7.86e13; %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
             v.name, W, lane / t, lane / t / 7.86e13, t * 2.4e9 / instr_per_simd);
    }
  return 0;
}
