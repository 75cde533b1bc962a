// valu_rate.hip - what one gfx950 SIMD really sustains: v_fma_f32 against v_pk_fma_f32, float64 chains, selects, Philox multiplies, the floor of a short launch, by
// waves per SIMD.  Settles how the VALU-bound kernels of the path (generated-grid rollout, k_actor_opt, k_ticks) are
// priced and what "packing two candidates per instruction" can buy.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o build/valu_rate && build/valu_rate
//
// Every wave runs ITER iterations of 8 independent accumulator chains (so no instruction waits for its predecessor);
// the grid is 256 CUs x 4 SIMDs x W waves.  Printed: lane-FMAs per second (one v_fma_f32 = 64, one v_pk_fma_f32 = 128),
// as a fraction of 7.86e13 (256 x 4 x 32 lanes x 2.4 GHz), and cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_ext.h>
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

// ---- round 4, second part: what bounds k_critic_fit (float64 chains, 64-bit selects), the candidate generator (Philox's
// 32 x 32 -> 64-bit multiplies) and the short launches of k_actor_dma_packed (the floor of a 1024-block launch) ----------------
__global__ __launch_bounds__(256) void k_fma64(float* out, float a, float b) {  // 8 independent float64 chains
  double acc[8];
  const double av = a, bv = b;
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_fma64_dep(float* out, float a, float b) {  // ONE chain: every fma waits for the last
  double acc = threadIdx.x * 1e-3;
  const double av = a, bv = b;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
__global__ __launch_bounds__(256) void k_fma32_dep(float* out, float a, float b) {
  float acc = threadIdx.x * 1e-3f;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_cndmask(float* out, float a, float b) {  // half of a 64-bit select
  float acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : );
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define RCG_RATE_KERNEL(NAME, ASM)                                                         \
  __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {              \
    float acc[8];                                                                          \
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3f + i;                          \
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21"); \
    for (int it = 0; it < ITER; ++it) {                                                    \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int i = 0; i < 8; ++i) \
          asm volatile(ASM : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc", "s20", "s21");          \
    }                                                                                      \
    float s = 0;                                                                           \
    for (int i = 0; i < 8; ++i) s += acc[i];                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
  }
RCG_RATE_KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
RCG_RATE_KERNEL(k_cnd_vcc2, "v_cndmask_b32 %0, %0, %1, vcc")
RCG_RATE_KERNEL(k_bfi, "v_bfi_b32 %0, %2, %1, %0")
RCG_RATE_KERNEL(k_and, "v_and_b32 %0, %1, %0")
RCG_RATE_KERNEL(k_mov, "v_mov_b32 %0, %1")
RCG_RATE_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
RCG_RATE_KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
RCG_RATE_KERNEL(k_max, "v_max_f32 %0, %0, %1")

// eight instructions in ONE asm statement (the compiler inserts nothing between them)
#define RCG_RATE_KERNEL8(NAME, I, SEP)                                                                           \
  __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                                    \
    float acc[8];                                                                                                \
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3f + i;                                                \
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");               \
    for (int it = 0; it < ITER; ++it) {                                                                          \
      _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                              \
          asm volatile(I(0) SEP I(1) SEP I(2) SEP I(3) SEP I(4) SEP I(5) SEP I(6) SEP I(7)                       \
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),     \
                         "+v"(acc[6]), "+v"(acc[7])                                                              \
                       : "v"(a), "v"(b));                                                                        \
    }                                                                                                            \
    float s = 0;                                                                                                 \
    for (int i = 0; i < 8; ++i) s += acc[i];                                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                              \
  }
#define I_CNDVCC(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc"
#define I_MOV(n) "v_mov_b32 %" #n ", %8"
#define I_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
RCG_RATE_KERNEL8(k8_cnd_vcc, I_CNDVCC, "\n")
RCG_RATE_KERNEL8(k8_mov, I_MOV, "\n")
RCG_RATE_KERNEL8(k8_mov_nop, I_MOV, "\n s_nop 0\n")
RCG_RATE_KERNEL8(k8_mov_nop1, I_MOV, "\n s_nop 1\n")
RCG_RATE_KERNEL8(k8_dpp, I_DPP, "\n")
#define I_CNDVCC64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc"
#define I_CNDS(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]"
#define I_CNDVCC_B(n) "v_cndmask_b32 %" #n ", %9, %8, vcc"
RCG_RATE_KERNEL8(k8_cnd_vcc64, I_CNDVCC64, "\n")
RCG_RATE_KERNEL8(k8_cnd_s, I_CNDS, "\n")
RCG_RATE_KERNEL8(k8_cnd_vcc_b, I_CNDVCC_B, "\n")

__global__ __launch_bounds__(256) void k_mul64(float* out, float a, float b) {  // Philox's multiply: 32 x 32 -> 64 bits
  unsigned long long acc[8];
  const unsigned m = 0xD2511F53u + (unsigned)a;
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 2654435761u + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = (unsigned long long)m * (unsigned)acc[i] + (acc[i] >> 32);
        asm volatile("" : "+v"(acc[i]));
      }
  }
  unsigned long long s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_rcp64(float* out, float a, float b) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i + 1.5;
  for (int it = 0; it < ITER / 8; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(acc[i]));
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_null(float* out, int never) {
  if (never) out[threadIdx.x] = 1.0f;
}
__global__ __launch_bounds__(256) void k_touch(const float4* in, float4* out) {  // one dependent HBM round trip per wave, 16 B per lane
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i];
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
  struct V2 {
    const char* name;
    void (*k)(float*, float, float);
    double instr_per_iter;
    int iters;
  } v2[] = {{"v_fma_f64 (8 chains)", k_fma64, 32, ITER}, {"v_fma_f64 (1 chain, dependent)", k_fma64_dep, 32, ITER},
            {"v_fma_f32 (1 chain, dependent)", k_fma32_dep, 32, ITER}, {"v_cndmask_b32", k_cndmask, 32, ITER},
            {"32x32->64 multiply-add (Philox)", k_mul64, 32, ITER}, {"v_rcp_f64", k_rcp64, 32, ITER / 8},
            {"v_cndmask_b32_e64 (sgpr mask)", k_cnd_e64, 32, ITER}, {"v_cndmask_b32 (vcc set)", k_cnd_vcc2, 32, ITER},
            {"v_bfi_b32", k_bfi, 32, ITER}, {"v_and_b32", k_and, 32, ITER}, {"v_mov_b32", k_mov, 32, ITER},
            {"v_cmp_lt_f32 -> vcc", k_cmp, 32, ITER}, {"v_cmp_lt_f32 + v_cndmask (pair = 1)", k_cmp_cnd, 32, ITER},
            {"v_max_f32", k_max, 32, ITER},
            {"8 x v_cndmask_b32 vcc, one asm block", k8_cnd_vcc, 32, ITER}, {"8 x v_mov_b32, one asm block", k8_mov, 32, ITER},
            {"8 x (v_mov_b32; s_nop 0)", k8_mov_nop, 32, ITER}, {"8 x (v_mov_b32; s_nop 1)", k8_mov_nop1, 32, ITER},
            {"8 x v_mov_b32_dpp quad_perm", k8_dpp, 32, ITER},
            {"8 x v_cndmask_b32_e64 ..., vcc", k8_cnd_vcc64, 32, ITER}, {"8 x v_cndmask_b32_e64 ..., s[20:21]", k8_cnd_s, 32, ITER},
            {"8 x v_cndmask_b32 d, b, a, vcc (no RAW)", k8_cnd_vcc_b, 32, ITER}};
  for (auto& v : v2)
    for (int W : {1, 2, 4}) {
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
      const double instr_per_simd = (double)W * v.iters * v.instr_per_iter;
      printf("%-32s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", v.name, W,
             t * 2.4e9 / instr_per_simd);
    }
  // the floor of a short launch: an empty kernel and one HBM round trip, 1024 blocks of 256 threads (the grid of
  // k_actor_dma_packed at K = 16), duration from the dispatch's own start / stop stamps as bench.py takes them
  {
    float4 *in, *out;
    const size_t n = (size_t)1024 * 256;
    hipMalloc(&in, n * sizeof(float4));
    hipMalloc(&out, n * sizeof(float4));
    hipMemset(in, 0, n * sizeof(float4));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
      double sum = 0, mn = 1e9;
      const int reps = 200;
      for (int r = 0; r < reps + 20; ++r) {
        if (which == 0)
          hipExtLaunchKernelGGL(k_null, dim3(1024), dim3(256), 0, 0, a, b, 0, d, 0);
        else
          hipExtLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, a, b, 0, (const float4*)in, out);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r >= 20) {
          sum += ms;
          mn = ms < mn ? ms : mn;
        }
      }
      printf("%-32s 1024 blocks x 256 threads: mean %.2f us, min %.2f us per launch (dispatch start/stop stamps)\n",
             which == 0 ? "empty kernel" : "one 16-B load + store per lane", sum / reps * 1e3, mn * 1e3);
    }
  }
  return 0;
}
