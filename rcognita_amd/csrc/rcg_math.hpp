// rcg_math.hpp - device math for the rcognita hot path on gfx950.
//
// The robot heading alpha is unbounded (it reached -72 rad in a 2 s constant-torque run,
// SURVEY.md hard part 3), so the f32 path cannot use the bare v_sin_f32/v_cos_f32 fast forms.
// sincos_r<float> does an FMA-based three-constant Cody-Waite reduction by pi/2 followed by the
// classic degree-7/8 minimax polynomials on [-pi/4, pi/4]: ~25 VALU ops for both values; max abs
// error 8.7e-8 for |x| <= 1e5 (checked against float64 libm with an exact-FMA float32 emulation).  sincos_r<double> defers to the device libm (parity/debug build of the path).
#pragma once
#include <hip/hip_runtime.h>

namespace rcg {

template <typename real>
__device__ __forceinline__ void sincos_r(real x, real* s, real* c);

template <>
__device__ __forceinline__ void sincos_r<float>(float x, float* s, float* c) {
  // k = nearest integer to x * 2/pi
  const float kf = __builtin_rintf(x * 0.63661977236758134308f);
  // pi/2 = C1 + C2 + C3 with C1 = fl(pi/2), C2 = fl(pi/2 - C1), C3 = fl(pi/2 - C1 - C2)
  float r = __builtin_fmaf(kf, -1.57079637050628662109375f, x);
  r = __builtin_fmaf(kf, 4.37113882867379288655e-8f, r);   // -C2
  r = __builtin_fmaf(kf, 1.71512451000588187280e-15f, r);  // -C3
  const float r2 = r * r;
  // sin(r) = r + r^3 * (S1 + r^2 (S2 + r^2 S3)),  cos(r) = 1 - r^2/2 + r^4 (C0 + r^2 (C1 + r^2 C2))
  float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
  const float sr = __builtin_fmaf(ps * r2, r, r);
  float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
  const float cr = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
  const int k = (int)kf;
  // quadrant: k&1 swaps, sign of sin flips for k&2, sign of cos flips for (k+1)&2
  const float ss = (k & 1) ? cr : sr;
  const float cc = (k & 1) ? sr : cr;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}

template <>
__device__ __forceinline__ void sincos_r<double>(double x, double* s, double* c) {
  ::sincos(x, s, c);
}

// Hardware form for the f32 rollout fast path: exact two-constant reduction to r in [-pi, pi], then
// v_sin_f32 / v_cos_f32 (argument in revolutions).  7 VALU ops, two of them quarter rate, instead of ~25.
// Measured on MI355X against float64 libm (tools/trig_probe.hip): max abs error 3.7e-7 for |x| <= 1e3.
__device__ __forceinline__ void sincos_hw(float x, float* s, float* c) {
  const float kf = __builtin_rintf(x * 0.15915494309189533577f);
  float r = __builtin_fmaf(kf, -6.28318548202514648438f, x);  // fl(2 pi)
  r = __builtin_fmaf(kf, 1.74845553146951715e-7f, r);         // fl(2 pi) - 2 pi
  const float t = r * 0.15915494309189533577f;
  *s = __builtin_amdgcn_sinf(t);
  *c = __builtin_amdgcn_cosf(t);
}

// f64 rollout fast path (k_actor_dma<double>): FMA-based two-constant Cody-Waite reduction by pi/2 (the third constant
// would contribute k * 1e-33) and the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4] (the published
// fdlibm coefficients): ~35 f64 VALU ops for both values where the device libm's sincos costs several hundred and
// made the float64 rollout VALU-bound (0.66 ms per C2 launch with the candidate stream needing 0.41 ms).  Reduction
// error <= 1.2e-16 absolute for |x| <= 1e6; measured against float64 libm (tools/trig_probe.hip): max abs error 2.3e-16.
__device__ __forceinline__ void sincos_fast(double x, double* s, double* c) {
  const double kf = __builtin_rint(x * 0.63661977236758134308);
  double r = __builtin_fma(kf, -1.57079632679489655800e+00, x);  // fl(pi/2)
  r = __builtin_fma(kf, -6.12323399573676603587e-17, r);         // pi/2 - fl(pi/2)
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(ps * z, r, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  const double cr = __builtin_fma(pc * z, z, __builtin_fma(z, -0.5, 1.0));
  const int k = (int)kf;
  const double ss = (k & 1) ? cr : sr;
  const double cc = (k & 1) ? sr : cr;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}

template <typename real, bool HW>
__device__ __forceinline__ void sincos_sel(real x, real* s, real* c) {
  sincos_r<real>(x, s, c);
}
template <>
__device__ __forceinline__ void sincos_sel<double, true>(double x, double* s, double* c) {
  sincos_fast(x, s, c);
}
template <>
__device__ __forceinline__ void sincos_sel<float, true>(float x, float* s, float* c) {
  sincos_hw(x, s, c);
}

// Two float lanes of work per instruction: ext-vector arithmetic that hipcc selects as v_pk_mul_f32 / v_pk_fma_f32 (a scalar
// operand is a splat: op_sel on a register, or an SGPR pair).  Component-wise IEEE: pk_fma({a, b}, ...) rounds each
// component exactly as v_fma_f32 does, so a packed rollout reproduces the scalar one bit for bit.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float s) { return v2f{s, s}; }

// typed fused multiply-add (NB: __builtin_fma is the double form; on float operands it would
// silently promote the whole expression to f64)
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename real>
__device__ __forceinline__ real clamp_r(real v, real lo, real hi) {
  // np.clip semantics (systems.py:243): min(max(v, lo), hi); NaN propagates
  return v < lo ? lo : (v > hi ? hi : v);
}

// np.clip for a FINITE v and lo <= hi (generated candidates, trial points of the line search): one v_med3_f32 instead of two
// compares and two selects; the same value as clamp_r for every finite input (a NaN would come out as a bound)
__device__ __forceinline__ float clamp_fin(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ double clamp_fin(double v, double lo, double hi) { return __builtin_fmin(__builtin_fmax(v, lo), hi); }

template <typename real>
__device__ __forceinline__ bool finite_r(real v) {
  return __builtin_isfinite(v);
}

template <typename real>
__device__ __forceinline__ real inf_r() {
  return (real)__builtin_huge_val();
}

// ---- wave-wide argmin without LDS round trips ---------------------------------------------------------------------
// A float cost and its candidate index are packed into one 64-bit key whose unsigned order is "lower cost, then lower
// index" (the build's argmin rule).  The minimum over the 64 lanes is then four DPP stages inside each row of 16 lanes
// (quad swaps, half mirror, mirror: register-to-register, VALU latency) and three scalar-side mins over the four rows
// (v_readlane).  The __shfl_xor butterfly it replaces is 6 dependent rounds of ds_bpermute per value; at the end of a
// short-lived wave that latency chain cost the production kernel 7 % (measured with RCG_DBG=2).
__device__ __forceinline__ unsigned float_order_key(float v) {
  const unsigned b = __float_as_uint(v + 0.0f);  // -0 -> +0, so that float ties stay ties
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k) {
#define RCG_DPP_MIN(CTRL)                                                                                      \
  {                                                                                                            \
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xF, 0xF, false);         \
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, false); \
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                          \
    k = o < k ? o : k;                                                                                         \
  }
  RCG_DPP_MIN(0xB1)   // quad_perm [1,0,3,2]
  RCG_DPP_MIN(0x4E)   // quad_perm [2,3,0,1]
  RCG_DPP_MIN(0x141)  // row_half_mirror
  RCG_DPP_MIN(0x140)  // row_mirror
#undef RCG_DPP_MIN
  unsigned long long r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 16 * i);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 16 * i);
    r[i] = ((unsigned long long)hi << 32) | lo;
  }
  const unsigned long long a = r[1] < r[0] ? r[1] : r[0], b = r[3] < r[2] ? r[3] : r[2];
  return b < a ? b : a;
}

// v_readlane of a real (the lane index is wave-uniform)
__device__ __forceinline__ float readlane_r(float v, int l) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}
__device__ __forceinline__ double readlane_r(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}


}  // namespace rcg
