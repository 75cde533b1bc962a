// rcg_disturb.hpp - disturbance model of the environments (System(is_disturb=1)), SURVEY.md 8f row f4.
//
// Reference (rcognita/systems.py): full state = [state, disturb];
//   _disturb_dyn (:325-345, :384-394)   dq_k/dt = -tau_k (q_k + sigma_k (randn() + mu_k));  2tank: 0 (:421-424)
//   _state_dyn   (:308-323) 3wrobot     dv/dt = (F + q_0)/m,  domega/dt = (M + q_1)/I
//                (:370-382) 3wrobotNI   dx/dt += q_0, dy/dt += q_0 (sic), dalpha/dt += q_1
//                (:412-419) 2tank       unaffected
// The reference draws from the unseeded global RNG inside every right-hand-side evaluation.  Build-defined instead
// (mirrors oracle/disturb_oracle.py statement by statement): the noise is drawn once per RK4 substep and env and held
// over the four stages; generator = Philox4x32-10 with counter (env id lo, env id hi, EPISODE_IDX, SUBSTEP_IDX) and
// key (seed lo, seed hi); two normals by Box-Muller on 24-bit uniforms, in float64 whatever the handle's dtype.
// An env's stream is therefore independent of batch size, sharding and launch geometry.
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

struct PhiloxOut {
  uint32_t w[4];
};

// ROUNDS = 10: the published generator (its known answers are pinned in tests/test_disturb_oracle.py) - the disturbance model
// and the per-env key of the candidate stream; ROUNDS = 7: the candidate draws of k_actor_search / k_cand_sample (the least
// round count Salmon et al. report as passing BigCrush; the integer stream is bit-exact against oracle/search_oracle.py)
template <int ROUNDS>
__device__ __forceinline__ PhiloxOut philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return PhiloxOut{{c0, c1, c2, c3}};
}
__device__ __forceinline__ PhiloxOut philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                   uint32_t k1) {
  return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}

__device__ __forceinline__ PhiloxOut noise_bits(uint64_t seed, int64_t env_id, int32_t episode, int32_t substep) {
  const uint64_t e = (uint64_t)env_id;
  return philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)episode, (uint32_t)substep, (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}

__device__ __forceinline__ void normals_from_bits(const PhiloxOut& o, double* xi) {
  const double u0 = ((double)(o.w[0] >> 8) + 0.5) * 5.9604644775390625e-08;  // 2^-24
  const double u1 = ((double)(o.w[1] >> 8) + 0.5) * 5.9604644775390625e-08;
  const double r = ::sqrt(-2.0 * ::log(u0));
  double s, c;
  ::sincos(6.283185307179586476925286766559 * u1, &s, &c);
  xi[0] = r * c;
  xi[1] = r * s;
}

struct DisturbPars {
  double sigma[2], mu[2], tau[2];
  uint64_t seed;
  int64_t env_id_base;
};

// how the disturbance enters _state_dyn, and its dimension
template <typename Sys>
struct Disturb;
template <>
struct Disturb<Sys3WRobot> {
  static constexpr int DD = 2;
  static constexpr bool inert = false;
  template <typename real>
  __device__ __forceinline__ static void apply(const Sys3WRobot::Pre<real>& p, const real* u, const real* q, real* d) {
    d[3] = p.inv_m * (u[0] + q[0]);  // 1/m * (action[0] + disturb[0])
    d[4] = p.inv_I * (u[1] + q[1]);
  }
};
template <>
struct Disturb<Sys3WRobotNI> {
  static constexpr int DD = 2;
  static constexpr bool inert = false;
  template <typename real>
  __device__ __forceinline__ static void apply(const Sys3WRobotNI::Pre<real>&, const real*, const real* q, real* d) {
    d[0] += q[0];
    d[1] += q[0];  // the reference adds disturb[0] to both (systems.py:374-375)
    d[2] += q[1];
  }
};
template <>
struct Disturb<Sys2Tank> {
  static constexpr int DD = 1;
  static constexpr bool inert = true;  // _disturb_dyn returns zeros and _state_dyn ignores it
  template <typename real>
  __device__ __forceinline__ static void apply(const Sys2Tank::Pre<real>&, const real*, const real*, real*) {}
};

// closed_loop_rhs on the full state with the (already clipped) action and a given noise value
template <typename Sys, typename real>
__device__ __forceinline__ void rhs_full(const typename Sys::template Pre<real>& pre, const DisturbPars& D, const real* x,
                                         const real* q, const real* u, const real* xi, real* dx, real* dq) {
  constexpr int DD = Disturb<Sys>::DD;
  Sys::template rhs<real>(pre, x, u, dx);
  Disturb<Sys>::template apply<real>(pre, u, q, dx);
#pragma unroll
  for (int k = 0; k < DD; ++k)
    dq[k] = Disturb<Sys>::inert ? (real)0
                                : -(real)D.tau[k] * (q[k] + (real)D.sigma[k] * (xi[k] + (real)D.mu[k]));
}

// classical RK4 on [state, disturb]; same combination order as rk4_step
template <typename Sys, typename real>
__device__ __forceinline__ void rk4_step_full(const typename Sys::template Pre<real>& pre, const DisturbPars& D, real* x,
                                              real* q, const real* u, const real* xi, real dt) {
  constexpr int DS = Sys::DS, DD = Disturb<Sys>::DD;
  const real h = dt, hh = (real)0.5 * dt, h6 = dt / (real)6;
  real k1[DS], k2[DS], k3[DS], k4[DS], t[DS], l1[DD], l2[DD], l3[DD], l4[DD], s[DD];
  rhs_full<Sys, real>(pre, D, x, q, u, xi, k1, l1);
#pragma unroll
  for (int c = 0; c < DS; ++c) t[c] = fma_r(hh, k1[c], x[c]);
#pragma unroll
  for (int c = 0; c < DD; ++c) s[c] = fma_r(hh, l1[c], q[c]);
  rhs_full<Sys, real>(pre, D, t, s, u, xi, k2, l2);
#pragma unroll
  for (int c = 0; c < DS; ++c) t[c] = fma_r(hh, k2[c], x[c]);
#pragma unroll
  for (int c = 0; c < DD; ++c) s[c] = fma_r(hh, l2[c], q[c]);
  rhs_full<Sys, real>(pre, D, t, s, u, xi, k3, l3);
#pragma unroll
  for (int c = 0; c < DS; ++c) t[c] = fma_r(h, k3[c], x[c]);
#pragma unroll
  for (int c = 0; c < DD; ++c) s[c] = fma_r(h, l3[c], q[c]);
  rhs_full<Sys, real>(pre, D, t, s, u, xi, k4, l4);
  // (the slopes as rounded values before they are combined: rk4_step, rcg_kernels.hpp - one set of bits in k_sim_dist and k_ticks)
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    pin_value(k1[c]);
    pin_value(k2[c]);
    pin_value(k3[c]);
    pin_value(k4[c]);
  }
#pragma unroll
  for (int c = 0; c < DD; ++c) {
    pin_value(l1[c]);
    pin_value(l2[c]);
    pin_value(l3[c]);
    pin_value(l4[c]);
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) x[c] = fma_r(h6, ((k1[c] + (real)2 * k2[c]) + (real)2 * k3[c]) + k4[c], x[c]);
#pragma unroll
  for (int c = 0; c < DD; ++c) q[c] = fma_r(h6, ((l1[c] + (real)2 * l2[c]) + (real)2 * l3[c]) + l4[c], q[c]);
}

template <typename real>
struct SimDistArgs {
  SimArgs<real> S;
  real* disturb;               // [dd][B] in/out
  int32_t* substep_idx;        // [B] in/out
  const int32_t* episode_idx;  // [B]
  DisturbPars D;
};

// Simulator.sim_step x n_sub on the full state [state, disturb] of one env held in registers (shared by k_sim_dist and
// k_ticks): the env_substeps of a handle with RCG_FLAG_DISTURB.  A frozen env is not stepped; a non-finite result freezes
// the env at its last finite state (nothing is updated, the noise counter included) and sets the bit.
template <typename Sys, typename real, bool TGT>
__device__ __forceinline__ bool env_substeps_dist(const KParams<real>& P, const DisturbPars& D,
                                                  const typename Sys::template Pre<real>& pre, int n_sub, int64_t env_id,
                                                  int32_t ep, real* x, real* xp, real* q, int32_t& sub, const real* a_held,
                                                  uint32_t& st, real& accum) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, DD = Disturb<Sys>::DD;
  if (st & 1u) return false;  // frozen env
  real u[DU], xn[DS], xq[DS], qn[DD];
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = P.clip ? clamp_r<real>(a_held[c], P.lo[c], P.hi[c]) : a_held[c];  // systems.py:241-243
#pragma unroll
  for (int c = 0; c < DS; ++c) xq[c] = xn[c] = x[c];
#pragma unroll
  for (int c = 0; c < DD; ++c) qn[c] = q[c];
  int32_t subn = sub;
  real acc = 0;
  for (int s = 0; s < n_sub; ++s) {
#pragma unroll
    for (int c = 0; c < DS; ++c) xq[c] = xn[c];
    double xd[2];
    normals_from_bits(noise_bits(D.seed, env_id, ep, subn), xd);
    const real xi[2] = {(real)xd[0], (real)xd[1]};
    rk4_step_full<Sys, real>(pre, D, xn, qn, u, xi, P.dt_sim);
    subn += 1;
    if (P.accum_every_substep) {
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, xn, u, chi);
      acc = fma_r(stage_any<NCHI, real>(P, chi), P.sampling_time, acc);
    }
  }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < DS; ++c) ok = ok && finite_r<real>(xn[c]);
#pragma unroll
  for (int c = 0; c < DD; ++c) ok = ok && finite_r<real>(qn[c]);
  if (!ok) {
    st |= 1u;
    return false;
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    x[c] = xn[c];
    xp[c] = xq[c];
  }
#pragma unroll
  for (int c = 0; c < DD; ++c) q[c] = qn[c];
  sub = subn;
  if (P.accum_every_substep) accum += acc;
  return true;
}

// k_sim for a handle with RCG_FLAG_DISTURB: same contract, full state [state, disturb]
template <typename Sys, typename real, bool TGT>
__global__ __launch_bounds__(256) void k_sim_dist(const SimDistArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, DD = Disturb<Sys>::DD;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long B = P.B;
  if (b >= B) return;
  uint32_t st = A.S.status[b];
  if (st & 1u) return;  // frozen env

  real x[DS], xp[DS], u[DU], q[DD];
#pragma unroll
  for (int c = 0; c < DS; ++c) xp[c] = x[c] = A.S.state[(long)c * B + b];
#pragma unroll
  for (int c = 0; c < DD; ++c) q[c] = A.disturb[(long)c * B + b];
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = A.S.action[(long)c * B + b];
  const auto pre = load_pre<Sys, real>(P, A.S.pars_env, b);
  int32_t sub = A.substep_idx[b];
  real accum = P.accum_every_substep ? A.S.accum[b] : (real)0;
  if (!env_substeps_dist<Sys, real, TGT>(P, A.D, pre, A.S.n_sub, A.D.env_id_base + b, A.episode_idx[b], x, xp, q, sub, u,
                                        st, accum)) {
    A.S.status[b] = st;  // became non-finite: frozen at its last finite state, nothing else is written
    return;
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    A.S.state[(long)c * B + b] = x[c];
    A.S.state_prev[(long)c * B + b] = xp[c];
  }
#pragma unroll
  for (int c = 0; c < DD; ++c) A.disturb[(long)c * B + b] = q[c];
  A.substep_idx[b] = sub;
  if (P.accum_every_substep) A.S.accum[b] = accum;
}

// unit operator: closed_loop_rhs on the full state for n points, noise given
template <typename Sys, typename real>
__global__ void k_rhs_full(const real* state, const real* disturb, const real* action, const real* xi_in, real* dstate,
                           real* ddisturb, real* clipped, const real* pars_env, long n, int clip, const DisturbPars D,
                           const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, DD = Disturb<Sys>::DD;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real x[DS], q[DD], xi[DD], u[DU], dx[DS], dq[DD];
#pragma unroll
  for (int c = 0; c < DS; ++c) x[c] = state[(long)c * n + i];
#pragma unroll
  for (int c = 0; c < DD; ++c) {
    q[c] = disturb[(long)c * n + i];
    xi[c] = xi_in[(long)c * n + i];
  }
#pragma unroll
  for (int c = 0; c < DU; ++c) {
    const real a = action[(long)c * n + i];
    u[c] = (clip && P.clip) ? clamp_r<real>(a, P.lo[c], P.hi[c]) : a;
    if (clipped) clipped[(long)c * n + i] = u[c];
  }
  const auto pre = load_pre<Sys, real>(P, pars_env, i);
  rhs_full<Sys, real>(pre, D, x, q, u, xi, dx, dq);
#pragma unroll
  for (int c = 0; c < DS; ++c) dstate[(long)c * n + i] = dx[c];
#pragma unroll
  for (int c = 0; c < DD; ++c) ddisturb[(long)c * n + i] = dq[c];
}

template <typename real>
__global__ void k_noise(const int32_t* episode_idx, const int32_t* substep_idx, uint32_t* bits, real* xi, long B,
                        uint64_t seed, int64_t env_id_base) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const PhiloxOut o = noise_bits(seed, env_id_base + b, episode_idx[b], substep_idx[b]);
  if (bits) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bits[(long)k * B + b] = o.w[k];
  }
  if (xi) {
    double xd[2];
    normals_from_bits(o, xd);
    xi[b] = (real)xd[0];
    xi[B + b] = (real)xd[1];
  }
}

}  // namespace rcg
