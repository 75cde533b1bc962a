// rcg_systems.hpp - the three built-in environments as compile-time policies.
//
// Each policy gives the dimensions and the open-loop right-hand side `Sys*._state_dyn` of the
// reference (disturbance-free branch; is_disturb = 0 in every preset).
// `Pre` holds per-env values derived once from the parameter vector (e.g. 1/m), so the rollout's
// inner loop does no divisions.  `rhs<real, HW>`: HW = true selects the rollouts' trig (rcg_math.hpp: f32 hardware
// v_sin / v_cos behind an exact reduction, f64 Cody-Waite + minimax polynomials, 2.3e-16) - every Euler rollout of
// _actor_cost, streamed, generated or inside the optimiser; the simulator's RK4 uses the accurate forms (f64: libm).
#pragma once
#include "rcg_math.hpp"

namespace rcg {

// rcognita/systems.py:308-323  state = (x, y, alpha, v, omega), action = (F, M), pars = (m, I)
struct Sys3WRobot {
  static constexpr int DS = 5, DU = 2, NP = 2;
  static constexpr bool TGT = false;  // observation_target of the preset: [] (k_actor_dma instances exist for this value)
  // chi components whose stage weight is zero in the preset (R1 = diag[1, 10, 1, 0, 0, 0, 0], main_3wrobot.py:147): v, omega,
  // F, M - the generated-candidate rollouts skip their (exactly zero) cost terms when the handle's R1 has them zero too
  static constexpr unsigned ZW_PRESET = 0x78u;
  // state components whose trajectory under the model depends only on (x_0, u[1]) - heading and turn rate follow the
  // torque alone; candidates of the generated grid that share u[1] share them (rollout_mpc_gen_multi, rcg_kernels.hpp)
  static constexpr unsigned SHARED_U1 = (1u << 2) | (1u << 4);
  template <typename real>
  struct Pre {
    real inv_m, inv_I;
  };
  template <typename real>
  __device__ __forceinline__ static Pre<real> prepare(const real* p) {
    return {(real)1 / p[0], (real)1 / p[1]};
  }
  // lane `l`'s Pre in every lane (l wave-uniform): kernels that keep one env per lane and decide with the whole wave
  template <typename real>
  __device__ __forceinline__ static Pre<real> bcast(const Pre<real>& q, int l) {
    return {readlane_r(q.inv_m, l), readlane_r(q.inv_I, l)};
  }
  template <typename real, bool HW = false>
  __device__ __forceinline__ static void rhs(const Pre<real>& q, const real* x, const real* u, real* d) {
    real s, c;
    sincos_sel<real, HW>(x[2], &s, &c);
    d[0] = x[3] * c;
    d[1] = x[3] * s;
    d[2] = x[4];
    d[3] = q.inv_m * u[0];  // 1/m * action[0]
    d[4] = q.inv_I * u[1];  // 1/I * action[1]
  }
  // (A^T lam, B^T lam), A = d f/d x, B = d f/d u at (x, u): the adjoint sweep of k_actor_opt
  template <typename real, bool HW = false>
  __device__ __forceinline__ static void jac_T(const Pre<real>& q, const real* x, const real*, const real* lam,
                                               real* ax, real* bu) {
    real s, c;
    sincos_sel<real, HW>(x[2], &s, &c);
    ax[0] = 0;
    ax[1] = 0;
    ax[2] = x[3] * (lam[1] * c - lam[0] * s);
    ax[3] = lam[0] * c + lam[1] * s;
    ax[4] = lam[2];
    bu[0] = lam[3] * q.inv_m;
    bu[1] = lam[4] * q.inv_I;
  }
};

// rcognita/systems.py:370-382  state = (x, y, alpha), action = (v, omega), no pars
struct Sys3WRobotNI {
  static constexpr int DS = 3, DU = 2, NP = 0;
  static constexpr bool TGT = false;
  static constexpr unsigned ZW_PRESET = 0x18u;  // R1 = diag[1, 10, 1, 0, 0] (main_3wrobot_NI.py): v, omega
  static constexpr unsigned SHARED_U1 = 1u << 2;  // the heading follows omega = u[1] alone
  template <typename real>
  struct Pre {};
  template <typename real>
  __device__ __forceinline__ static Pre<real> prepare(const real*) {
    return {};
  }
  template <typename real>
  __device__ __forceinline__ static Pre<real> bcast(const Pre<real>&, int) {
    return {};
  }
  template <typename real, bool HW = false>
  __device__ __forceinline__ static void rhs(const Pre<real>&, const real* x, const real* u, real* d) {
    real s, c;
    sincos_sel<real, HW>(x[2], &s, &c);
    d[0] = u[0] * c;
    d[1] = u[0] * s;
    d[2] = u[1];
  }
  template <typename real, bool HW = false>
  __device__ __forceinline__ static void jac_T(const Pre<real>&, const real* x, const real* u, const real* lam,
                                               real* ax, real* bu) {
    real s, c;
    sincos_sel<real, HW>(x[2], &s, &c);
    ax[0] = 0;
    ax[1] = 0;
    ax[2] = u[0] * (lam[1] * c - lam[0] * s);
    bu[0] = lam[0] * c + lam[1] * s;
    bu[1] = lam[2];
  }
};

// rcognita/systems.py:412-419  state = (h1, h2), action = (u), pars = (tau1, tau2, K1, K2, K3)
struct Sys2Tank {
  static constexpr int DS = 2, DU = 1, NP = 5;
  static constexpr bool TGT = true;  // main_2tank.py:211: observation_target = [0.5, 0.5]
  static constexpr unsigned ZW_PRESET = 0u;  // R1 = diag[10, 10, 1]: every term counts
  static constexpr unsigned SHARED_U1 = 0;  // one input
  template <typename real>
  struct Pre {
    real inv_tau1, inv_tau2, K1, K2, K3;
  };
  template <typename real>
  __device__ __forceinline__ static Pre<real> prepare(const real* p) {
    return {(real)1 / p[0], (real)1 / p[1], p[2], p[3], p[4]};
  }
  template <typename real>
  __device__ __forceinline__ static Pre<real> bcast(const Pre<real>& q, int l) {
    return {readlane_r(q.inv_tau1, l), readlane_r(q.inv_tau2, l), readlane_r(q.K1, l), readlane_r(q.K2, l), readlane_r(q.K3, l)};
  }
  template <typename real, bool HW = false>
  __device__ __forceinline__ static void rhs(const Pre<real>& q, const real* x, const real* u, real* d) {
    // 1/tau1 (-h1 + K1 u), 1/tau2 (-h2 + K2 h1 + K3 h2^2) (systems.py:416-417) with the multiply-adds written out: under
    // -ffp-contract=fast the compiler otherwise picks the fusions per call site, and the kernels that share this function
    // (k_sim, k_sim_v, k_ticks, the rollouts) must round identically
    // (HW marks the Euler rollouts of _actor_cost: there the expression is left to the compiler, which folds it into the
    // step and the cost accumulation - with the fusions written out the generated-candidate kernels of configs[2] ran
    // 18-26 % slower; a rollout never has to reproduce the simulator's bits, only its own across launches)
    if constexpr (HW) {
      d[0] = q.inv_tau1 * (-x[0] + q.K1 * u[0]);
      d[1] = q.inv_tau2 * (-x[1] + q.K2 * x[0] + q.K3 * (x[1] * x[1]));
    } else {
      d[0] = q.inv_tau1 * fma_r(q.K1, u[0], -x[0]);
      d[1] = q.inv_tau2 * fma_r(q.K3, x[1] * x[1], fma_r(q.K2, x[0], -x[1]));
    }
  }
  template <typename real, bool HW = false>
  __device__ __forceinline__ static void jac_T(const Pre<real>& q, const real* x, const real*, const real* lam,
                                               real* ax, real* bu) {
    ax[0] = -lam[0] * q.inv_tau1 + lam[1] * q.inv_tau2 * q.K2;
    ax[1] = lam[1] * q.inv_tau2 * ((real)-1 + (real)2 * q.K3 * x[1]);
    bu[0] = lam[0] * q.inv_tau1 * q.K1;
  }
};

}  // namespace rcg
