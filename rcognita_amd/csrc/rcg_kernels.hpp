// rcg_kernels.hpp - HIP kernels of the rcognita hot path for gfx950 (CDNA4, wave64).
//
// Data layout in HBM: struct-of-arrays with the env index innermost ("[d][B]": component c of env
// b at c*B + b), so that with lane == env every load/store of a wave is one contiguous 256 B (f32)
// segment.  Candidate action sequences are [B][K][N][du]: one candidate is one contiguous row of
// R = N*du reals, the rows of one env are contiguous, the envs follow each other.
//
// Kernels
//   k_actor   _actor_cost for K candidates per env + wave argmin (+ tick epilogue).  A wave owns a
//             contiguous tile of <= 64 candidate rows (one env, or 64/Kp envs when K < 64); the
//             tile is pulled from HBM with coalesced 16 B/lane loads, parked in LDS, and each lane
//             then walks ITS row step by step while unrolling the Euler rollout in registers.
//             The running cost is a register; the argmin over candidates is a wave butterfly.
//   k_sim     closed_loop_rhs + fixed-step RK4, lane == env.
//   k_rhs / k_stage_obj / k_critic / k_critic_cost   the reference's operators, lane == point,
//             for unit parity.
//   k_episode_reset, k_stats                         bookkeeping (push_vec of the critic buffers: rcg_critic_fit.hpp).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "../../include/rcg.h"
#include "rcg_systems.hpp"

namespace rcg {

enum : int { STAGE_FULL = 1, STAGE_BIQUAD = 2 };

// Timing-only switches of k_actor_dma (they skip work, so results are WRONG): compiled in only with -DRCG_DEV
// (`make dev` -> librcg_dev.so, used by tools/), constant 0 in the production library.
#ifdef RCG_DEV
#define RCG_DBG(A, bit) ((A).dbg & (bit))
#else
#define RCG_DBG(A, bit) 0
#endif

// Wave-uniform parameters, passed by value in the kernarg segment (=> scalar loads / SGPRs).
// Kept small on purpose: the two full n x n stage-cost matrices live in a device buffer (`Rfull`)
// that only the non-diagonal path reads; with them inline the struct overflowed the SGPR file and
// every kernel spilled scalars into VGPRs.
template <typename real>
struct KParams {
  real pars[RCG_MAX_PARS];
  real lo[RCG_MAX_DU], hi[RCG_MAX_DU];
  const real* __restrict__ Rfull;           // [2][49]: R1 then R2, row-major, leading dimension n
  real R1d[RCG_MAX_CHI], R2d[RCG_MAX_CHI];  // diagonals (used when !(stage_kind & STAGE_FULL))
  real target[RCG_MAX_DS];
  real gamma, h_pred, dt_sim, sampling_time;
  int B, n_actor, mode, critic_struct, dc, n_critic, buffer_size;
  int stage_kind, has_target, clip, per_env_pars, ref_lag, accum_every_substep;
  unsigned zero_w;  // bit i: R1_ii == 0 (diagonal stage cost only): that component's cost term is exactly zero
};

// ---------------------------------------------------------------------------------------------
// chi = [obs - target, act]   (controllers.py:1069-1072, 1200-1203)
template <int DS, int DU, bool TGT, typename real>
__device__ __forceinline__ void make_chi(const KParams<real>& P, const real* y, const real* u, real* chi) {
#pragma unroll
  for (int i = 0; i < DS; ++i) chi[i] = TGT ? y[i] - P.target[i] : y[i];
#pragma unroll
  for (int c = 0; c < DU; ++c) chi[DS + c] = u[c];
}

// stage_obj on a ready chi: diagonal R1 only (the presets' case, main_3wrobot.py:177)
template <int NCHI, typename real>
__device__ __forceinline__ real stage_diag(const KParams<real>& P, const real* chi) {
  real q = 0;
#pragma unroll
  for (int i = 0; i < NCHI; ++i) q = fma_r(P.R1d[i] * chi[i], chi[i], q);
  return q;
}

// A full n x n stage matrix as its symmetrised upper triangle, row-major (i <= j): S_ii = R_ii, S_ij = R_ij + R_ji.
// chi @ R @ chi = sum_i chi_i (S_ii chi_i + sum_{j > i} S_ij chi_j): n (n + 1) / 2 + n fused multiply-adds instead of n^2 + n
// (n = 7: 35 instead of 56), for ANY R (the reference takes whatever matrix it is handed; a non-symmetric one is covered by
// fixture F2's `quad_nonsym`).  Every kernel evaluates a full-matrix stage cost through these two functions, so the streamed
// production instance (k_actor_dma, DMA_MPC_GENF - the triangle held in registers across the tile loop) and the kernels that
// re-read the matrix (k_actor, k_ticks, k_sim ...) leave the same bits.
template <int NCHI>
constexpr int sym_len() { return NCHI * (NCHI + 1) / 2; }
template <int NCHI, typename real>
__device__ __forceinline__ void load_sym(const real* __restrict__ R, real* Sq) {
  int idx = 0;
#pragma unroll
  for (int i = 0; i < NCHI; ++i)
#pragma unroll
    for (int j = i; j < NCHI; ++j) {
      Sq[idx] = i == j ? R[i * NCHI + i] : R[i * NCHI + j] + R[j * NCHI + i];
      ++idx;
    }
}
template <int NCHI, typename real>
__device__ __forceinline__ real quad_sym(const real* Sq, const real* c) {
  real q = 0;
  int idx = 0;
#pragma unroll
  for (int i = 0; i < NCHI; ++i) {
    real v = Sq[idx] * c[i];
    ++idx;
#pragma unroll
    for (int j = i + 1; j < NCHI; ++j) {
      v = fma_r(Sq[idx], c[j], v);
      ++idx;
    }
    q = fma_r(c[i], v, q);
  }
  return q;
}

// stage_obj, every structure (controllers.py:1076-1082):
//   quadratic   chi @ R1 @ chi            biquadratic   chi**2 @ R2 @ chi**2 + chi @ R1 @ chi
// `sk` = P.stage_kind, or a compile-time constant when the caller has already dispatched on it
template <int NCHI, typename real>
__device__ __forceinline__ real stage_with(const KParams<real>& P, const real* chi, const int sk) {
  real q;
  if (!(sk & STAGE_FULL)) {
    q = stage_diag<NCHI, real>(P, chi);
    if (sk & STAGE_BIQUAD) {
      real q4 = 0;
#pragma unroll
      for (int i = 0; i < NCHI; ++i) {
        const real c2 = chi[i] * chi[i];
        q4 = fma_r(P.R2d[i] * c2, c2, q4);
      }
      q = q4 + q;
    }
  } else {
    real Sq[sym_len<NCHI>()];
    load_sym<NCHI, real>(P.Rfull, Sq);
    q = quad_sym<NCHI, real>(Sq, chi);
    if (sk & STAGE_BIQUAD) {
      real c2[NCHI];
#pragma unroll
      for (int i = 0; i < NCHI; ++i) c2[i] = chi[i] * chi[i];
      load_sym<NCHI, real>(P.Rfull + 49, Sq);
      const real q4 = quad_sym<NCHI, real>(Sq, c2);
      q = q4 + q;
    }
  }
  return q;
}
template <int NCHI, typename real>
__device__ __forceinline__ real stage_any(const KParams<real>& P, const real* chi) {
  return stage_with<NCHI, real>(P, chi, P.stage_kind);
}

// _critic = w @ regressor (controllers.py:1192-1214).  `w(i)` returns weight i of this lane's env.
// chi already holds [obs - target, act]; quad-mix uses the RAW observation y (controllers.py:1212).
// `cs` = P.critic_struct, or a compile-time constant when the caller has already dispatched on it.
template <int DS, int DU, typename real, typename WGet>
__device__ __forceinline__ real critic_with(const real* chi, const real* y, const real* u, WGet w, const int cs) {
  constexpr int NCHI = DS + DU;
  real acc = 0;
  if (cs == RCG_CRITIC_QUAD_LIN || cs == RCG_CRITIC_QUADRATIC) {
    int idx = 0;  // uptria2vec: row-major upper triangle incl. diagonal (utilities.py:81-96)
#pragma unroll
    for (int i = 0; i < NCHI; ++i)
#pragma unroll
      for (int j = i; j < NCHI; ++j) acc = fma_r(w(idx++), chi[i] * chi[j], acc);
    if (cs == RCG_CRITIC_QUAD_LIN) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) acc = fma_r(w(idx++), chi[i], acc);
    }
  } else if (cs == RCG_CRITIC_QUAD_NOMIX) {
#pragma unroll
    for (int i = 0; i < NCHI; ++i) acc = fma_r(w(i), chi[i] * chi[i], acc);
  } else {  // quad-mix: [obs**2, kron(obs, act), act**2]
#pragma unroll
    for (int i = 0; i < DS; ++i) acc = fma_r(w(i), y[i] * y[i], acc);
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int c = 0; c < DU; ++c) acc = fma_r(w(DS + i * DU + c), y[i] * u[c], acc);
#pragma unroll
    for (int c = 0; c < DU; ++c) acc = fma_r(w(DS + DS * DU + c), u[c] * u[c], acc);
  }
  return acc;
}
// SQL sums the critic over the horizon, J = sum_k w . phi(chi_k) = w . sum_k phi(chi_k): the regressor is accumulated
// per feature (one fma per feature and step) and dotted with the weights once, instead of a product AND an fma per
// feature at every step.  Feature order = critic_with's.  `cs` must be a compile-time constant at the call site.
template <int DS, int DU>
__host__ __device__ constexpr int critic_dim(int cs) {
  return cs == RCG_CRITIC_QUAD_LIN ? (DS + DU) * (DS + DU + 1) / 2 + (DS + DU)
                                   : (cs == RCG_CRITIC_QUADRATIC ? (DS + DU) * (DS + DU + 1) / 2
                                                                 : (cs == RCG_CRITIC_QUAD_NOMIX ? DS + DU : DS + DS * DU + DU));
}
template <int DS, int DU, typename real>
__device__ __forceinline__ void critic_phi_accum(const real* chi, const real* y, const real* u, real* Phi, const int cs) {
  constexpr int NCHI = DS + DU;
  if (cs == RCG_CRITIC_QUAD_LIN || cs == RCG_CRITIC_QUADRATIC) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < NCHI; ++i)
#pragma unroll
      for (int j = i; j < NCHI; ++j) {
        Phi[idx] = fma_r(chi[i], chi[j], Phi[idx]);
        ++idx;
      }
    if (cs == RCG_CRITIC_QUAD_LIN) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) {
        Phi[idx] += chi[i];
        ++idx;
      }
    }
  } else if (cs == RCG_CRITIC_QUAD_NOMIX) {
#pragma unroll
    for (int i = 0; i < NCHI; ++i) Phi[i] = fma_r(chi[i], chi[i], Phi[i]);
  } else {  // quad-mix: [obs**2, kron(obs, act), act**2] on the RAW observation (controllers.py:1212)
#pragma unroll
    for (int i = 0; i < DS; ++i) Phi[i] = fma_r(y[i], y[i], Phi[i]);
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int c = 0; c < DU; ++c) Phi[DS + i * DU + c] = fma_r(y[i], u[c], Phi[DS + i * DU + c]);
#pragma unroll
    for (int c = 0; c < DU; ++c) Phi[DS + DS * DU + c] = fma_r(u[c], u[c], Phi[DS + DS * DU + c]);
  }
}

template <int DS, int DU, typename real, typename WGet>
__device__ __forceinline__ real critic_value(const KParams<real>& P, const real* chi, const real* y,
                                             const real* u, WGet w) {
  return critic_with<DS, DU, real>(chi, y, u, w, P.critic_struct);
}

template <typename Sys, typename real>
__device__ __forceinline__ typename Sys::template Pre<real> load_pre(const KParams<real>& P,
                                                                     const real* pars_env, long b) {
  real pv[Sys::NP > 0 ? Sys::NP : 1];
#pragma unroll
  for (int i = 0; i < Sys::NP; ++i) pv[i] = pars_env ? pars_env[(long)i * P.B + b] : P.pars[i];
  return Sys::template prepare<real>(pv);
}

// One classical RK4 step of the closed loop under a held (already clipped) action u: the build's
// the value as it stands in a register: an empty asm the optimiser cannot see through (no instruction is emitted)
__device__ __forceinline__ void pin_value(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_value(double& v) { asm volatile("" : "+v"(v)); }

// replacement of scipy RK45 inside Simulator.sim_step (simulator.py:156-168).  Same operation order as
// oracle rk4_step: x + h/6 * (((k1 + 2 k2) + 2 k3) + k4).
template <typename Sys, typename real>
__device__ __forceinline__ void rk4_step(const typename Sys::template Pre<real>& pre, real* x, const real* u,
                                         real dt) {
  constexpr int DS = Sys::DS;
  const real h = dt, hh = (real)0.5 * dt, h6 = dt / (real)6;
  real k1[DS], k2[DS], k3[DS], k4[DS], t[DS];
  Sys::template rhs<real>(pre, x, u, k1);
#pragma unroll
  for (int c = 0; c < DS; ++c) t[c] = fma_r(hh, k1[c], x[c]);
  Sys::template rhs<real>(pre, t, u, k2);
#pragma unroll
  for (int c = 0; c < DS; ++c) t[c] = fma_r(hh, k2[c], x[c]);
  Sys::template rhs<real>(pre, t, u, k3);
#pragma unroll
  for (int c = 0; c < DS; ++c) t[c] = fma_r(h, k3[c], x[c]);
  Sys::template rhs<real>(pre, t, u, k4);
  // k4 is used once, in the sum below, and a right-hand side usually ends in a multiply (1 / tau * (...), v * cos): under
  // -ffp-contract=fast the compiler is free to fold that multiply into the "+ k4" - it did inside k_actor_dma_packed's fused env
  // step and did not inside k_sim / k_ticks, and one state component in a thousand came out one ulp apart (found by
  // tools/fuzz_parity.py, round 6; a contract(off) pragma does not reach the backend's fusion in this mode).  Every kernel that
  // steps an env must produce the same bits: the slopes are pinned as rounded values before they are combined.
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    pin_value(k1[c]);
    pin_value(k2[c]);
    pin_value(k3[c]);
    pin_value(k4[c]);
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) x[c] = fma_r(h6, ((k1[c] + (real)2 * k2[c]) + (real)2 * k3[c]) + k4[c], x[c]);
}

// ---------------------------------------------------------------------------------------------
// k_actor
// ---------------------------------------------------------------------------------------------
template <typename real>
struct ActorArgs {
  const real* cand;       // [B][K][N][du], or nullptr: generated level grid
  const real* obs;        // [dy][B]
  const real* state_sys;  // [ds][B]
  const real* pars_env;   // [np][B] or nullptr
  const real* w;          // [dc][B] (RQL/SQL)
  real* J;                // [B][K] or nullptr
  real* action_out;       // [du][B] or nullptr
  real* best_J;           // [B] or nullptr
  int32_t* best_idx;      // [B] or nullptr
  real* accum;            // tick epilogue: accum += rho(obs, action) * sampling_time; or nullptr
  int32_t* step_idx;      // tick epilogue: += 1; or nullptr
  int K;                  // candidates per env
  int Kp;                 // K rounded up to a power of two (K < 64), else 64
  int G;                  // envs per wave: 64 / Kp (K < 64), else 1
  int n_tiles;            // ceil(K / 64) (K >= 64), else 1
  int grid_g;             // generated grid: levels per input
  int vec_ok;             // rows are 16-B granular: stage with dwordx4
  int gpw;                // k_actor_dma: consecutive envs per (persistent) wave
  int jwave;              // k_actor_dma, J output: stage the costs of all envs of the wave in LDS (else env by env)
  int no_multi;           // development (env RCG_NO_GEN_MULTI): generated tiles one at a time (A/B of rollout_mpc_gen_multi)
  int dbg;                // -DRCG_DEV builds only (env RCG_DBG): bits skip parts of k_actor_dma for timing
  int env_lo, env_hi;     // k_actor_dma: the envs [env_lo, env_hi) of the batch this launch serves (env_hi == 0: all of them) -
                          // a handle that splits its tick into halves on two streams (rcg_control_tick, RQL / SQL)
  // k_actor_dma_packed, MPC tick: the env step of the tick (Simulator.sim_step, k_sim's code) fused into the launch -
  // sim_n_sub > 0: the wave steps ITS envs first (lane == env), writes STATE / STATE_PREV / STATUS and hands the new states to
  // its tiles through LDS; obs / state_sys are then not read
  real* sim_state;        // [ds][B] in/out
  real* sim_state_prev;   // [ds][B] out
  const real* sim_action; // [du][B] the held action
  uint32_t* sim_status;   // [B]
  int sim_n_sub;
};

typedef float v4f __attribute__((ext_vector_type(4)));  // one 16-B global_load_dwordx4 / ds_write_b128

// Pull `n` reals (a whole tile of candidate rows, contiguous in HBM) into this wave's LDS region.
// Control flow is wave-uniform (scalar compares): full 1-KiB rows of 64 lanes x 16 B, up to 8 of them
// (8 KiB per wave) in flight before the first LDS write, then one partial row.
template <typename real>
__device__ __forceinline__ void stage_tile(const real* __restrict__ g, real* __restrict__ l, int n,
                                           int lane, int vec_ok) {
  if (vec_ok) {
    const int nv = (int)(((long)n * (long)sizeof(real)) >> 4);
    const v4f* __restrict__ gv = reinterpret_cast<const v4f*>(g);
    v4f* __restrict__ lv = reinterpret_cast<v4f*>(l);
    const int nfull = nv >> 6;
    for (int base = 0; base < nfull; base += 8) {
      const int cnt = nfull - base;  // wave-uniform
      v4f r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cnt) r[j] = __builtin_nontemporal_load(&gv[(base + j) * 64 + lane]);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cnt) lv[(base + j) * 64 + lane] = r[j];
    }
    const int i = nfull * 64 + lane;
    if (i < nv) lv[i] = __builtin_nontemporal_load(&gv[i]);
  } else {
    for (int i = lane; i < n; i += 64) l[i] = g[i];
  }
}

// ---- pieces shared by k_actor and k_ticks (same source => the persistent multi-tick kernel reproduces a sequence of
// single ticks bit for bit; everything that could be contracted differently is an explicit fma) -------------------------
// generated candidates: constant-over-horizon level grid (du = 1: K levels; du = 2: k -> (k / g, k % g))
template <int DU, typename real>
__device__ __forceinline__ void gen_candidate(const KParams<real>& P, int g, int k, real* ugen) {
  const real den = (real)(g > 1 ? g - 1 : 1);
  if (DU == 1) {
    ugen[0] = fma_r((real)k, (P.hi[0] - P.lo[0]) / den, P.lo[0]);
  } else {
    const int gi = k / g, gj = k - gi * g;
    ugen[0] = fma_r((real)gi, (P.hi[0] - P.lo[0]) / den, P.lo[0]);
    ugen[DU - 1] = fma_r((real)gj, (P.hi[DU - 1] - P.lo[DU - 1]) / den, P.lo[DU - 1]);
  }
}

// _actor_cost: explicit-Euler rollout + running cost (controllers.py:1284-1326) of ONE candidate: the row `urow`
// (STREAM, step-major in LDS) or the constant sequence `ugen`.  MODE_C / SK_C / CS_C are compile-time values of mode /
// stage_kind / critic_struct, -1 = read the runtime value: the caller dispatches once per tile, so the step loop
// carries no mode / stage-structure / critic-structure branches (they cost more than the ~14 VALU ops of a 2tank step).
// G1 (MPC, diagonal quadratic stage cost, gamma == 1 - every preset): the sum of weighted squares is accumulated per
// component, S_i += chi_i^2, and weighted once at the end, J = sum_i R1_ii S_i - 7 fma per 3wrobot step instead of 14
// ops + the discount bookkeeping, in a rollout of ~33 (the generated-candidate regime is VALU-issue-bound,
// profiles/r02_*_valu_pmc.json; k_actor_dma has had the same variant since round 1).
// NC > 0 (k_actor_search): the horizon is the compile-time constant NC and the loop is fully unrolled, so a row handed over as
// a REGISTER array (urow[kk * DU + c] with static indices) never touches memory.
template <typename Sys, typename real, bool TGT, bool STREAM, int MODE_C, int SK_C, int CS_C, bool G1 = false, unsigned ZW = 0u,
          int NC = 0, typename WGet>
__device__ __forceinline__ real rollout_cost(const KParams<real>& P, const typename Sys::template Pre<real>& pre, int N_rt,
                                             const real* xs, const real* y0, const real* urow, const real* ugen,
                                             WGet wget, real* u0) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const int N = NC > 0 ? NC : N_rt;
  const int mode = MODE_C >= 0 ? MODE_C : P.mode;
  const int sk = SK_C >= 0 ? SK_C : P.stage_kind;
  const int cs = CS_C >= 0 ? CS_C : P.critic_struct;
  const real h = P.h_pred;
  real x[DS], y[DS];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    x[c] = xs[c];
    y[c] = y0[c];
  }
  static_assert(!G1 || ((MODE_C == RCG_MODE_MPC || MODE_C == RCG_MODE_RQL) && SK_C == 0),
                "the per-component sum is a variant of the diagonal-R1 stage sums (MPC: all N steps, RQL: the first N - 1)");
  real J = 0, gk = 1;
  real u[DU], up[DU];
  real S[G1 ? NCHI : 1];
#pragma unroll
  for (int i = 0; i < (G1 ? NCHI : 1); ++i) S[i] = 0;
  constexpr bool SUMF = MODE_C == RCG_MODE_SQL && CS_C >= 0;  // SQL: per-feature sums (critic_phi_accum)
  constexpr int NPHI = SUMF ? critic_dim<DS, DU>(CS_C >= 0 ? CS_C : 0) : 1;
  real Phi[NPHI];
#pragma unroll
  for (int i = 0; i < NPHI; ++i) Phi[i] = 0;
#pragma unroll
  for (int c = 0; c < DU; ++c) up[c] = 0;
  real zw0 = 0;  // zero-weighted state components of the observation: 0, or NaN if one of them is not finite (see the end)
#pragma unroll
  for (int i = 0; i < DS; ++i)
    if (G1 && ((ZW >> i) & 1u)) zw0 = fma_r(P.R1d[i], y0[i] * y0[i], zw0);
// One step kk of the rollout, expanded below in the runtime-horizon loop and in the fully unrolled compile-time one (a macro,
// not a lambda: wrapping the body in a lambda changed one fma contraction in one instance, and T ticks in one launch must
// stay bit-identical to T single ticks).  The Euler step is unclipped, as sys_rhs([], state, u[k-1]); f32: hardware
// v_sin / v_cos behind the exact reduction, as in k_actor_dma (3.7e-7 max abs error, 7 VALU ops instead of ~25).  G1: per-
// component sums (ZW: weight exactly zero, term skipped); RQL's last step is Q_w(y_{N-1}, u_{N-1}) (controllers.py:1310); SQL
// with a compile-time structure sums the regressor per feature (critic_phi_accum).
#define RCG_ROLLOUT_STEP                                                                                             \
_Pragma("unroll")                                                                                                        \
    for (int c = 0; c < DU; ++c) u[c] = STREAM ? urow[kk * DU + c] : ugen[c];                                         \
    if (kk == 0) {                                                                                                    \
_Pragma("unroll")                                                                                                        \
      for (int c = 0; c < DU; ++c) u0[c] = u[c];                                                                      \
    } else {                                                                                                          \
      real d[DS];                                                                                                     \
                                                                                                                      \
                                                                                                                      \
      Sys::template rhs<real, true>(pre, x, up, d);                                                                   \
_Pragma("unroll")                                                                                                        \
      for (int c = 0; c < DS; ++c) {                                                                                  \
        x[c] = fma_r(h, d[c], x[c]);                                                                                  \
        y[c] = x[c];                                                                                                  \
      }                                                                                                               \
    }                                                                                                                 \
    real chi[NCHI];                                                                                                   \
    make_chi<DS, DU, TGT, real>(P, y, u, chi);                                                                        \
    if (G1 && (MODE_C == RCG_MODE_MPC || kk < N - 1)) {                                                               \
_Pragma("unroll")                                                                                                        \
      for (int i = 0; i < NCHI; ++i)                                                                                  \
        if (!((ZW >> i) & 1u)) S[G1 ? i : 0] = fma_r(chi[i], chi[i], S[G1 ? i : 0]);                                  \
    } else if (G1) {                                                                                                  \
      J += critic_with<DS, DU, real>(chi, y, u, wget, cs);                                                            \
    } else if (mode == RCG_MODE_MPC) {                                                                                \
      J = fma_r(gk, stage_with<NCHI, real>(P, chi, sk), J);                                                           \
    } else if (mode == RCG_MODE_RQL) {                                                                                \
      if (kk < N - 1)                                                                                                 \
        J = fma_r(gk, stage_with<NCHI, real>(P, chi, sk), J);                                                         \
      else                                                                                                            \
        J += critic_with<DS, DU, real>(chi, y, u, wget, cs);                                                          \
    } else if (SUMF) {                                                                                                \
      critic_phi_accum<DS, DU, real>(chi, y, u, Phi, CS_C);                                                           \
    } else {                                                                                                          \
      J += critic_with<DS, DU, real>(chi, y, u, wget, cs);                                                            \
    }                                                                                                                 \
    if (!G1) gk *= P.gamma;                                                                                           \
_Pragma("unroll")                                                                                                        \
    for (int c = 0; c < DU; ++c) up[c] = u[c];                                                                        \
                                                                                                                      \
  /* end of RCG_ROLLOUT_STEP */
  if constexpr (NC > 0) {
#pragma unroll
    for (int kk = 0; kk < NC; ++kk) {
      RCG_ROLLOUT_STEP
    }
  } else {
    for (int kk = 0; kk < N; ++kk) {
      RCG_ROLLOUT_STEP
    }
  }
#undef RCG_ROLLOUT_STEP
  if (G1) {
#pragma unroll
    for (int i = 0; i < NCHI; ++i)
      if (!((ZW >> i) & 1u)) J = fma_r(P.R1d[i], S[G1 ? i : 0], J);  // fma(0, S_i, J) == J for finite S_i
    // ... and NaN for a non-finite S_i (0 * inf), which is what numpy's chi R1 chi gives the reference: the skipped state
    // components are tested once, on the observation (zw0, formed before the loop) and on the last rolled-out state
    // (inf / NaN is sticky under x += h f), so that a component that overflows under a zero weight disqualifies the
    // candidate here as it does in the streamed kernels.  (The skipped inputs are the grid's own bounded levels.)
#ifndef RCG_AB_NO_ZWP
#pragma unroll
    for (int i = 0; i < DS; ++i)
      if ((ZW >> i) & 1u) J = fma_r(P.R1d[i], y[i] * y[i], J);
    J += zw0;  // (exactly 0, or NaN: J is unchanged bit for bit whenever every zero-weighted component is finite)
#endif
  }
  if (SUMF) {
#pragma unroll
    for (int i = 0; i < NPHI; ++i) J = fma_r(wget(i), Phi[i], J);
  }
  return J;
}

// Once per tile: pick the specialisation of rollout_cost for this handle's (mode, stage structure, critic structure).
template <typename Sys, typename real, bool GENERIC, bool TGT, bool STREAM, int NC = 0, typename WGet>
__device__ __forceinline__ real rollout_dispatch(const KParams<real>& P, const typename Sys::template Pre<real>& pre,
                                                 int N, const real* xs, const real* y0, const real* urow,
                                                 const real* ugen, WGet wget, real* u0) {
#define RCG_ROLL(M, S, C) rollout_cost<Sys, real, TGT, STREAM, M, S, C>(P, pre, N, xs, y0, urow, ugen, wget, u0)
  if constexpr (!GENERIC) {  // MPC, quadratic, diagonal R1
    if (P.gamma == (real)1) {  // wave-uniform
      // the preset's zero stage weights (Sys::ZW_PRESET) are zero in this handle too: their terms - exact zeros - are
      // not computed (the generated-candidate regime is bound by instruction issue; streamed rollouts hide them anyway)
      // (NC > 0: k_actor_search's register rows - generated inside the bounds, so the skipped inputs are finite there too)
      if ((!STREAM || NC > 0) && Sys::ZW_PRESET != 0u && (P.zero_w & Sys::ZW_PRESET) == Sys::ZW_PRESET)
        return rollout_cost<Sys, real, TGT, STREAM, RCG_MODE_MPC, 0, -1, true, Sys::ZW_PRESET, NC>(P, pre, N, xs, y0, urow,
                                                                                                 ugen, wget, u0);
      return rollout_cost<Sys, real, TGT, STREAM, RCG_MODE_MPC, 0, -1, true, 0u, NC>(P, pre, N, xs, y0, urow, ugen, wget, u0);
    }
    return rollout_cost<Sys, real, TGT, STREAM, RCG_MODE_MPC, 0, -1, false, 0u, NC>(P, pre, N, xs, y0, urow, ugen, wget, u0);
  } else {
  if (P.mode == RCG_MODE_MPC) return RCG_ROLL(RCG_MODE_MPC, -1, -1);
  if (P.mode == RCG_MODE_RQL && P.stage_kind == 0) {
#define RCG_ROLL_G1(C) rollout_cost<Sys, real, TGT, STREAM, RCG_MODE_RQL, 0, C, true>(P, pre, N, xs, y0, urow, ugen, wget, u0)
    // (generated candidates only: a STREAMED row is accumulated as k_actor_dma / k_actor_dma_packed accumulate it - gamma^k rho_k
    // step by step - so that the decision phase of k_ticks_mem over a caller's tensor, which is this code, leaves the bits the
    // single ticks on those kernels leave: round 5)
    if (P.gamma == (real)1 && !STREAM) {  // wave-uniform: per-component stage sums over the first N - 1 steps
      switch (P.critic_struct) {
        case RCG_CRITIC_QUAD_LIN: return RCG_ROLL_G1(RCG_CRITIC_QUAD_LIN);
        case RCG_CRITIC_QUADRATIC: return RCG_ROLL_G1(RCG_CRITIC_QUADRATIC);
        case RCG_CRITIC_QUAD_NOMIX: return RCG_ROLL_G1(RCG_CRITIC_QUAD_NOMIX);
        default: return RCG_ROLL_G1(RCG_CRITIC_QUAD_MIX);
      }
    }
#undef RCG_ROLL_G1
    switch (P.critic_struct) {
      case RCG_CRITIC_QUAD_LIN: return RCG_ROLL(RCG_MODE_RQL, 0, RCG_CRITIC_QUAD_LIN);
      case RCG_CRITIC_QUADRATIC: return RCG_ROLL(RCG_MODE_RQL, 0, RCG_CRITIC_QUADRATIC);
      case RCG_CRITIC_QUAD_NOMIX: return RCG_ROLL(RCG_MODE_RQL, 0, RCG_CRITIC_QUAD_NOMIX);
      default: return RCG_ROLL(RCG_MODE_RQL, 0, RCG_CRITIC_QUAD_MIX);
    }
  }
  if (P.mode == RCG_MODE_SQL) {  // no stage cost inside the SQL sum
    switch (P.critic_struct) {
      case RCG_CRITIC_QUAD_LIN: return RCG_ROLL(RCG_MODE_SQL, -1, RCG_CRITIC_QUAD_LIN);
      case RCG_CRITIC_QUADRATIC: return RCG_ROLL(RCG_MODE_SQL, -1, RCG_CRITIC_QUADRATIC);
      case RCG_CRITIC_QUAD_NOMIX: return RCG_ROLL(RCG_MODE_SQL, -1, RCG_CRITIC_QUAD_NOMIX);
      default: return RCG_ROLL(RCG_MODE_SQL, -1, RCG_CRITIC_QUAD_MIX);
    }
  }
  return RCG_ROLL(-1, -1, -1);  // RQL with a full-matrix / biquadratic stage cost
  }
#undef RCG_ROLL
}

// argmin over a segment of `seg` lanes (a power of two): lower J wins, ties -> lower candidate index; every lane of
// the segment ends up with the winner
template <int DU, typename real>
__device__ __forceinline__ void segment_argmin(int seg, real& bestJ, int& bestI, real* bestU) {
  for (int m = 1; m < seg; m <<= 1) {
    const real oJ = __shfl_xor(bestJ, m, 64);
    const int oI = __shfl_xor(bestI, m, 64);
    real oU[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) oU[c] = __shfl_xor(bestU[c], m, 64);
    const bool take = (oJ < bestJ) || (oJ == bestJ && oI < bestI);
    if (take) {
      bestJ = oJ;
      bestI = oI;
#pragma unroll
      for (int c = 0; c < DU; ++c) bestU[c] = oU[c];
    }
  }
}

// NC generated candidates of ONE lane (tiles t .. t + NC - 1 of the level grid: same second input, NC different first
// inputs) rolled out together, MPC with a diagonal quadratic stage cost.  Every candidate goes through exactly the
// operation sequence of rollout_cost<MPC, 0, -1, G1> - its cost is the same bits - but the state components that depend
// only on (x_0, u[1]) (Sys::SHARED_U1: heading and turn rate of the robots) and their cost terms are THE SAME VALUES for
// all NC candidates; after each step they are taken from candidate 0, which makes them the same SSA values, and the
// compiler's common-subexpression elimination then keeps one sin/cos, one heading update and one alpha^2 / omega^2 / M^2
// accumulation per lane and step instead of NC.  The generated-candidate regime is bound by VALU instruction issue
// (profiles/r02_valu_pmc.json: > 100 % of the 4-cycle issue slots), so instructions are the currency: 3wrobot, NC = 4,
// gamma = 1: ~13 instead of 27 per candidate-step.
template <typename Sys, typename real, bool TGT, bool G1, int NC, unsigned ZW = 0u>
__device__ __forceinline__ void rollout_mpc_gen_multi(const KParams<real>& P, const typename Sys::template Pre<real>& pre,
                                                      int N, const real* xs, const real* y0, const real* u0v, real u1,
                                                      real* Jout) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  constexpr unsigned SH = Sys::SHARED_U1;
  static_assert(DU == 2, "candidates of a lane share their second input");
  const real h = P.h_pred;
  real x[NC][DS], y[NC][DS], u[NC][DU], J[NC], S[NC][G1 ? NCHI : 1];
  real gk = 1;
  real zw0 = 0;  // as rollout_cost
#pragma unroll
  for (int i = 0; i < DS; ++i)
    if (G1 && ((ZW >> i) & 1u)) zw0 = fma_r(P.R1d[i], y0[i] * y0[i], zw0);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      x[c][i] = xs[i];
      y[c][i] = y0[i];
    }
    u[c][0] = u0v[c];
    u[c][1] = u1;
    J[c] = 0;
#pragma unroll
    for (int i = 0; i < (G1 ? NCHI : 1); ++i) S[c][i] = 0;
  }
  for (int kk = 0; kk < N; ++kk) {
    if (kk > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        real d[DS];
        Sys::template rhs<real, true>(pre, x[c], u[c], d);  // unclipped (controllers.py:1294)
#pragma unroll
        for (int i = 0; i < DS; ++i) {
          x[c][i] = fma_r(h, d[i], x[c][i]);
          y[c][i] = x[c][i];  // sys_out is the identity
        }
      }
#pragma unroll
      for (int c = 1; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < DS; ++i)
          if ((SH >> i) & 1u) {  // identical by construction: name them identically
            x[c][i] = x[0][i];
            y[c][i] = y[0][i];
          }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, y[c], u[c], chi);
      if (G1) {
#pragma unroll
        for (int i = 0; i < NCHI; ++i)
          if (!((ZW >> i) & 1u)) S[c][G1 ? i : 0] = fma_r(chi[i], chi[i], S[c][G1 ? i : 0]);
      } else {
        J[c] = fma_r(gk, stage_diag<NCHI, real>(P, chi), J[c]);
      }
    }
    if (G1) {
#pragma unroll
      for (int c = 1; c < NC; ++c) {
#pragma unroll
        for (int i = 0; i < DS; ++i)
          if ((SH >> i) & 1u) S[c][G1 ? i : 0] = S[0][G1 ? i : 0];
        S[c][G1 ? DS + 1 : 0] = S[0][G1 ? DS + 1 : 0];  // the shared input's own term
      }
    } else {
      gk *= P.gamma;
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (G1) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i)
        if (!((ZW >> i) & 1u)) J[c] = fma_r(P.R1d[i], S[c][G1 ? i : 0], J[c]);
#ifndef RCG_AB_NO_ZWP
#pragma unroll
      for (int i = 0; i < DS; ++i)  // zero-weighted state components: 0 * inf = NaN, as rollout_cost
        if ((ZW >> i) & 1u) J[c] = fma_r(P.R1d[i], y[c][i] * y[c][i], J[c]);
      J[c] += zw0;
#endif
    }
    Jout[c] = J[c];
  }
}

// ---- hand-packed form of rollout_mpc_gen_multi<Sys, float, false, true, 4, Sys::ZW_PRESET> ---------------------------
// What the measurements said (tools/valu_rate.hip on MI355X, profiles/r04_valu_rate.txt): one wave issues one vector
// instruction per ~5.5 cycles whether it is v_fma_f32 or v_pk_fma_f32, and a SIMD reaches the 2-cycle v_fma_f32 rate only
// with 8 waves resident - at the 3-4 waves per SIMD these register-heavy rollouts run at, a packed instruction delivers
// 0.91 of the lane peak where scalar ones deliver 0.73.  hipcc's SLP vectoriser already packed the multi-candidate rollout,
// but around the value-aliasing trick above: 64 vector instructions per step of four candidates, 19 of them moves that
// build register pairs, shared quantities computed per candidate and discarded, 125-135 VGPRs.  Written out with the two
// candidates of a pair in the two halves of an ext-vector and the shared heading sub-trajectory computed once, the same
// step is 26 instructions (15 packed) in 40 VGPRs.  Every component goes through the operation sequence of
// rollout_cost<MPC, 0, -1, G1, ZW_PRESET> - fma for fma, product for product - so the costs are the same bits
// (tests/test_hip_knobs.py: RCG_NO_GEN_MULTI, which runs the scalar form, hashes identically).
template <typename Sys>
struct GenPk {
  static constexpr bool supported = false;
  static constexpr bool fuse_tick = false;
};
template <>
struct GenPk<Sys3WRobot> {  // state (x, y, alpha, v, omega), inputs (F, M); candidates share M, hence alpha and omega
  static constexpr bool supported = true;
  // rcg_control_tick as ONE launch of k_ticks_pk (env step + decision): 5-8 % shorter than k_sim + k_actor's packed instance
  // from 8192 to 65536 envs, equal at configs[4]'s 21 846 (profiles/r04_ab_pool_gpw.txt)
  static constexpr bool fuse_tick = true;
  __device__ __forceinline__ static void run(const KParams<float>& P, const Sys3WRobot::Pre<float>& pre, int N,
                                             const float* xs, const float* y0, const float* u0v, float u1, float* Jout) {
    const float h = P.h_pred;
    v2f X[2], Y[2], V[2], S0[2], S1[2], D3[2];
    float al = xs[2], om = xs[4];
    float S2 = fma_r(y0[2], y0[2], 0.0f);
    const float d4 = pre.inv_I * u1;
    const float zw0 = fma_r(P.R1d[4], y0[4] * y0[4], fma_r(P.R1d[3], y0[3] * y0[3], 0.0f));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      X[p] = pk_splat(xs[0]);
      Y[p] = pk_splat(xs[1]);
      V[p] = pk_splat(xs[3]);
      S0[p] = pk_splat(fma_r(y0[0], y0[0], 0.0f));
      S1[p] = pk_splat(fma_r(y0[1], y0[1], 0.0f));
      D3[p] = pk_splat(pre.inv_m) * v2f{u0v[2 * p], u0v[2 * p + 1]};
    }
    for (int kk = 1; kk < N; ++kk) {
      float s, c;
      sincos_hw(al, &s, &c);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const v2f d0 = V[p] * pk_splat(c), d1 = V[p] * pk_splat(s);
        X[p] = pk_fma(pk_splat(h), d0, X[p]);
        Y[p] = pk_fma(pk_splat(h), d1, Y[p]);
        V[p] = pk_fma(pk_splat(h), D3[p], V[p]);
      }
      const float aln = fma_r(h, om, al);
      om = fma_r(h, d4, om);
      al = aln;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        S0[p] = pk_fma(X[p], X[p], S0[p]);
        S1[p] = pk_fma(Y[p], Y[p], S1[p]);
      }
      S2 = fma_r(al, al, S2);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      v2f J = pk_fma(pk_splat(P.R1d[0]), S0[p], pk_splat(0.0f));
      J = pk_fma(pk_splat(P.R1d[1]), S1[p], J);
      J = pk_fma(pk_splat(P.R1d[2]), pk_splat(S2), J);
#ifndef RCG_AB_NO_ZWP
      // the zero-weighted state components (v, omega): 0 * inf = NaN, as rollout_cost (J itself is untouched, bit for
      // bit, whenever they are finite: the additions are of exact zeros)
      const v2f vl = N > 1 ? V[p] : pk_splat(y0[3]);
      const float ol = N > 1 ? om : y0[4];
      J = pk_fma(pk_splat(P.R1d[3]), vl * vl, J);
      J = J + pk_splat(fma_r(P.R1d[4], ol * ol, zw0));
#endif
      Jout[2 * p] = J.x;
      Jout[2 * p + 1] = J.y;
    }
  }
};
template <>
struct GenPk<Sys3WRobotNI> {  // state (x, y, alpha), inputs (v, omega); candidates share omega, hence alpha
  static constexpr bool supported = true;
  // the kinematic robot's tick stays two launches: its k_ticks_pk lasts 40 us at 21 845 envs and 60.5 at 65 536, k_sim +
  // k_actor's packed instance 25 + 4.5 and 50 + 4.5 (profiles/r04_ab_pool_gpw.txt); rcg_control_ticks (T ticks in one launch,
  // small batches) still uses k_ticks_pk
  static constexpr bool fuse_tick = false;
  __device__ __forceinline__ static void run(const KParams<float>& P, const Sys3WRobotNI::Pre<float>&, int N,
                                             const float* xs, const float* y0, const float* u0v, float u1, float* Jout) {
    const float h = P.h_pred;
    v2f X[2], Y[2], U0[2], S0[2], S1[2];
    float al = xs[2];
    float S2 = fma_r(y0[2], y0[2], 0.0f);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      X[p] = pk_splat(xs[0]);
      Y[p] = pk_splat(xs[1]);
      U0[p] = v2f{u0v[2 * p], u0v[2 * p + 1]};
      S0[p] = pk_splat(fma_r(y0[0], y0[0], 0.0f));
      S1[p] = pk_splat(fma_r(y0[1], y0[1], 0.0f));
    }
    for (int kk = 1; kk < N; ++kk) {
      float s, c;
      sincos_hw(al, &s, &c);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const v2f d0 = U0[p] * pk_splat(c), d1 = U0[p] * pk_splat(s);
        X[p] = pk_fma(pk_splat(h), d0, X[p]);
        Y[p] = pk_fma(pk_splat(h), d1, Y[p]);
      }
      al = fma_r(h, u1, al);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        S0[p] = pk_fma(X[p], X[p], S0[p]);
        S1[p] = pk_fma(Y[p], Y[p], S1[p]);
      }
      S2 = fma_r(al, al, S2);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      v2f J = pk_fma(pk_splat(P.R1d[0]), S0[p], pk_splat(0.0f));
      J = pk_fma(pk_splat(P.R1d[1]), S1[p], J);
      J = pk_fma(pk_splat(P.R1d[2]), pk_splat(S2), J);
      Jout[2 * p] = J.x;
      Jout[2 * p + 1] = J.y;
    }
  }
};

// NaN -> +inf in one instruction: v_min_f32 returns its non-NaN operand (IEEE minNum), every other value is its own
// minimum with +inf
__device__ __forceinline__ float nan_to_inf(float v) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(__builtin_huge_valf()));
  return r;
}
__device__ __forceinline__ double nan_to_inf(double v) { return (v != v) ? __builtin_huge_val() : v; }

// Tiles t .. t + NC - 1 of the generated grid for this lane through rollout_mpc_gen_multi, folded into the lane's
// running (bestJ, bestI, bestU) in candidate order.  Requires one env per wave (K >= 64), du = 2 and 64 % g == 0, so
// that candidate k + 64 has the same second level as candidate k.
// PKONLY: the hand-packed rollout and nothing else - the caller (a kernel instance that exists for this one regime) has
// checked gamma == 1, the preset's zero weights, float, no target: the register budget of that instance is the packed
// rollout's (the instances that carry every variant need 125-235 VGPRs).
// LEAN (k_ticks_pk: K a multiple of 256, every lane has a candidate in every tile, the caller starts with bestI = its first
// candidate and regenerates the winner's action from the index): the running best is folded in with one v_min (NaN -> +inf:
// v_min_f32 returns the other operand) and one compare per candidate - no sentinel test, no action bookkeeping.  Same winner:
// a candidate replaces the best iff its cost is strictly lower, and an all-+inf lane keeps its first index either way.
template <typename Sys, typename real, bool TGT, int NC, bool PKONLY = false, bool LEAN = false>
__device__ __forceinline__ void gen_multi_tiles(const KParams<real>& P, const typename Sys::template Pre<real>& pre, int N,
                                                int K, int g, int t, int lane, bool env_ok, const real* xs,
                                                const real* y0, real& bestJ, int& bestI, real* bestU) {
  const int k0 = t * 64 + lane;
  real ua[NC][2], u0v[NC], J[NC];
#ifdef RCG_AB_DIV
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ua[c][0] = ua[c][1] = 0;
    gen_candidate<2, real>(P, g, k0 + 64 * c, ua[c]);
    u0v[c] = ua[c][0];
  }
#else
  {
    // gen_candidate's levels without its integer divisions: 64 % g == 0 makes g a power of two, so candidate k0 + 64 c has
    // first level (k0 >> lg) + c (64 >> lg) and second level k0 & (g - 1) - the same integers, fed to the same arithmetic
    // (one runtime division per candidate was 20 VALU instructions, a third of what a 4-candidate group costs outside
    // its horizon loop)
    const int lg = 31 - __builtin_clz((unsigned)g);
    const int gi0 = k0 >> lg, gj = k0 & (g - 1), dgi = 64 >> lg;
    const real den = (real)(g > 1 ? g - 1 : 1);
    const real u1 = fma_r((real)gj, (P.hi[1] - P.lo[1]) / den, P.lo[1]);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      ua[c][0] = fma_r((real)(gi0 + c * dgi), (P.hi[0] - P.lo[0]) / den, P.lo[0]);
      ua[c][1] = u1;
      u0v[c] = ua[c][0];
    }
  }
#endif
  if constexpr (PKONLY) {
    static_assert(std::is_same<real, float>::value && !TGT && NC == 4 && GenPk<Sys>::supported, "see GenPk");
    GenPk<Sys>::run(P, pre, N, xs, y0, u0v, ua[0][1], J);  // the same bits, two candidates per instruction
  } else if (P.gamma == (real)1)  // wave-uniform
  {
    if (Sys::ZW_PRESET != 0u && (P.zero_w & Sys::ZW_PRESET) == Sys::ZW_PRESET)  // wave-uniform, see rollout_dispatch
      rollout_mpc_gen_multi<Sys, real, TGT, true, NC, Sys::ZW_PRESET>(P, pre, N, xs, y0, u0v, ua[0][1], J);
    else
      rollout_mpc_gen_multi<Sys, real, TGT, true, NC>(P, pre, N, xs, y0, u0v, ua[0][1], J);
  }
  else
    rollout_mpc_gen_multi<Sys, real, TGT, false, NC>(P, pre, N, xs, y0, u0v, ua[0][1], J);
  if constexpr (LEAN) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const real Jc = nan_to_inf(J[c]);
      const bool take = Jc < bestJ;
      bestJ = take ? Jc : bestJ;
      bestI = take ? k0 + 64 * c : bestI;
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = k0 + 64 * c;
    const real Jc = (J[c] != J[c]) ? inf_r<real>() : J[c];  // NaN counts as +inf
    if (env_ok && k < K && (Jc < bestJ || bestI == 0x7fffffff)) {
      bestJ = Jc;
      bestI = k;
      bestU[0] = ua[c][0];
      bestU[1] = ua[0][1];
    }
  }
}

// upd_accum_obj (controllers.py:1086-1093): accum_obj_val += stage_obj(obs, action) * sampling_time - the product rounded,
// then the sum rounded, as numpy evaluates the reference's statement and as the streamed production kernels do (their
// sum is a no-return atomic add of the rounded product).  Under -ffp-contract=fast the backend fuses a product into the
// sum that consumes it whatever the source says (a `#pragma clang fp contract(off)` does not reach it), so the product is
// passed through an empty asm statement, which makes it a value of its own: every tick kernel of the library then leaves
// the same ACCUM bits (rounds 1-3 had a fused multiply-add in k_actor / k_ticks and two roundings in k_actor_dma).
__device__ __forceinline__ float opaque_r(float v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ double opaque_r(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <typename Sys, bool TGT, typename real>
__device__ __forceinline__ real accum_update(const KParams<real>& P, const real* obs, const real* act, real accum) {
  constexpr int NCHI = Sys::DS + Sys::DU;
  real chi[NCHI];
  make_chi<Sys::DS, Sys::DU, TGT, real>(P, obs, act, chi);
  const real inc = opaque_r(stage_any<NCHI, real>(P, chi) * P.sampling_time);
  return accum + inc;
}

// The decision of one wave's env(s): K x _actor_cost + argmin + tick epilogue.  The body of k_actor, and of the decision
// phase of k_ticks_mem (rcg_ticks.hpp).  `wave`: the wave's index in the grid (wave-uniform), `lds`: its LDS region (streamed).
// `staged` (k_ticks_mem over a caller's tensor, ticks after the first, one tile per wave): the wave's tile is still in its LDS
// region from the previous tick - nothing else writes there - and is not staged again.
// DIRECT (streamed rows too long for an LDS tile - beyond RCG_MAX_ROW reals, rcg.h): nothing is staged, every lane walks ITS row
// straight from HBM (a row is contiguous: each of a lane's cache lines serves 16 / DU of its steps; the reference's horizon is
// unbounded, controllers.py:965 - any Nactor runs here, the short rows every preset has run on the staged kernels)
template <typename Sys, typename real, bool GENERIC, bool TGT, bool STREAM, bool PKONLY = false, bool DIRECT = false>
__device__ __forceinline__ void actor_wave(const ActorArgs<real>& A, const KParams<real>& P, const long wave, real* const lds,
                                           const bool staged = false) {
  static_assert(!DIRECT || (STREAM && GENERIC && !PKONLY), "DIRECT is a variant of the streamed generic instance");
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const int lane = threadIdx.x & 63;
  const long B = P.B;
  const int K = A.K, N = P.n_actor, R = N * DU;
  if (wave * A.G >= B) return;  // wave-uniform; no workgroup barrier is used below

  const bool big = K >= 64;
  const int seg = big ? 64 : A.Kp;          // lanes that share one env
  const int e = big ? 0 : lane / seg;       // env slot inside the wave
  const int kl = big ? lane : lane - e * seg;
  const long b_raw = wave * A.G + e;
  const bool env_ok = b_raw < B;
  const long b = env_ok ? b_raw : B - 1;

  // per-env inputs (broadcast loads: the lanes of one segment read the same address)
  real y0[DS], xs[DS];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    y0[c] = A.obs[(long)c * B + b];
    xs[c] = A.state_sys[(long)c * B + b];
  }
  const auto pre = load_pre<Sys, real>(P, A.pars_env, b);
  // critic weights of this lane's env, once, into registers (they were re-read from memory at every use inside the
  // horizon loop: the J store below may alias them, so the compiler cannot hoist the loads itself)
  constexpr int DCMAX = GENERIC ? NCHI * (NCHI + 1) / 2 + NCHI : 1;
  real wreg[DCMAX];
  if (GENERIC) {
    const bool has_w = P.mode != RCG_MODE_MPC && A.w != nullptr;
#pragma unroll
    for (int i = 0; i < DCMAX; ++i) wreg[i] = (has_w && i < P.dc) ? A.w[(long)i * B + b] : (real)0;
  }
  auto wget = [&](int i) -> real { return wreg[GENERIC ? i : 0]; };

  real bestJ = inf_r<real>();
  int bestI = 0x7fffffff;
  real bestU[DU];
#pragma unroll
  for (int c = 0; c < DU; ++c) bestU[c] = 0;

  const int envs_here = big ? 1 : (int)((B - wave * A.G) < A.G ? (B - wave * A.G) : A.G);

  // generated grid, MPC / diagonal R1, two inputs, 64 % g == 0 (K = g * g in {256, 1024, 4096}: a multiple of four
  // tiles): the lane's tiles share their second input and are rolled out four at a time with the shared sub-trajectory
  // computed once (rollout_mpc_gen_multi)
  const bool multi_ok = !STREAM && !GENERIC && DU == 2 && Sys::SHARED_U1 != 0 && big && A.grid_g > 0 &&
                        (64 % A.grid_g) == 0 && !A.no_multi;
  for (int t = 0; t < A.n_tiles; ++t) {
    if constexpr (PKONLY) {
      gen_multi_tiles<Sys, real, TGT, 4, true>(P, pre, N, K, A.grid_g, t, lane, env_ok, xs, y0, bestJ, bestI, bestU);
      t += 3;
      continue;
    }
    if constexpr (!STREAM && !GENERIC && DU == 2 && Sys::SHARED_U1 != 0) {
      if (multi_ok && t + 4 <= A.n_tiles) {
        gen_multi_tiles<Sys, real, TGT, 4>(P, pre, N, K, A.grid_g, t, lane, env_ok, xs, y0, bestJ, bestI, bestU);
        t += 3;
        continue;
      }
    }
    const int k = big ? t * 64 + kl : kl;
    const bool valid = env_ok && k < K;
    int r = 0;  // my row inside the tile
    real ugen[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) ugen[c] = 0;
    if (STREAM) {
      const long row0 = big ? b * K + (long)t * 64 : wave * A.G * (long)K;
      const int nrows = big ? (K - t * 64 < 64 ? K - t * 64 : 64) : envs_here * K;
      if (!staged && !DIRECT) {  // wave-uniform
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous tile's LDS reads are done
        stage_tile<real>(A.cand + row0 * R, lds, nrows * R, lane, A.vec_ok);
        // LDS ops of one wave are executed in order; wait for our own writes, no workgroup barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      r = valid ? (big ? kl : e * K + kl) : 0;
    } else {
      gen_candidate<DU, real>(P, A.grid_g, k, ugen);
    }
    const real* urow = lds + (size_t)r * R;
    if constexpr (DIRECT) {
      const long row0 = big ? b * K + (long)t * 64 : wave * A.G * (long)K;
      urow = A.cand + (row0 + r) * R;
    }
    real u0[DU];
    const real J = rollout_dispatch<Sys, real, GENERIC, TGT, STREAM>(P, pre, N, xs, y0, urow, ugen, wget, u0);

    if (A.J && valid) A.J[b * K + k] = J;
    const real Jc = (J != J) ? inf_r<real>() : J;  // NaN counts as +inf
    if (valid && (Jc < bestJ || bestI == 0x7fffffff)) {
      bestJ = Jc;
      bestI = k;
#pragma unroll
      for (int c = 0; c < DU; ++c) bestU[c] = u0[c];
    }
  }

  segment_argmin<DU, real>(seg, bestJ, bestI, bestU);

  if (kl == 0 && env_ok) {
#pragma unroll
    for (int c = 0; c < DU; ++c)
      if (A.action_out) A.action_out[(long)c * B + b] = bestU[c];
    if (A.best_J) A.best_J[b] = bestJ;
    if (A.best_idx) A.best_idx[b] = bestI;
    if (A.accum) A.accum[b] = accum_update<Sys, TGT, real>(P, y0, bestU, A.accum[b]);
    if (A.step_idx) A.step_idx[b] += 1;
  }
}

// (the instances without the generic cost structures ask for 4 waves per SIMD, i.e. <= 128 VGPRs: the generated-grid
// instance sits at 125-129 registers depending on details, and the step from 4 to 3 resident waves costs it 15 %)
template <typename Sys, typename real, bool GENERIC, bool TGT, bool STREAM, bool PKONLY = false, bool DIRECT = false>
// Resident blocks per CU asked of the compiler for the float64 generated-grid instance.  Round 6, interleaved A/B on one device
// (profiles/r06_ab_gen64_occupancy.txt): unconstrained the 3-wheel robot's instance takes 206 VGPRs = 2 waves per SIMD, 157.4 us per
// C2-shape tick; 3 (168 VGPRs, 8 spilled) 139.2 us; 4 (128 VGPRs, 48 spilled) 196.7 us.  NI and the tank are below 168 anyway.
#ifndef RCG_GEN64_OCC
#define RCG_GEN64_OCC 3
#endif
__global__ __launch_bounds__(256, GENERIC ? 1 : (sizeof(real) > 4 ? (STREAM ? 1 : RCG_GEN64_OCC) : 4)) void k_actor(const ActorArgs<real> A, const KParams<real> P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // readfirstlane makes the wave index provably wave-uniform: tile bases, row counts and the env's
  // addresses then live in SGPRs and the staging control flow is scalar (no exec-mask branches)
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  real* const lds = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * 64 * (P.n_actor * Sys::DU);
  actor_wave<Sys, real, GENERIC, TGT, STREAM, PKONLY, DIRECT>(A, P, wave, lds);
}

// ---------------------------------------------------------------------------------------------
// k_sim: closed_loop_rhs (systems.py:213-253) under classical RK4, lane == env
// ---------------------------------------------------------------------------------------------
template <typename real>
struct SimArgs {
  real* state;           // [ds][B] in/out
  real* state_prev;      // [ds][B] out: state before the last substep
  const real* action;    // [du][B]
  const real* pars_env;  // [np][B] or nullptr
  real* accum;           // [B] (only touched with accum_every_substep)
  uint32_t* status;      // [B]
  int n_sub;
};

// Simulator.sim_step x n_sub for one env held in registers (shared by k_sim and k_ticks): clip the held action
// (systems.py:241-243), n_sub RK4 substeps; a frozen env (status bit 0) is not stepped; a non-finite result freezes the
// env at its last finite state and sets the bit (error convention).  On success x / xp (the state before
// the last substep) / accum (accum_every_substep only) are updated and true is returned.
template <typename Sys, typename real, bool TGT>
__device__ __forceinline__ bool env_substeps(const KParams<real>& P, const typename Sys::template Pre<real>& pre,
                                             int n_sub, real* x, real* xp, const real* a_held, uint32_t& st,
                                             real& accum) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  if (st & 1u) return false;  // frozen env
  real u[DU], xn[DS], xq[DS];
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = P.clip ? clamp_r<real>(a_held[c], P.lo[c], P.hi[c]) : a_held[c];
#pragma unroll
  for (int c = 0; c < DS; ++c) xq[c] = xn[c] = x[c];
  real acc = 0;
  for (int s = 0; s < n_sub; ++s) {
#pragma unroll
    for (int c = 0; c < DS; ++c) xq[c] = xn[c];
    rk4_step<Sys, real>(pre, xn, u, P.dt_sim);
    if (P.accum_every_substep) {
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, xn, u, chi);
      acc = fma_r(stage_any<NCHI, real>(P, chi), P.sampling_time, acc);
    }
  }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < DS; ++c) ok = ok && finite_r<real>(xn[c]);
  if (!ok) {
    st |= 1u;
    return false;
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    x[c] = xn[c];
    xp[c] = xq[c];
  }
  if (P.accum_every_substep) accum += acc;
  return true;
}

template <typename Sys, typename real, bool TGT>
__global__ __launch_bounds__(256) void k_sim(const SimArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long B = P.B;
  if (b >= B) return;
  uint32_t st = A.status[b];
  if (st & 1u) return;  // frozen env

  real x[DS], xp[DS], u[DU];
#pragma unroll
  for (int c = 0; c < DS; ++c) xp[c] = x[c] = A.state[(long)c * B + b];
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = A.action[(long)c * B + b];
  const auto pre = load_pre<Sys, real>(P, A.pars_env, b);
  real accum = P.accum_every_substep ? A.accum[b] : (real)0;
  if (!env_substeps<Sys, real, TGT>(P, pre, A.n_sub, x, xp, u, st, accum)) {
    A.status[b] = st;  // became non-finite: frozen at its last finite state, nothing else is written
    return;
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    A.state[(long)c * B + b] = x[c];
    A.state_prev[(long)c * B + b] = xp[c];
  }
  if (P.accum_every_substep) A.accum[b] = accum;
}

// k_sim with 16 bytes per lane and component: a lane owns VEC = 16 / sizeof(real) CONSECUTIVE envs (the struct-of-arrays
// layout makes their values of one component contiguous), so every load and store is a dwordx4 (1 KiB per wave
// instruction instead of 256 B).  The env step moves 72 B per env for ~400 flop: bandwidth-bound once the batch is large
// (2^24 envs: 5.75 TB/s with 4-byte accesses).  Needs B % VEC == 0; a lane with a frozen env or one that has just gone
// non-finite falls back to per-env stores (k_sim's semantics exactly: that env's state is left as it was).
template <typename Sys, typename real, bool TGT>
__global__ __launch_bounds__(256) void k_sim_v(const SimArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NP = Sys::NP, VEC = 16 / (int)sizeof(real);
  typedef real vreal __attribute__((ext_vector_type(VEC)));
  typedef uint32_t vu32 __attribute__((ext_vector_type(VEC)));
  const long B = P.B;
  const long b0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (b0 >= B) return;
  vreal xv[DS], uv[DU], pv[NP > 0 ? NP : 1], av;
#pragma unroll
  for (int c = 0; c < DS; ++c) xv[c] = *reinterpret_cast<const vreal*>(A.state + (long)c * B + b0);
#pragma unroll
  for (int c = 0; c < DU; ++c) uv[c] = *reinterpret_cast<const vreal*>(A.action + (long)c * B + b0);
  const vu32 stv = *reinterpret_cast<const vu32*>(A.status + b0);
  if (A.pars_env) {
#pragma unroll
    for (int i = 0; i < NP; ++i) pv[i] = *reinterpret_cast<const vreal*>(A.pars_env + (long)i * B + b0);
  }
  if (P.accum_every_substep) av = *reinterpret_cast<const vreal*>(A.accum + b0);
  vreal xo[DS], xpo[DS], ao;
  uint32_t sto[VEC];
  bool all_ok = true;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    real x[DS], xp[DS], u[DU], pe[NP > 0 ? NP : 1];
#pragma unroll
    for (int c = 0; c < DS; ++c) xp[c] = x[c] = xv[c][e];
#pragma unroll
    for (int c = 0; c < DU; ++c) u[c] = uv[c][e];
#pragma unroll
    for (int i = 0; i < NP; ++i) pe[i] = A.pars_env ? pv[i][e] : P.pars[i];
    const auto pre = Sys::template prepare<real>(pe);
    uint32_t st = stv[e];
    real accum = P.accum_every_substep ? av[e] : (real)0;
    const bool ok = env_substeps<Sys, real, TGT>(P, pre, A.n_sub, x, xp, u, st, accum);
    all_ok = all_ok && ok;
    sto[e] = ok ? 0xffffffffu : st;  // marker: stepped; otherwise the status to keep / write
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      xo[c][e] = x[c];
      xpo[c][e] = xp[c];
    }
    ao[e] = accum;
  }
  if (all_ok) {
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      *reinterpret_cast<vreal*>(A.state + (long)c * B + b0) = xo[c];
      *reinterpret_cast<vreal*>(A.state_prev + (long)c * B + b0) = xpo[c];
    }
    if (P.accum_every_substep) *reinterpret_cast<vreal*>(A.accum + b0) = ao;
    return;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {  // rare: per-env writes, exactly what k_sim does for each of them
    const long b = b0 + e;
    if (sto[e] == 0xffffffffu) {
#pragma unroll
      for (int c = 0; c < DS; ++c) {
        A.state[(long)c * B + b] = xo[c][e];
        A.state_prev[(long)c * B + b] = xpo[c][e];
      }
      if (P.accum_every_substep) A.accum[b] = ao[e];
    } else if (!(stv[e] & 1u)) {
      A.status[b] = sto[e];  // became non-finite in this step: frozen at its last finite state
    }
  }
}

// ---------------------------------------------------------------------------------------------
// unit operators, lane == point
// ---------------------------------------------------------------------------------------------
template <typename Sys, typename real>
__global__ void k_rhs(const real* state, const real* action, real* dstate, real* clipped, const real* pars_env,
                      long n, int clip, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real x[DS], u[DU], d[DS];
#pragma unroll
  for (int c = 0; c < DS; ++c) x[c] = state[(long)c * n + i];
#pragma unroll
  for (int c = 0; c < DU; ++c) {
    const real a = action[(long)c * n + i];
    u[c] = (clip && P.clip) ? clamp_r<real>(a, P.lo[c], P.hi[c]) : a;
    if (clipped) clipped[(long)c * n + i] = u[c];
  }
  const auto pre = load_pre<Sys, real>(P, pars_env, i);
  Sys::template rhs<real>(pre, x, u, d);
#pragma unroll
  for (int c = 0; c < DS; ++c) dstate[(long)c * n + i] = d[c];
}

template <typename Sys, typename real>
__global__ void k_stage_obj(const real* obs, const real* act, real* out, long n, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real y[DS], u[DU], chi[NCHI];
#pragma unroll
  for (int c = 0; c < DS; ++c) y[c] = obs[(long)c * n + i];
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = act[(long)c * n + i];
  if (P.has_target)
    make_chi<DS, DU, true, real>(P, y, u, chi);
  else
    make_chi<DS, DU, false, real>(P, y, u, chi);
  out[i] = stage_any<NCHI, real>(P, chi);
}

template <typename Sys, typename real>
__global__ void k_critic(const real* obs, const real* act, const real* w, real* out, long n, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real y[DS], u[DU], chi[NCHI];
#pragma unroll
  for (int c = 0; c < DS; ++c) y[c] = obs[(long)c * n + i];
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = act[(long)c * n + i];
  if (P.has_target)
    make_chi<DS, DU, true, real>(P, y, u, chi);
  else
    make_chi<DS, DU, false, real>(P, y, u, chi);
  out[i] = critic_value<DS, DU, real>(P, chi, y, u, [&](int k) -> real { return w[(long)k * n + i]; });
}

// _critic_cost (controllers.py:1216-1245) on the OLDEST Ncritic buffer rows, lane == env
template <typename Sys, typename real>
__global__ void k_critic_cost(const real* w, const real* w_prev, const real* obs_buf, const real* act_buf, real* Jc,
                              const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long B = P.B;
  if (b >= B) return;
  real acc = 0;
  for (int k = P.n_critic - 1; k >= 1; --k) {
    real yp[DS], yn[DS], up[DU], un[DU], chip[NCHI], chin[NCHI];
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      yp[c] = obs_buf[((long)(k - 1) * DS + c) * B + b];
      yn[c] = obs_buf[((long)k * DS + c) * B + b];
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) {
      up[c] = act_buf[((long)(k - 1) * DU + c) * B + b];
      un[c] = act_buf[((long)k * DU + c) * B + b];
    }
    if (P.has_target) {
      make_chi<DS, DU, true, real>(P, yp, up, chip);
      make_chi<DS, DU, true, real>(P, yn, un, chin);
    } else {
      make_chi<DS, DU, false, real>(P, yp, up, chip);
      make_chi<DS, DU, false, real>(P, yn, un, chin);
    }
    const real cp = critic_value<DS, DU, real>(P, chip, yp, up, [&](int i) -> real { return w[(long)i * B + b]; });
    const real cn =
        critic_value<DS, DU, real>(P, chin, yn, un, [&](int i) -> real { return w_prev[(long)i * B + b]; });
    const real e = cp - P.gamma * cn - stage_any<NCHI, real>(P, chip);
    acc += (real)0.5 * e * e;
  }
  Jc[b] = acc;
}

template <typename real>
__global__ void k_episode_reset(real* state, real* state_prev, const real* state_init, real* action, real* accum,
                                real* returns, int32_t* step_idx, int32_t* episode_idx, uint32_t* status, int ds,
                                int du, real a0, real a1, long B) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  returns[b] = accum[b];
  accum[b] = 0;
  for (int c = 0; c < ds; ++c) {
    const real v = state_init[(long)c * B + b];
    state[(long)c * B + b] = v;
    state_prev[(long)c * B + b] = v;
  }
  action[b] = a0;
  if (du > 1) action[B + b] = a1;
  step_idx[b] = 0;
  episode_idx[b] += 1;
  status[b] = 0;
}

// One workgroup of 1024 lanes strides over the shard: deterministic (count, sum, sumsq, min, max, n_failed).
template <typename real>
__global__ __launch_bounds__(1024) void k_stats(const real* v, const uint32_t* status, long B, double* out) {
  __shared__ double s_sum[16], s_sq[16], s_min[16], s_max[16], s_fail[16];
  double sum = 0, sq = 0, mn = __builtin_huge_val(), mx = -__builtin_huge_val(), nf = 0;
  for (long i = threadIdx.x; i < B; i += blockDim.x) {
    const double x = (double)v[i];
    sum += x;
    sq += x * x;
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
    nf += (status[i] & 1u) ? 1.0 : 0.0;
  }
  for (int m = 1; m < 64; m <<= 1) {
    sum += __shfl_xor(sum, m, 64);
    sq += __shfl_xor(sq, m, 64);
    nf += __shfl_xor(nf, m, 64);
    const double omn = __shfl_xor(mn, m, 64), omx = __shfl_xor(mx, m, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_sum[wv] = sum;
    s_sq[wv] = sq;
    s_min[wv] = mn;
    s_max[wv] = mx;
    s_fail[wv] = nf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    double a = 0, q = 0, lo = __builtin_huge_val(), hi = -__builtin_huge_val(), f = 0;
    for (int i = 0; i < nw; ++i) {
      a += s_sum[i];
      q += s_sq[i];
      lo = s_min[i] < lo ? s_min[i] : lo;
      hi = s_max[i] > hi ? s_max[i] : hi;
      f += s_fail[i];
    }
    out[0] = (double)B;
    out[1] = a;
    out[2] = q;
    out[3] = lo;
    out[4] = hi;
    out[5] = f;
  }
}

}  // namespace rcg
