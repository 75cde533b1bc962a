// rcg_loop.hpp - k_loop: the glue of rcg_loop_step (rcg.h) - one iteration of the reference's headless loop
// (presets/main_3wrobot.py:419-429) for the drop-in classes at small batch.  lane == env; up to three stages in ONE launch:
//   set   System.receive_action: ACTION := the caller's action (read from the handle's pinned host buffer, [B][du] doubles)
//   sim   Simulator.sim_step: env_substeps - the code k_sim runs (clip the held action, RK4 substeps, freeze on non-finite)
//   tail  CtrlOptPred.stage_obj(observation = STATE, action = ACTION) with k_stage_obj's statements, and everything the loop body
//         reads back, env by env, written straight into the pinned host buffer (no device-to-host copy call)
// A step that is not a controller sample is ONE launch (set + sim + tail); a sample is set + sim, the decision (k_actor_opt; in
// RQL / SQL k_critic_fit's env step + push + fit instead of `sim`), tail - three launches, except for the plain MPC decision
// (diagonal stage cost, no curvature pairs: what the presets run), where k_actor_opt's LOOP instance does head and tail itself:
// ONE launch (rcg_actor_opt.hpp).  Same device functions as the separate entry points,
// so every number equals theirs bit for bit (tests/test_hip_loop_step.py).
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

template <typename real>
struct LoopArgs {
  SimArgs<real> sim;      // state / state_prev / status / accum / pars_env / n_sub (action: see below)
  real* action;           // [du][B] ACTION (written by `set`, read by `sim` / `tail`)
  const double* act_in;   // pinned host [B][du], or nullptr: keep ACTION
  const real* best_J;     // [B]
  const real* w;          // [dc][B] or nullptr
  double* out;            // pinned host [B][ds + du + 2 + dc]
  double* flag;           // pinned host [B]: := seq once the env's row is out (the host polls it instead of a stream wait:
  double seq;             //   6.9 us per launch + wait against 12.4 with hipStreamSynchronize, tools/sync_probe.hip)
  int do_sim, do_tail, decided, dc;
};

// set + sim for env b: leaves u = ACTION, x = STATE, xprev = STATE_PREV as the fields stand afterwards
template <typename Sys, typename real>
__device__ __forceinline__ void loop_head(const LoopArgs<real>& A, const KParams<real>& P, long b, real* u, real* x, real* xprev) {
  constexpr int DS = Sys::DS, DU = Sys::DU;
  const long B = P.B;
  if (A.act_in) {
#pragma unroll
    for (int c = 0; c < DU; ++c) {
      u[c] = (real)A.act_in[b * DU + c];
      A.action[(long)c * B + b] = u[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < DU; ++c) u[c] = A.action[(long)c * B + b];
  }
#pragma unroll
  for (int c = 0; c < DS; ++c) x[c] = A.sim.state[(long)c * B + b];
  bool stepped = false;
  if (A.do_sim) {  // k_sim's body
    uint32_t st = A.sim.status[b];
    if (!(st & 1u)) {
      real xn[DS], xp[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) xn[c] = xp[c] = x[c];
      const auto pre = load_pre<Sys, real>(P, A.sim.pars_env, b);
      real accum = P.accum_every_substep ? A.sim.accum[b] : (real)0;
      const bool ok = P.has_target ? env_substeps<Sys, real, true>(P, pre, A.sim.n_sub, xn, xp, u, st, accum)
                                   : env_substeps<Sys, real, false>(P, pre, A.sim.n_sub, xn, xp, u, st, accum);
      if (!ok) {
        A.sim.status[b] = st;  // became non-finite: frozen at its last finite state, nothing else is written
      } else {
        stepped = true;
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          x[c] = xn[c];
          if (xprev) xprev[c] = xp[c];
          A.sim.state[(long)c * B + b] = xn[c];
          A.sim.state_prev[(long)c * B + b] = xp[c];
        }
        if (P.accum_every_substep) A.sim.accum[b] = accum;
      }
    }
  }
  if (xprev && !stepped) {
#pragma unroll
    for (int c = 0; c < DS; ++c) xprev[c] = A.sim.state_prev[(long)c * B + b];
  }
}

// tail for env b: stage_obj(x, u) and the row of the loop body into the pinned host buffer, then the sequence number
template <typename Sys, typename real>
__device__ __forceinline__ void loop_tail(const LoopArgs<real>& A, const KParams<real>& P, long b, const real* u, const real* x,
                                          double best_J) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const long B = P.B;
  real chi[NCHI];
  if (P.has_target)
    make_chi<DS, DU, true, real>(P, x, u, chi);
  else
    make_chi<DS, DU, false, real>(P, x, u, chi);
  const real stage = stage_any<NCHI, real>(P, chi);
  double* const o = A.out + (size_t)b * (DS + DU + 2 + A.dc);
#pragma unroll
  for (int c = 0; c < DS; ++c) o[c] = (double)x[c];
#pragma unroll
  for (int c = 0; c < DU; ++c) o[DS + c] = (double)u[c];
  o[DS + DU] = (double)stage;
  o[DS + DU + 1] = best_J;
  for (int i = 0; i < A.dc; ++i) o[DS + DU + 2 + i] = (double)A.w[(long)i * B + b];
  __threadfence_system();  // the row is visible to the host before its sequence number is
  *(volatile double*)(A.flag + b) = A.seq;
}

template <typename Sys, typename real>
__global__ __launch_bounds__(64) void k_loop(const LoopArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  real u[DU], x[DS];
  loop_head<Sys, real>(A, P, b, u, x, (real*)nullptr);
  if (A.do_tail) loop_tail<Sys, real>(A, P, b, u, x, A.decided ? (double)A.best_J[b] : __builtin_nan(""));
}

}  // namespace rcg
