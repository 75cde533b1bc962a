// rcg_loop.hpp - k_loop: the glue of rcg_loop_step (rcg.h) - one iteration of the reference's headless loop
// (presets/main_3wrobot.py:419-429) for the drop-in classes at small batch.  lane == env; up to three stages in ONE launch:
//   set   System.receive_action: ACTION := the caller's action (read from the handle's pinned host buffer, [B][du] doubles)
//   sim   Simulator.sim_step: env_substeps - the code k_sim runs (clip the held action, RK4 substeps, freeze on non-finite)
//   tail  CtrlOptPred.stage_obj(observation = STATE, action = ACTION) with k_stage_obj's statements, and everything the loop body
//         reads back, env by env, written straight into the pinned host buffer (no device-to-host copy call)
// A step that is not a controller sample is ONE launch (set + sim + tail); a sample is set + sim, the decision (k_actor_opt; in
// RQL / SQL k_critic_fit's env step + push + fit instead of `sim`), tail.  Same device functions as the separate entry points,
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

template <typename Sys, typename real>
__global__ __launch_bounds__(64) void k_loop(const LoopArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  const long B = P.B;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  real u[DU], x[DS];
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
  if (A.do_sim) {  // k_sim's body
    uint32_t st = A.sim.status[b];
    if (!(st & 1u)) {
      real xp[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) xp[c] = x[c];
      const auto pre = load_pre<Sys, real>(P, A.sim.pars_env, b);
      real accum = P.accum_every_substep ? A.sim.accum[b] : (real)0;
      const bool ok = P.has_target ? env_substeps<Sys, real, true>(P, pre, A.sim.n_sub, x, xp, u, st, accum)
                                   : env_substeps<Sys, real, false>(P, pre, A.sim.n_sub, x, xp, u, st, accum);
      if (!ok) {
        A.sim.status[b] = st;  // became non-finite: frozen at its last finite state, nothing else is written
      } else {
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          A.sim.state[(long)c * B + b] = x[c];
          A.sim.state_prev[(long)c * B + b] = xp[c];
        }
        if (P.accum_every_substep) A.sim.accum[b] = accum;
      }
    }
  }
  if (A.do_tail) {
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
    o[DS + DU + 1] = A.decided ? (double)A.best_J[b] : __builtin_nan("");
    for (int i = 0; i < A.dc; ++i) o[DS + DU + 2 + i] = (double)A.w[(long)i * B + b];
    __threadfence_system();  // the row is visible to the host before its sequence number is
    *(volatile double*)(A.flag + b) = A.seq;
  }
}

}  // namespace rcg
