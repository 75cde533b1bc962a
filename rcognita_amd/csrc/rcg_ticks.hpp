// rcg_ticks.hpp - T control ticks in ONE launch (rcg_control_ticks, rcg_control_tick_n): k_ticks and k_ticks_pk.
//
// Envs never interact, so a tick needs no grid-wide step: the wave that owns an env (or, K < 64, a segment of it) keeps
// the env's state, held action, accum and counters in registers and loops `T` times over {Simulator.sim_step,
// K x _actor_cost, argmin, upd_accum_obj} - the loop body of presets/main_3wrobot.py:415-468 for MPC.  Every lane of the
// env's segment integrates the env step redundantly - no cross-lane traffic, no barrier - and the argmin butterfly leaves
// the winner in every lane, which becomes the held action of the next tick.  The arithmetic is the code k_sim / k_sim_dist
// and k_actor run (env_substeps, env_substeps_dist, rollout_dispatch, segment_argmin, accum_update), so T ticks here equal T
// calls of rcg_control_tick bit for bit.  What it removes is the launch-bound regime of small batches: two launches
// (~8 us) per tick against ~1 us of work at B = 1024, K = 64.
//   candidates   the generated level grid, or (STREAM) the caller's tensor [B][K][N][du]: the wave's rows are staged into
//                its LDS region ONCE, before the first tick, when they fit (`stage_once`: K rows of an env, or the rows of
//                the 64 / Kp envs of a packed wave) - T ticks then re-walk LDS, not HBM -, else tile by tile every tick as
//                k_actor does;
//   disturbance  (`dist`, RCG_FLAG_DISTURB) the env step is env_substeps_dist: the disturbance state, the substep counter
//                of the noise stream and the episode index travel in registers with the state.
// RQL / SQL ticks refit the critic between the env step and the decision; they go through k_ticks_mem (below), which runs
// the launches of rcg_control_tick as phases of one persistent launch.
#pragma once
#include "rcg_critic_fit.hpp"
#include "rcg_disturb.hpp"
#include "rcg_kernels.hpp"

namespace rcg {

template <typename real>
struct TicksArgs {
  real* state;          // [ds][B] in/out
  real* state_prev;     // [ds][B] in/out
  real* action;         // [du][B] in/out: the held action (ZOH)
  const real* pars_env; // [np][B] or nullptr
  real* accum;          // [B] in/out
  int32_t* step_idx;    // [B] in/out
  uint32_t* status;     // [B] in/out
  real* best_J;         // [B] out (last tick)
  int32_t* best_idx;    // [B] out (last tick)
  const real* cand;     // STREAM: [B][K][N][du]
  real* disturb;        // dist: [dd][B] in/out
  int32_t* substep_idx; // dist: [B] in/out
  const int32_t* episode_idx;  // dist: [B]
  DisturbPars D;        // dist
  int T;                // ticks
  int n_sub;            // RK4 substeps per tick
  int K, Kp, G, n_tiles, grid_g;  // as ActorArgs
  int no_multi;                   // as ActorArgs
  int gpw;                        // k_ticks_pk: consecutive envs per wave
  int vec_ok, stage_once, dist;   // STREAM: 16-byte staging; all rows of the wave resident in LDS; disturbance model
  int lds_reals;                  // STREAM: reals of LDS per wave
};

template <typename Sys, typename real, bool GENERIC, bool TGT, bool STREAM>
__global__ __launch_bounds__(256) void k_ticks(const TicksArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, DD = Disturb<Sys>::DD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const int K = A.K, N = P.n_actor, R = N * DU;
  if (wave * A.G >= B) return;  // wave-uniform: every wave that stays runs all T ticks and exits

  const bool big = K >= 64;
  const int seg = big ? 64 : A.Kp;
  const int e = big ? 0 : lane / seg;
  const int kl = big ? lane : lane - e * seg;
  const long b_raw = wave * A.G + e;
  const bool env_ok = b_raw < B;
  const long b = env_ok ? b_raw : B - 1;
  const int envs_here = big ? 1 : (int)((B - wave * A.G) < A.G ? (B - wave * A.G) : A.G);

  real x[DS], xp[DS], u[DU], q[DD];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    x[c] = A.state[(long)c * B + b];
    xp[c] = A.state_prev[(long)c * B + b];
  }
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = A.action[(long)c * B + b];
#pragma unroll
  for (int c = 0; c < DD; ++c) q[c] = A.dist ? A.disturb[(long)c * B + b] : (real)0;
  int32_t sub = A.dist ? A.substep_idx[b] : 0;
  const int32_t ep = A.dist ? A.episode_idx[b] : 0;
  const auto pre = load_pre<Sys, real>(P, A.pars_env, b);
  uint32_t st = A.status[b];
  real accum = A.accum[b];
  int32_t steps = A.step_idx[b];
  auto wget = [&](int) -> real { return (real)0; };  // MPC: no critic
  real bestJ = inf_r<real>();
  int bestI = 0x7fffffff;

  real* const lds = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * (STREAM ? A.lds_reals : 0);
  const long row_base = big ? b * K : wave * A.G * (long)K;  // first candidate row of this wave (wave-uniform)
  if (STREAM && A.stage_once) {  // the wave's rows, all of them, once: T ticks re-walk LDS
    stage_tile<real>(A.cand + row_base * R, lds, (big ? K : envs_here * K) * R, lane, A.vec_ok);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }

  for (int t = 0; t < A.T; ++t) {
    if (A.dist)  // k_sim_dist
      env_substeps_dist<Sys, real, TGT>(P, A.D, pre, A.n_sub, A.D.env_id_base + b, ep, x, xp, q, sub, u, st, accum);
    else  // k_sim
      env_substeps<Sys, real, TGT>(P, pre, A.n_sub, x, xp, u, st, accum);
    const real* const xs = P.ref_lag ? xp : x;  // rcg_control_tick's state_sys
    bestJ = inf_r<real>();
    bestI = 0x7fffffff;
    real bestU[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) bestU[c] = 0;
    const bool multi_ok = !STREAM && !GENERIC && DU == 2 && Sys::SHARED_U1 != 0 && big && A.grid_g > 0 &&
                          (64 % A.grid_g) == 0 && !A.no_multi;
    for (int tl = 0; tl < A.n_tiles; ++tl) {  // k_actor
      if constexpr (!STREAM && !GENERIC && DU == 2 && Sys::SHARED_U1 != 0) {
        if (multi_ok && tl + 4 <= A.n_tiles) {
          gen_multi_tiles<Sys, real, TGT, 4>(P, pre, N, K, A.grid_g, tl, lane, env_ok, xs, x, bestJ, bestI, bestU);
          tl += 3;
          continue;
        }
      }
      const int k = big ? tl * 64 + kl : kl;
      const bool valid = env_ok && k < K;
      real ugen[DU], u0[DU];
#pragma unroll
      for (int c = 0; c < DU; ++c) ugen[c] = 0;
      const real* urow = nullptr;
      if (STREAM) {
        int r = valid ? (big ? kl : e * K + kl) : 0;  // my row inside the tile
        if (A.stage_once) {
          r += big ? tl * 64 : 0;
          if (!valid) r = 0;
        } else {
          const int nrows = big ? (K - tl * 64 < 64 ? K - tl * 64 : 64) : envs_here * K;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's LDS reads are done
          stage_tile<real>(A.cand + (row_base + (big ? (long)tl * 64 : 0)) * R, lds, nrows * R, lane, A.vec_ok);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
        urow = lds + (size_t)r * R;
      } else {
        gen_candidate<DU, real>(P, A.grid_g, k, ugen);
      }
      const real J = rollout_dispatch<Sys, real, GENERIC, TGT, STREAM>(P, pre, N, xs, x, urow, ugen, wget, u0);
      const real Jc = (J != J) ? inf_r<real>() : J;
      if (valid && (Jc < bestJ || bestI == 0x7fffffff)) {
        bestJ = Jc;
        bestI = k;
#pragma unroll
        for (int c = 0; c < DU; ++c) bestU[c] = u0[c];
      }
    }
    segment_argmin<DU, real>(seg, bestJ, bestI, bestU);
#pragma unroll
    for (int c = 0; c < DU; ++c) u[c] = bestU[c];  // receive_action: held until the next tick
    if (!P.accum_every_substep) accum = accum_update<Sys, TGT, real>(P, x, u, accum);
    steps += 1;
  }

  if (kl == 0 && env_ok) {
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      A.state[(long)c * B + b] = x[c];
      A.state_prev[(long)c * B + b] = xp[c];
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) A.action[(long)c * B + b] = u[c];
    A.accum[b] = accum;
    A.step_idx[b] = steps;
    A.status[b] = st;
    A.best_J[b] = bestJ;
    A.best_idx[b] = bestI;
    if (A.dist) {
#pragma unroll
      for (int c = 0; c < DD; ++c) A.disturb[(long)c * B + b] = q[c];
      A.substep_idx[b] = sub;
    }
  }
}

// k_ticks for the regime every preset benchmark runs, around the hand-packed rollout (GenPk): float, MPC with the preset's
// diagonal R1 (its zero weights) and gamma == 1, no target, K = g * g a multiple of 256 with 64 % g == 0.  Same arithmetic as
// k_ticks - env_substeps, the rollout's operation sequence (GenPk::run), accum_update - and the same results bit for bit; what differs is the
// shell.  A wave owns `gpw` consecutive envs and keeps env e in LANE e: state, held action, accum, counters are loaded
// once (coalesced), the env step of a tick runs for all of the wave's envs at once (lane = env: one RK4 per wave and tick
// instead of one per env in all 64 lanes - in k_ticks the redundant env step is 40 % of a K = 256 tick), then the wave decides
// env after env: the env's state is read out of its lane (v_readlane), the 4 x 64 candidates are rolled out two per
// instruction, the argmin is DPP + readlane on the packed (cost, index) key instead of 6 x 4 ds_bpermute, the winner's
// action is regenerated from its index and written back into the env's lane.  Every field is stored once, coalesced.
template <typename Sys>
__global__ __launch_bounds__(256, 4) void k_ticks_pk(const TicksArgs<float> A, const KParams<float> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU;
  static_assert(DU == 2 && GenPk<Sys>::supported, "see GenPk");
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const int K = A.K, N = P.n_actor, gpw = A.gpw;
  const long env0 = wave * gpw;
  if (env0 >= B) return;  // wave-uniform
  const int ne = (int)((B - env0) < gpw ? (B - env0) : gpw);
  const bool mine = lane < ne;         // this lane holds env (env0 + lane)
  const long bm = env0 + (mine ? lane : 0);

  float x[DS], xp[DS], u[DU];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    x[c] = A.state[(long)c * B + bm];
    xp[c] = A.state_prev[(long)c * B + bm];
  }
#pragma unroll
  for (int c = 0; c < DU; ++c) u[c] = A.action[(long)c * B + bm];
  const auto pre = load_pre<Sys, float>(P, A.pars_env, bm);
  uint32_t st = A.status[bm];
  float accum = A.accum[bm];
  int32_t steps = A.step_idx[bm];
  float myJ = inf_r<float>();
  int myI = 0x7fffffff;

  for (int t = 0; t < A.T; ++t) {
    if (mine) env_substeps<Sys, float, false>(P, pre, A.n_sub, x, xp, u, st, accum);  // k_sim, lane = env
    for (int e = 0; e < ne; ++e) {  // the decision of env e, by the whole wave
      float ye[DS], xse[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) {
        ye[c] = readlane_r(x[c], e);
        xse[c] = P.ref_lag ? readlane_r(xp[c], e) : ye[c];  // rcg_control_tick's state_sys
      }
      const auto pre_e = A.pars_env ? Sys::template bcast<float>(pre, e) : pre;
      float bestJ = inf_r<float>();
      int bestI = lane;  // the lane's first candidate (tile 0): what an all-+inf lane keeps, as numpy.argmin keeps index 0
      float bestU[DU] = {0, 0};
      for (int tl = 0; tl < A.n_tiles; tl += 4)
        gen_multi_tiles<Sys, float, false, 4, true, true>(P, pre_e, N, K, A.grid_g, tl, lane, true, xse, ye, bestJ, bestI, bestU);
      const unsigned long long key = wave_min_u64(((unsigned long long)float_order_key(bestJ) << 32) | (unsigned)bestI);
      myJ = lane == e ? float_from_order_key((unsigned)(key >> 32)) : myJ;  // env e's result waits in the env's lane
      myI = lane == e ? (int)(unsigned)key : myI;
    }
    // receive_action (held until the next tick), upd_accum_obj, the tick counter - ONCE per tick for all envs of the wave,
    // lane == env (round 4 did this inside the env loop under `lane == e`: 25 instructions issued ne times per tick)
    if (mine) {
#ifdef RCG_AB_DIV
      gen_candidate<DU, float>(P, A.grid_g, myI, u);
#else
      {  // gen_candidate(myI) with shifts: g is a power of two in this regime (64 % g == 0)
        const int lg = 31 - __builtin_clz((unsigned)A.grid_g);
        const float den = (float)(A.grid_g > 1 ? A.grid_g - 1 : 1);
        u[0] = fma_r((float)(myI >> lg), (P.hi[0] - P.lo[0]) / den, P.lo[0]);
        u[1] = fma_r((float)(myI & (A.grid_g - 1)), (P.hi[1] - P.lo[1]) / den, P.lo[1]);
      }
#endif
      if (!P.accum_every_substep) accum = accum_update<Sys, false, float>(P, x, u, accum);
      steps += 1;
    }
  }
  if (mine) {  // one coalesced write per field for the envs of this wave
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      A.state[(long)c * B + bm] = x[c];
      A.state_prev[(long)c * B + bm] = xp[c];
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) A.action[(long)c * B + bm] = u[c];
    A.accum[bm] = accum;
    A.step_idx[bm] = steps;
    A.status[bm] = st;
    A.best_J[bm] = myJ;
    A.best_idx[bm] = myI;
  }
}

// ---------------------------------------------------------------------------------------------
// k_ticks_mem: T RQL / SQL ticks in ONE launch
// ---------------------------------------------------------------------------------------------
// An RQL / SQL tick is two launches - k_critic_fit (env step + buffer push + critic fit, lane = env) and the decision
// kernel - that exchange everything through the handle's tensors, and an env's tick depends on nothing but that env.  So
// the wave that decides for env(s) [wave G, wave G + G) can run both launches' bodies for THOSE envs back to back, T times:
// phase 1, the lanes that stand for the wave's envs run critic_update_env (the body of k_critic_fit); phase 2, the whole
// wave runs actor_wave (the body of k_actor).  Same functions on the same memory: every field ends bit-identical to T
// single ticks.  Between the phases the wave's stores have to be visible to its own later loads through the CU's vector
// L1: a workgroup-scope release / acquire pair (a wait for the stores and an L1 invalidate of ~100 cycles: no other CU is
// involved, the wave talks to itself).  Generated candidates; with a caller's tensor the single ticks of a critic-mode
// handle run on k_actor_dma, whose rollouts round differently, so those stay per-tick launches (rcg_control_tick_n).
template <typename real>
struct TicksMemArgs {
  ActorArgs<real> A;
  FitArgs<real> F;
  int T;       // ticks
  int tick0;   // control ticks of the episode issued before this launch (the critic period's phase)
  int every;   // critic_every_ticks (>= 1)
};

// ML (round 5): the critic structures with >= 20 weights, whose single ticks fit with FOUR LANES PER ENV (k_critic_fit_ml): the
// wave's envs (G <= 16) take the quads 0 .. G - 1 of the wave for phase 1 - critic_update_env_ml, the body of that kernel.
// STREAM (round 5): the decision phase walks a caller's tensor (actor_wave's streamed form: the wave's tile staged through
// LDS every tick - at the batch sizes this launch is for the tensor stays in L2 / the Infinity Cache).
template <typename Sys, typename real, int CS, int MAXM, bool TGT, bool ML = false, bool STREAM = false>
__global__ __launch_bounds__(256) void k_ticks_mem(const TicksMemArgs<real> M, const KParams<double> P64,
                                                   const KParams<real> P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_tm[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const int G = M.A.G;
  if (wave * G >= B) return;  // wave-uniform: every wave that stays runs all T ticks and exits
  const int envs_here = (int)((B - wave * G) < G ? (B - wave * G) : G);
  FitArgs<real> F = M.F;
  const int bs = P.buffer_size;
  int ring = 1;  // the buffers are rings inside the launch (FitArgs::ring): tick t overwrites physical row t mod buffer_size
  for (int t = 0; t < M.T; ++t) {
    F.do_fit = ((M.tick0 + t + 1) % M.every) == 0 ? 1 : 0;  // fits on ticks every - 1, 2 every - 1, ... of the episode
    F.ring = ring;
    ring = ring == bs ? 1 : ring + 1;
    if constexpr (ML) {
      if (lane < FIT_L * envs_here)  // (whole quads: the DPP exchanges of the four-lane walk stay inside a quad)
        critic_update_env_ml<Sys, real, CS, MAXM>(F, P64, P, wave * G + (lane / FIT_L), lane & (FIT_L - 1));
    } else {
      if (lane < envs_here) critic_update_env<Sys, real, CS, MAXM>(F, P64, P, wave * G + lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if constexpr (STREAM) {
      real* const lds = reinterpret_cast<real*>(smem_raw_tm) + (size_t)wave_in_wg * 64 * (P.n_actor * Sys::DU);
      actor_wave<Sys, real, true, TGT, true>(M.A, P, wave, lds, t > 0 && M.A.n_tiles == 1);  // one tile per wave: it stays
    } else {
      actor_wave<Sys, real, true, TGT, false>(M.A, P, wave, nullptr);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // the rows back into place: by the lane that stored them (its own stores, its own loads)
  if constexpr (ML) {
    if (lane < FIT_L * envs_here && (lane & (FIT_L - 1)) == 0) critic_ring_unrotate<Sys, real>(F, P, wave * G + (lane / FIT_L), M.T);
  } else {
    if (lane < envs_here) critic_ring_unrotate<Sys, real>(F, P, wave * G + lane, M.T);
  }
}

}  // namespace rcg
