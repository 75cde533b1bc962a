// rcg_sysops.hpp - host launchers of every system-templated kernel, instantiated once per environment
// by rcg_sys_<system>.hip through make_vtable<Sys>().
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "rcg_actor_dma.hpp"
#include "rcg_actor_dma_packed.hpp"
#include "rcg_critic_fit_ml.hpp"
#include "rcg_actor_opt.hpp"
#include "rcg_critic_fit.hpp"
#include "rcg_critic_fit_gen.hpp"
#include "rcg_disturb.hpp"
#include "rcg_handle.hpp"
#include "rcg_loop.hpp"
#include "rcg_nominal.hpp"
#include "rcg_search.hpp"
#include "rcg_ticks.hpp"

namespace rcg {

// f(real{}) for the handle's dtype
template <typename F>
static int by_dtype(rcg_handle* h, F&& f) {
  return h->cfg.dtype == RCG_F64 ? f(double{}) : f(float{});
}

template <typename Sys>
static int op_rhs(rcg_handle* h, const void* state, const void* action, void* dstate, void* clipped, int32_t n,
                  int32_t clip) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    const real* pe = (h->f[RCG_FIELD_PARS] && n == h->cfg.batch) ? (const real*)h->f[RCG_FIELD_PARS] : nullptr;
    hipLaunchKernelGGL((k_rhs<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)state,
                       (const real*)action, (real*)dstate, (real*)clipped, pe, (long)n, (int)clip, params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

static inline DisturbPars disturb_pars(const rcg_handle* h) {
  DisturbPars D;
  for (int k = 0; k < 2; ++k) {
    D.sigma[k] = h->cfg.pars_disturb[k];
    D.mu[k] = h->cfg.pars_disturb[2 + k];
    D.tau[k] = h->cfg.pars_disturb[4 + k];
  }
  D.seed = h->cfg.seed;
  D.env_id_base = h->cfg.env_id_base;
  return D;
}

// closed_loop_rhs on the full state [state, disturb], noise given (rcg_rhs_full)
template <typename Sys>
static int op_rhs_full(rcg_handle* h, const void* state, const void* disturb, const void* action, const void* xi,
                       void* dstate, void* ddisturb, void* clipped, int32_t n, int32_t clip) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    const real* pe = (h->f[RCG_FIELD_PARS] && n == h->cfg.batch) ? (const real*)h->f[RCG_FIELD_PARS] : nullptr;
    hipLaunchKernelGGL((k_rhs_full<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)state,
                       (const real*)disturb, (const real*)action, (const real*)xi, (real*)dstate, (real*)ddisturb,
                       (real*)clipped, pe, (long)n, (int)clip, disturb_pars(h), params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

template <typename Sys>
static int op_stage_obj(rcg_handle* h, const void* obs, const void* act, void* out, int32_t n) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    hipLaunchKernelGGL((k_stage_obj<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)obs,
                       (const real*)act, (real*)out, (long)n, params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// rcg_loop_step's glue kernel (rcg_loop.hpp): [set ACTION from the pinned host buffer] -> [sim step] -> [stage cost + pack]
template <typename real>
static void fill_loop_args(rcg_handle* h, LoopArgs<real>& A, const double* act_in, int32_t n_substeps, int32_t do_sim,
                           int32_t do_tail, int32_t decided, int32_t dc, double* out, double* flag, double seq) {
  memset(&A, 0, sizeof A);
  A.sim.state = (real*)h->f[RCG_FIELD_STATE];
  A.sim.state_prev = (real*)h->f[RCG_FIELD_STATE_PREV];
  A.sim.action = (const real*)h->f[RCG_FIELD_ACTION];
  A.sim.pars_env = (const real*)h->f[RCG_FIELD_PARS];
  A.sim.accum = (real*)h->f[RCG_FIELD_ACCUM];
  A.sim.status = (uint32_t*)h->f[RCG_FIELD_STATUS];
  A.sim.n_sub = n_substeps;
  A.action = (real*)h->f[RCG_FIELD_ACTION];
  A.act_in = act_in;
  A.best_J = (const real*)h->f[RCG_FIELD_BEST_J];
  A.w = (const real*)h->f[RCG_FIELD_W_CRITIC];
  A.out = out;
  A.flag = flag;
  A.seq = seq;
  A.do_sim = do_sim;
  A.do_tail = do_tail;
  A.decided = decided;
  A.dc = dc;
}

template <typename Sys>
static int op_loop(rcg_handle* h, const double* act_in, int32_t n_substeps, int32_t do_sim, int32_t do_tail, int32_t decided,
                   int32_t dc, double* out, double* flag, double seq) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    LoopArgs<real> A;
    fill_loop_args<real>(h, A, act_in, n_substeps, do_sim, do_tail, decided, dc, out, flag, seq);
    hipLaunchKernelGGL((k_loop<Sys, real>), dim3(blocks_for(h->cfg.batch, 64)), dim3(64), 0, h->stream, A, params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

template <typename Sys>
static int op_critic(rcg_handle* h, const void* obs, const void* act, const void* w, void* out, int32_t n) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    hipLaunchKernelGGL((k_critic<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)obs,
                       (const real*)act, (const real*)w, (real*)out, (long)n, params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

template <typename Sys>
static int op_critic_cost(rcg_handle* h, const void* w, void* Jc) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    hipLaunchKernelGGL((k_critic_cost<Sys, real>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, h->stream,
                       w ? (const real*)w : (const real*)h->f[RCG_FIELD_W_CRITIC], (const real*)h->f[RCG_FIELD_W_PREV],
                       (const real*)h->f[RCG_FIELD_OBS_BUF], (const real*)h->f[RCG_FIELD_ACT_BUF], (real*)Jc,
                       params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

template <typename Sys>
int op_sim_step(rcg_handle* h, int32_t n_substeps) {
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    SimArgs<real> A;
    A.state = (real*)h->f[RCG_FIELD_STATE];
    A.state_prev = (real*)h->f[RCG_FIELD_STATE_PREV];
    A.action = (const real*)h->f[RCG_FIELD_ACTION];
    A.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    A.accum = (real*)h->f[RCG_FIELD_ACCUM];
    A.status = (uint32_t*)h->f[RCG_FIELD_STATUS];
    A.n_sub = n_substeps;
    ProfScope prof_scope(h, RCG_KERNEL_SIM);
    if (h->cfg.flags & RCG_FLAG_DISTURB) {  // full state [state, disturb] (rcg_disturb.hpp)
      SimDistArgs<real> D;
      D.S = A;
      D.disturb = (real*)h->f[RCG_FIELD_DISTURB];
      D.substep_idx = (int32_t*)h->f[RCG_FIELD_SUBSTEP_IDX];
      D.episode_idx = (const int32_t*)h->f[RCG_FIELD_EPISODE_IDX];
      D.D = disturb_pars(h);
      if (h->cfg.flags & RCG_FLAG_HAS_TARGET)
        RCG_LAUNCH(h, (k_sim_dist<Sys, real, true>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, D,
                           params<real>(h));
      else
        RCG_LAUNCH(h, (k_sim_dist<Sys, real, false>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, D,
                           params<real>(h));
      note_launch(h, RCG_KERNEL_SIM, RCG_KID_SIM_DIST, 0, 64);
      HIPCHK(h, hipGetLastError());
      return (int)RCG_OK;
    }
    constexpr long VEC = 16 / (long)sizeof(real);
    // 16 B per lane and component (k_sim_v) pays once the launch is bandwidth- rather than latency-bound, and only for
    // the light dynamics: 2tank at 2^24 envs 4.7 -> 5.6 TB/s; the robots' RK4 (four accurate sin/cos per substep) with
    // VEC envs per lane needs 112 VGPRs instead of 70 and got SLOWER (5.76 -> 5.2 TB/s), so they stay on k_sim
    note_launch(h, RCG_KERNEL_SIM, RCG_KID_SIM, 0, 64);
    if (Sys::DS <= 2 && h->cfg.batch % VEC == 0 && h->cfg.batch >= (1 << 18)) {
      const dim3 gridv(blocks_for(h->cfg.batch / VEC));
      if (h->cfg.flags & RCG_FLAG_HAS_TARGET)
        RCG_LAUNCH(h, (k_sim_v<Sys, real, true>), gridv, dim3(256), 0, A, params<real>(h));
      else
        RCG_LAUNCH(h, (k_sim_v<Sys, real, false>), gridv, dim3(256), 0, A, params<real>(h));
      note_launch(h, RCG_KERNEL_SIM, RCG_KID_SIM_V, 0, 64 * (int)VEC);
    } else if (h->cfg.flags & RCG_FLAG_HAS_TARGET)
      RCG_LAUNCH(h, (k_sim<Sys, real, true>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, A,
                         params<real>(h));
    else
      RCG_LAUNCH(h, (k_sim<Sys, real, false>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, A,
                         params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

#ifdef RCG_DEV
static int fit_lanes_knob();  // (DevKnobs, below)
#endif

// The exact-m instance (Ncritic - 1 <= 3: every preset).  One lane per env (k_critic_fit) for the structures with fewer than
// kFitLanesMinDc weights; four lanes per env (k_critic_fit_ml, rcg_critic_fit_ml.hpp) from there on: the one-lane walk of a
// 35-weight structure takes 2.3 ms for 32 768 envs, the four-lane one 0.41 (profiles/r04_fit_four_lanes.txt); with few
// weights the four-lane form is the slower one (configs[2], 6 weights: 65 -> 88 us).  Returns which form ran.
// Round 6: from 9 weights (was 20) - the structures with 9-17 weights measured 1.2-1.3 x faster on four lanes in round 4 (2tank
// quad-lin 83 -> 66 us, 3wrobotNI quad-mix 145 -> 116, 3wrobotNI quadratic 238 -> 196; 7 weights: 61 -> 59, 6 weights: 65 -> 89).
constexpr int kFitLanesMinDc = 9;
template <typename Sys, typename real, int CS>
static bool launch_fit3(rcg_handle* h, const FitArgs<real>& F, bool force_ml) {
  constexpr int DC = CriticDim<CS, Sys::DS, Sys::DU>::value;
  const long n_env = h->sub_hi > 0 ? h->sub_hi - h->sub_lo : h->cfg.batch;
  const dim3 block(64), grid(blocks_for(n_env, 64)), grid_ml(blocks_for(n_env, 64 / FIT_L));
  if constexpr (DC >= kFitLanesMinDc) {
    RCG_LAUNCH(h, (k_critic_fit_ml<Sys, real, CS, 3>), grid_ml, block, 0, F, h->p64, params<real>(h));
    return true;
  } else {
#ifdef RCG_DEV
    if (force_ml) {
      RCG_LAUNCH(h, (k_critic_fit_ml<Sys, real, CS, 3>), grid_ml, block, 0, F, h->p64, params<real>(h));
      return true;
    }
#endif
    RCG_LAUNCH(h, (k_critic_fit<Sys, real, CS, 3>), grid, block, 0, F, h->p64, params<real>(h));
    return false;
  }
}

template <typename Sys>
int op_critic_update(rcg_handle* h, int32_t n_substeps, int32_t do_push, int32_t do_fit) {
  const int m = h->cfg.n_critic - 1;
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    ProfScope prof_scope(h, RCG_KERNEL_CRITIC);
    FitArgs<real> F;
    memset(&F, 0, sizeof F);
    F.w_critic = (real*)h->f[RCG_FIELD_W_CRITIC];
    F.w_prev = (real*)h->f[RCG_FIELD_W_PREV];
    F.obs_buf = (real*)h->f[RCG_FIELD_OBS_BUF];
    F.act_buf = (real*)h->f[RCG_FIELD_ACT_BUF];
    F.wcfg = reinterpret_cast<const double*>((unsigned char*)h->d_const + kConstW);
    F.do_sim = n_substeps > 0;
    F.do_push = do_push;
    F.do_fit = do_fit;
    F.state = (const real*)h->f[RCG_FIELD_STATE];
    F.action = (const real*)h->f[RCG_FIELD_ACTION];
    F.sim.state = (real*)h->f[RCG_FIELD_STATE];
    F.sim.state_prev = (real*)h->f[RCG_FIELD_STATE_PREV];
    F.sim.action = (const real*)h->f[RCG_FIELD_ACTION];
    F.sim.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    F.sim.accum = (real*)h->f[RCG_FIELD_ACCUM];
    F.sim.status = (uint32_t*)h->f[RCG_FIELD_STATUS];
    F.sim.n_sub = n_substeps;
    F.env_lo = h->sub_lo;  // (a half of a split tick; 0, 0: the whole batch)
    F.env_hi = h->sub_hi;
    const dim3 grid(blocks_for(h->sub_hi > 0 ? h->sub_hi - h->sub_lo : h->cfg.batch, 64)), block(64);
    bool fit_ml = false;
#ifdef RCG_DEV
    const bool force_ml = fit_lanes_knob() == FIT_L;  // RCG_FIT_LANES=4: the four-lane form for every structure (experiments)
#else
    const bool force_ml = false;
#endif
    // more TD rows than the register kernels hold (Ncritic - 1 > 8; the reference only clips Ncritic to buffer_size - 1,
    // controllers.py:1015): k_critic_fit_gen with the env's stack and factor in a scratch tensor of the handle, allocated on
    // first use (rcg_critic_fit_gen.hpp)
    const bool gen = m > kFitMaxRows;
    if (gen && do_fit) {
      const size_t need = (size_t)fit_gen_scratch_doubles(m, h->dc) * (size_t)h->cfg.batch * sizeof(double);
      if (h->fit_scratch_bytes < need) {
        if (h->fit_scratch) {
          HIPCHK(h, hipStreamSynchronize(h->stream));
          HIPCHK(h, hipFree(h->fit_scratch));
          h->fit_scratch = nullptr;
          h->fit_scratch_bytes = 0;
        }
        if (hipMalloc(&h->fit_scratch, need) != hipSuccess) {
          (void)hipGetLastError();
          return rcg_fail(h, RCG_ERR_HIP, "critic fit with %d TD rows: cannot allocate %zu bytes of scratch for %d envs", m, need,
                          h->cfg.batch);
        }
        h->fit_scratch_bytes = need;
      }
    }
#define RCG_FIT(CS)                                                                                                    \
  do {                                                                                                                 \
    if (m <= 3)                                                                                                        \
      fit_ml = launch_fit3<Sys, real, CS>(h, F, force_ml);                                                             \
    else if (!gen)                                                                                                     \
      RCG_LAUNCH(h, (k_critic_fit<Sys, real, CS, kFitMaxRows>), grid, block, 0, F, h->p64,             \
                         params<real>(h));                                                                             \
    else                                                                                                               \
      RCG_LAUNCH(h, (k_critic_fit_gen<Sys, real, CS>), grid, block, 0, F, h->p64, params<real>(h),                     \
                 (double*)h->fit_scratch);                                                                             \
  } while (0)
    switch (h->cfg.critic_struct) {
      case RCG_CRITIC_QUAD_LIN: RCG_FIT(RCG_CRITIC_QUAD_LIN); break;
      case RCG_CRITIC_QUADRATIC: RCG_FIT(RCG_CRITIC_QUADRATIC); break;
      case RCG_CRITIC_QUAD_NOMIX: RCG_FIT(RCG_CRITIC_QUAD_NOMIX); break;
      default: RCG_FIT(RCG_CRITIC_QUAD_MIX); break;
    }
#undef RCG_FIT
    note_launch(h, RCG_KERNEL_CRITIC, RCG_KID_CRITIC_FIT,
                h->cfg.critic_struct + 16 * (m <= 3 ? 3 : (gen ? 0 : kFitMaxRows)) + (F.do_sim ? 256 : 0) + (do_fit ? 512 : 0) +
                    (fit_ml ? 1024 : 0) + (gen ? 2048 : 0),
                fit_ml ? 16 : 64);
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// Development knobs of the actor launcher.  They select between variants of the same computation for A/B measurements
// (none changes results beyond rounding) and exist ONLY in -DRCG_DEV builds (`make dev`, librcg_dev.so, chosen by a tool
// with rcognita_amd._native.use_library): the production library never reads the environment - its schedule is the one measured and shipped.
//   RCG_ACTOR_KERNEL=plain  force k_actor instead of k_actor_dma      RCG_GPW=<n>  envs per persistent wave
//   RCG_DBG=<bits>          1 skip the rollout, 2 skip argmin + writes, 4 skip env-state loads - timing only, wrong results
//   RCG_NO_G1=1             no gamma == 1 specialisation               RCG_DMA_MPC_ONLY=1  RQL / SQL on k_actor
//   RCG_PER_CU=2|4|8, RCG_LDS_PAD=<bytes>|-1   resident blocks per CU of k_actor_dma (via its LDS request)
//   RCG_PLAIN_LDS=<bytes>   residency cap for the streamed k_actor      RCG_NO_GEN_MULTI=1  generated tiles one at a time
//   RCG_NO_PACK=1           streamed K <= 32 without the packed-tile instances (k_actor_dma from RCG_DMA_MINK on, else k_actor)
//   RCG_DMA_MINK=<k>        fewest candidates per env k_actor_dma serves as one ragged tile (default 33; RQL / SQL: min(k, 20))
//   RCG_NO_PK=1             generated grid / k_ticks without the hand-packed instances (scalar-form rollouts, same bits)
//   RCG_NO_TICK_FUSE=1      generated-grid tick as k_sim + k_actor's packed instance instead of k_ticks_pk
//   RCG_FIT_LANES=4         the critic fit with four lanes per env (k_critic_fit_ml; results differ on degenerate stacks: an experiment)
// tests/test_hip_knobs.py checks (on librcg_dev.so) that the scheduling variants reproduce the default launch bit for
// bit, and that the production library ignores every one of them; bench.py refuses to run with any RCG_* variable set.
struct DevKnobs {
  int dbg = 0;
  bool force_plain = false, no_g1 = false;
  long gpw = 0;
  long lds_pad = 0;  // RCG_LDS_PAD=<bytes>: extra dynamic LDS per block, caps the resident blocks per CU (-1: no cap)
  int per_cu = 0;    // RCG_PER_CU=2|4|8: resident blocks per CU for k_actor_dma (0: by row length)
  // RCG_PLAIN_LDS=<bytes>: minimum dynamic-LDS request of the streamed k_actor, i.e. a residency cap.  Unlike
  // k_actor_dma, k_actor has no direct-to-LDS prefetch and hides latency with occupancy: 4 blocks/CU measured 7 %
  // slower than 8, 2 blocks/CU 68 % slower (configs[2], SQL, streamed) - the default is no cap.
  long plain_lds = 0;
  bool mpc_only = false;  // RCG_DMA_MPC_ONLY=1: RQL and SQL go to k_actor (A/B against the critic instances)
  bool no_gen_multi = false;  // RCG_NO_GEN_MULTI=1: generated tiles one at a time (no shared sub-trajectory)
  bool no_pack = false;
  int fit_lanes = 0;  // RCG_FIT_LANES=4: the critic fit with four lanes per env (k_critic_fit_ml), m <= 3
  int dma_min_k = 33;  // RCG_DMA_MINK=<k>: fewest candidates per env served by k_actor_dma (one ragged tile below 64)
  bool no_tick_fuse = false;  // RCG_NO_TICK_FUSE=1: generated-grid tick as k_sim + k_actor (packed instance) instead of k_ticks_pk
  bool no_pk = false;  // RCG_NO_PK=1: generated grid on the instances that carry every variant (A/B of the packed rollout)
};
static inline const DevKnobs& dev_knobs() {
  static const DevKnobs k = [] {
    DevKnobs v;
#ifdef RCG_DEV
    if (const char* e = getenv("RCG_DBG")) v.dbg = atoi(e);
    if (const char* e = getenv("RCG_ACTOR_KERNEL")) v.force_plain = !strcmp(e, "plain");
    if (const char* e = getenv("RCG_GPW")) v.gpw = atol(e);
    if (const char* e = getenv("RCG_LDS_PAD")) v.lds_pad = atol(e);
    if (const char* e = getenv("RCG_PER_CU")) v.per_cu = atoi(e);
    if (const char* e = getenv("RCG_PLAIN_LDS")) v.plain_lds = atol(e);
    v.mpc_only = getenv("RCG_DMA_MPC_ONLY") != nullptr;
    v.no_gen_multi = getenv("RCG_NO_GEN_MULTI") != nullptr;
    v.no_g1 = getenv("RCG_NO_G1") != nullptr;
    v.no_pack = getenv("RCG_NO_PACK") != nullptr;
    v.no_pk = getenv("RCG_NO_PK") != nullptr;
    v.no_tick_fuse = getenv("RCG_NO_TICK_FUSE") != nullptr;
    if (const char* e = getenv("RCG_DMA_MINK")) v.dma_min_k = atoi(e);
    if (const char* e = getenv("RCG_FIT_LANES")) v.fit_lanes = atoi(e);
#endif
    return v;
  }();
  return k;
}
#ifdef RCG_DEV
static int fit_lanes_knob() { return dev_knobs().fit_lanes; }
#endif

template <typename Sys>
int op_ticks(rcg_handle* h, int32_t T, int32_t K, const void* cand);

// ---- k_actor / k_actor_dma ---------------------------------------------------------------------
// `sim_first`: rcg_control_tick (MPC) - run the env step of the tick before the decision.
template <typename Sys, typename real>
static int launch_actor(rcg_handle* h, const char* who, const void* cand, int K, const void* obs,
                        const void* state_sys, const void* w, void* J, void* action, void* best_J, int32_t* best_idx,
                        bool tick, bool sim_first) {
  constexpr int DU = Sys::DU;
  const rcg_cfg& c = h->cfg;
  if (K < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: K must be >= 1", who);
  ActorArgs<real> A;
  memset(&A, 0, sizeof A);
  A.cand = (const real*)cand;
  A.obs = obs ? (const real*)obs : (const real*)h->f[RCG_FIELD_STATE];
  if (state_sys)
    A.state_sys = (const real*)state_sys;
  else if (obs)
    A.state_sys = (const real*)obs;
  else
    A.state_sys = (const real*)h->f[(tick && (c.flags & RCG_FLAG_REF_LAG)) ? RCG_FIELD_STATE_PREV : RCG_FIELD_STATE];
  A.pars_env = (const real*)h->f[RCG_FIELD_PARS];
  A.w = w ? (const real*)w : (const real*)h->f[RCG_FIELD_W_CRITIC];
  if (c.mode != RCG_MODE_MPC && !A.w)
    return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: RQL/SQL need critic weights (buffer_size > 0 or an explicit w)", who);
  A.J = (real*)J;
  A.action_out = (real*)action;
  A.best_J = (real*)best_J;
  A.best_idx = best_idx;
  A.accum = (tick && !(c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP)) ? (real*)h->f[RCG_FIELD_ACCUM] : nullptr;
  A.step_idx = tick ? (int32_t*)h->f[RCG_FIELD_STEP_IDX] : nullptr;
  A.K = K;
  if (K >= 64) {
    A.Kp = 64;
    A.G = 1;
    A.n_tiles = (K + 63) / 64;
  } else {
    int kp = 1;
    while (kp < K) kp <<= 1;
    A.Kp = kp;
    A.G = 64 / kp;
    A.n_tiles = 1;
  }
  A.grid_g = 0;
  A.no_multi = dev_knobs().no_gen_multi ? 1 : 0;
  if (!cand) {
    if (DU == 1) {
      A.grid_g = K;
    } else {
      int g = (int)std::floor(std::sqrt((double)K) + 1e-9);
      if (g * g != K)
        return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: generated grid for du = 2 needs a square K (got %d)", who, K);
      A.grid_g = g;
    }
  }
  const int R = c.n_actor * DU;
  const size_t row_bytes = (size_t)R * sizeof(real);
  A.vec_ok = (cand && row_bytes % 16 == 0 && ((uintptr_t)cand % 16) == 0) ? 1 : 0;
  const long B = c.batch;
  const long n_waves = (B + A.G - 1) / A.G;
  int wpb = 4;  // waves per workgroup
  // rows beyond RCG_MAX_ROW reals (the reference's horizon is unbounded, controllers.py:965): no LDS tile - the generic
  // instance's DIRECT form, every lane walking its row straight from HBM (rcg_kernels.hpp::actor_wave)
  const bool long_row = cand && R > RCG_MAX_ROW;
  size_t lds_per_wave = (cand && !long_row) ? 64 * row_bytes : 0;
  while (wpb > 1 && lds_per_wave * wpb > 64 * 1024) wpb >>= 1;
  size_t lds = lds_per_wave * wpb;
  if (cand && (size_t)dev_knobs().plain_lds > lds) lds = (size_t)dev_knobs().plain_lds;  // residency experiments
  const unsigned blocks = (unsigned)((n_waves + wpb - 1) / wpb);
  const KParams<real>& P = params<real>(h);
  const bool generic = !(c.mode == RCG_MODE_MPC && P.stage_kind == 0);
  const bool tgt = (c.flags & RCG_FLAG_HAS_TARGET) != 0;

  // Production shape -> k_actor_dma (rcg_actor_dma.hpp): streamed candidates, K >= 33 with K * R * esz % 16 == 0 (33 .. 63: one
  // ragged tile per env - K = 48: 4.6 TB/s against 2.9 on k_actor, K = 36: 3.6 against 2.3 (RQL: 3.1 x, profiles/r04_ab_min_k.txt); at K <= 32 k_actor, which packs 64 / K envs into a tile,
  // is faster: 3.7 against 3.4 TB/s at K = 32, 3.4 against 1.75 at K = 16), diagonal quadratic
  // stage cost, the preset's observation target (an instance that subtracts a target also serves a handle without one: its
  // target is all zeros, y - 0 = y exactly); rows of <= 40 reals; MPC / RQL / SQL in f32 and f64.
  const DevKnobs& knobs = dev_knobs();
  A.dbg = knobs.dbg;
  constexpr size_t esz = sizeof(real);
  const size_t tile = (size_t)64 * dma_rpl(R, (int)esz) * R * esz;  // one wave's LDS tile (64 x rows-per-lane rows)
  const bool mode_ok = c.mode == RCG_MODE_MPC || !knobs.mpc_only;
  // MPC with a stage cost no preset has (a full R1, the biquadratic structure; with or without an observation target): the
  // instances DMA_MPC_GEND / DMA_MPC_GENF (round 6; until then k_actor's plain staging, 0.09-0.37 of the
  // HBM peak at the C2 shape: profiles/r06_generic_stream_probe_*.txt)
  // (a diagonal quadratic cost with a target on a robot stays on k_actor's target instance: 0.78 of the peak there against 0.75 on
  // DMA_MPC_GEND, and its gamma == 1 accumulation - per component - is the one k_ticks re-walks the rows with)
  const bool std_cost = P.stage_kind == 0 && (tgt == Sys::TGT || !tgt);
  const bool gen_cost = c.mode == RCG_MODE_MPC && P.stage_kind != 0 && !knobs.force_plain;
  // RQL with such a stage cost (or a target its system's preset has not): DMA_RQL_GEN_* (stage_any per step); SQL has no stage
  // cost inside the rollout - its instances serve any stage structure (only upd_accum_obj sees it)
  const bool gen_rql = c.mode == RCG_MODE_RQL && !std_cost && !knobs.force_plain;
  const bool sql_any = c.mode == RCG_MODE_SQL && (tgt == Sys::TGT || !tgt);
  int variant;
  if (gen_cost)
    variant = (P.stage_kind & STAGE_FULL) ? DMA_MPC_GENF : DMA_MPC_GEND;
  else if (c.mode == RCG_MODE_MPC)
    variant = (c.gamma == 1.0 && !knobs.no_g1) ? DMA_MPC_G1 : DMA_MPC;  // per-component accumulation when gamma == 1
  else if (c.mode == RCG_MODE_RQL)
    variant = (gen_rql ? DMA_RQL_GEN_0 : DMA_RQL_0) + c.critic_struct;
  else
    variant = DMA_SQL_0 + c.critic_struct;
  const size_t wslot = (size_t)4 * dma_wslot((int)esz, variant, Sys::DS, DU);  // critic weights parked in LDS (> 9 of them)
  // (an env's rows must be a whole number of 16-byte pieces, K * R * esz % 16 == 0 - any K for rows of 16 n bytes such as C2's
  // 80, every 4th K for the shortest rows: then every env starts 16-B aligned and a ragged last tile ends on a piece)
  const bool slab16 = ((size_t)K * row_bytes) % 16 == 0;
  // (RQL / SQL without a packed instance - f64 with more than 18 weights: one ragged tile already from K = 20, where it
  // overtakes k_actor: 66 against 105 us at K = 24, 65 against 50 at K = 16; profiles/r04_ab_min_k.txt)
  const int dma_min_k = (c.mode != RCG_MODE_MPC && knobs.dma_min_k > 20) ? 20 : knobs.dma_min_k;
  const bool dma_ok = cand && ((uintptr_t)cand % 16) == 0 && K >= dma_min_k && slab16 && R <= dma_max_row<real>() &&
                      (std_cost || gen_cost || gen_rql || sql_any) && mode_ok && !knobs.force_plain &&
                      // J staging must fit next to the tiles (one block per CU then)
                      !(A.J && 4 * tile + wslot + 4 * esz * K > (size_t)160 * 1024);
  // Few candidates per env (4 <= K <= 32, whole 16-byte pieces per env) -> k_actor_dma_packed (rcg_actor_dma_packed.hpp): 64 / K envs
  // share a DMA tile (MPC; RQL / SQL with at most 36 dwords of critic weights - otherwise the launcher below finds no
  // instance and the tick goes on to k_actor_dma / k_actor).  J staging (operator mode) must fit next to the four tiles.
  const int pack_g = (K >= 4 && K <= 32) ? 64 / K : 0;  // envs per tile
  const bool pack_ok = cand && ((uintptr_t)cand % 16) == 0 && pack_g >= 2 && slab16 && R <= dma_max_row<real>() &&
                       P.stage_kind == 0 && mode_ok && (tgt == Sys::TGT || !tgt) && !knobs.force_plain && !knobs.no_pack;
  // rcg_control_tick with the generated grid in the regime of the hand-packed rollout: env step and decision in ONE launch
  // (k_ticks_pk with T = 1 - what rcg_control_ticks runs, so the two entry points cannot differ by a bit)
  if (h->probe == 1) {  // rcg_control_tick asking, before it launches anything, whether this tick's decision runs on k_actor_dma
    h->probe = (dma_ok && !pack_ok) ? 3 : 2;
    return RCG_OK;
  }
  if (h->sub_hi > 0 && !(dma_ok && !pack_ok))
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "%s: a split tick needs the k_actor_dma shape", who);
  if constexpr (std::is_same<real, float>::value && GenPk<Sys>::supported && GenPk<Sys>::fuse_tick) {
    if (tick && sim_first && !cand && !generic && !tgt && c.gamma == 1.0 && Sys::ZW_PRESET != 0u &&
        (P.zero_w & Sys::ZW_PRESET) == Sys::ZW_PRESET && K >= 256 && A.n_tiles % 4 == 0 && A.grid_g > 0 &&
        (64 % A.grid_g) == 0 && !A.no_multi && !knobs.no_pk && !knobs.no_tick_fuse && !(c.flags & RCG_FLAG_DISTURB) && !obs &&
        !state_sys &&
        action == h->f[RCG_FIELD_ACTION] && best_J == h->f[RCG_FIELD_BEST_J] && (void*)best_idx == h->f[RCG_FIELD_BEST_IDX])
      return op_ticks<Sys>(h, 1, K, nullptr);
  }
  // The env step of the tick (Simulator.sim_step) precedes the decision: its own launch (k_sim, 6.8 us at C2) - except in front
  // of k_actor_dma_packed, whose launches are short enough (11-31 us) for the k_sim launch and the gap behind it to be 15-20 %
  // of the tick: there the wave steps its own envs in its prologue (rcg_actor_dma_packed.hpp)
  const bool fuse_sim = pack_ok && tick && sim_first && !obs && !state_sys && !A.J &&
                        !(c.flags & (RCG_FLAG_DISTURB | RCG_FLAG_ACCUM_EVERY_SUBSTEP)) && !knobs.no_tick_fuse;
  if (sim_first && !fuse_sim) {
    int rc = op_sim_step<Sys>(h, c.substeps_per_tick);
    if (rc) return rc;
  }
  ProfScope prof_scope(h, RCG_KERNEL_ACTOR);
  if (pack_ok) {
    // a wave owns gpw = G * 2^n <= 64 consecutive envs (their results wait in its lanes); these launches are small (K = 16,
    // B = 65536, Nactor = 10: 84 MB), so the grid is kept at >= 4096 waves and 4 blocks per CU stay resident
    long gpw = pack_g;
    while (gpw * 2 <= 64 && B / (gpw * 2) >= 4096) gpw *= 2;
    if (knobs.gpw > 0 && knobs.gpw % pack_g == 0 && knobs.gpw <= 64) gpw = knobs.gpw;  // (dev build only)
    const long pw = (B + gpw - 1) / gpw;
    const dim3 grid((unsigned)((pw + 3) / 4)), block(256);
    const size_t full_tile = (size_t)64 * R * esz;
    size_t lds_req = 4 * full_tile + (A.J ? 4 * esz * (size_t)gpw * K : 0) + (fuse_sim ? 4 * esz * 2 * Sys::DS * 64 : 0);
    const size_t cap = knobs.per_cu == 2 ? (size_t)56 * 1024 : (knobs.per_cu == 8 ? 0 : (size_t)36 * 1024);
    if (lds_req < cap) lds_req = cap;  // 4 resident blocks per CU (dev build: RCG_PER_CU = 2 | 8)
    const ProfPair pp = prof_take(h);  // a due ProfScope's pair travels in the dispatch
    ActorArgs<real> Ap = A;
    Ap.G = pack_g;
    Ap.gpw = (int)gpw;
    Ap.jwave = 1;
    if (fuse_sim) {
      Ap.sim_state = (real*)h->f[RCG_FIELD_STATE];
      Ap.sim_state_prev = (real*)h->f[RCG_FIELD_STATE_PREV];
      Ap.sim_action = (const real*)h->f[RCG_FIELD_ACTION];
      Ap.sim_status = (uint32_t*)h->f[RCG_FIELD_STATUS];
      Ap.sim_n_sub = c.substeps_per_tick;
    }
    // (no instance - RQL / SQL with more than 36 dwords of weights: the tick is served by k_actor_dma / k_actor below)
    const bool launched =
        variant < DMA_RQL_0    ? launch_dma_packed<Sys, real, 3>(R, variant, grid, block, lds_req, h->stream, Ap, P, pp.a, pp.b)
        : variant >= DMA_SQL_0 ? launch_dma_packed<Sys, real, 4>(R, variant, grid, block, lds_req, h->stream, Ap, P, pp.a, pp.b)
                               : launch_dma_packed<Sys, real, 5>(R, variant, grid, block, lds_req, h->stream, Ap, P, pp.a, pp.b);
    if (launched) {
      note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR_DMA_PACKED, variant | (fuse_sim ? 16 : 0), (int)gpw);
      HIPCHK(h, hipGetLastError());
      return RCG_OK;
    }
    prof_give_back(h, pp);
    if (fuse_sim) {  // no packed instance after all: the env step as its own launch, then the other kernels
      int rc = op_sim_step<Sys>(h, c.substeps_per_tick);
      if (rc) return rc;
    }
  }
  if (dma_ok) {
    // Launch geometry, measured on MI355X at C2 (B = 65536, K = 256, N = 10):
    //  * residency: 2 blocks (8 waves) per CU stream faster than 8 blocks per CU - 0.204 ms against 0.213-0.218 ms.
    //    The dynamic-LDS request is raised to 56 KB so that at most two blocks fit into the CU's 160 KB;
    //  * envs per wave (gpw): each wave writes the results of its gpw envs once, coalesced, so gpw >= 4 turns 6
    //    scattered 4-byte writes per env into 16-64-byte segments; powers of two only (3, 6 measured 2-3 % slower);
    //  * rounds: the grid must be several times the 512 resident blocks so that the CUs stay balanced (single-round
    //    grids that do not divide evenly over 256 CUs lost 10 %: gpw = 20, 28, 48) - gpw is the largest power of
    //    two <= 16 that still leaves >= 8192 waves.
    const long Bn = h->sub_hi > 0 ? h->sub_hi - h->sub_lo : B;  // envs of this launch (a half of a split tick)
    long gpw = 1;
    while (gpw < 16 && Bn / (gpw * 2) >= 8192) gpw *= 2;
    if (knobs.gpw > 0) gpw = knobs.gpw;
    gpw = gpw < 1 ? 1 : (gpw > 64 ? 64 : gpw);
    ActorArgs<real> Ad = A;
    Ad.gpw = (int)gpw;
    Ad.env_lo = h->sub_lo;
    Ad.env_hi = h->sub_hi;
    const long pw = (Bn + gpw - 1) / gpw;
    const dim3 grid((unsigned)((pw + 3) / 4)), block(256);
    // blocks per CU: 2 for rows of >= 20 reals (a block keeps R KiB in flight in f32), 4 for shorter rows, which need more
    // waves to keep enough bytes on the wire (measured R = 6 ... 32 floats: 2 vs 4 differ by 1-3 % either side of
    // R = 20, R = 10 with 2 blocks/CU is 9 % slower than with 4; 8 blocks/CU is 5-15 % slower than the better of the two)
    // ... and 4 as well when a wave's whole slab is short (< 16 Ki reals: K = 64 at Nactor = 10 is 8 tiles per wave - the launch
    // is ramp-up and tail, more resident waves fill it better: 5.15 -> 5.57 TB/s)
    // (both thresholds count ELEMENTS - 16 Ki per wave, rows of 20 - since round 6: measured in f32 at first and kept in bytes, they
    // sent the float64 shapes K = 64 and Nactor = 5 to 2 blocks per CU, where 4 stream 5 % / 3 % faster: profiles/r06_sweep_f64_geometry.txt)
    const bool long_slab = (size_t)gpw * K * row_bytes >= (size_t)16 * 1024 * esz;
    // ... and 4 for the critic instances with many weights (>= 68 bytes of them: the robots' quad-lin / quadratic / quad-mix
    // structures in f32, 2tank quad-lin in f64), which are bound by VALU issue, not by the stream: more resident waves hide
    // more of it - 4-5 % on random weights, 8-11 % inside a closed loop (profiles/r04_per_cu_matrix.txt, r04_ab_per_cu.txt:
    // SQL quad-lin 307 -> 280 us, SQL quadratic 261 -> 232); MPC and the small structures lose 1-2 % with 4
    const bool valu_heavy = variant == DMA_MPC_GENF ||  // (35-77 fused multiply-adds per step of stage cost)
                            variant >= DMA_RQL_GEN_0 ||
                            ((dma_is_rql(variant) || dma_is_sql(variant)) && (size_t)dma_dc(dma_cs(variant), Sys::DS, DU) * esz >= 68);
    const int per_cu = knobs.per_cu > 0 ? knobs.per_cu : ((row_bytes >= 20 * esz && long_slab && !valu_heavy) ? 2 : 4);
    // J staging (operator mode): all envs of the wave when that fits under 64 KB next to the tiles, else env by env
    Ad.jwave = (A.J && 4 * tile + wslot + 4 * esz * gpw * K <= (size_t)64 * 1024) ? 1 : 0;
    size_t lds_req = 4 * tile + wslot + (A.J ? 4 * esz * K * (Ad.jwave ? gpw : 1) : 0);
    if (knobs.lds_pad > 0) {
      lds_req += (size_t)knobs.lds_pad;
    } else if (knobs.lds_pad == 0) {  // RCG_LDS_PAD=-1: no residency cap
      const size_t want = per_cu <= 2 ? (size_t)56 * 1024 : (per_cu <= 4 ? (size_t)36 * 1024 : 0);
      if (lds_req < want) lds_req = want;
    }
    // (blocks of 4 waves = one wave per SIMD: blocks of 2 or 1 waves at the same 8 resident waves per CU measured
    // 10-13 % slower)
    bool ok = false;
    const ProfPair pp = prof_take(h);  // a due ProfScope's pair travels in the dispatch
    if (variant >= DMA_RQL_GEN_0)
      ok = launch_dma<Sys, real, 7>(R, variant, grid, block, lds_req, h->stream, Ad, P, pp.a, pp.b);
    else if (variant >= DMA_MPC_GEND)
      ok = launch_dma<Sys, real, 6>(R, variant, grid, block, lds_req, h->stream, Ad, P, pp.a, pp.b);
    else if (variant < DMA_RQL_0)
      ok = launch_dma<Sys, real, 0>(R, variant, grid, block, lds_req, h->stream, Ad, P, pp.a, pp.b);
    else if (variant >= DMA_SQL_0)
      ok = launch_dma<Sys, real, 1>(R, variant, grid, block, lds_req, h->stream, Ad, P, pp.a, pp.b);
    else
      ok = launch_dma<Sys, real, 2>(R, variant, grid, block, lds_req, h->stream, Ad, P, pp.a, pp.b);
    if (!ok) prof_give_back(h, pp);
    if (ok) {  // (otherwise - unreachable for the rows dma_ok admits - k_actor below serves the tick: never refused half-way)
      note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR_DMA, variant, (int)gpw);
      HIPCHK(h, hipGetLastError());
      return RCG_OK;
    }
  }
  // Generated level grid in the regime every preset benchmark runs (float, MPC, the preset's diagonal R1 with its zero
  // weights, gamma == 1, no target, K = g * g a multiple of 256 with 64 % g == 0): the instance that holds the hand-packed
  // four-tile rollout and nothing else (rcg_kernels.hpp::GenPk)
  if constexpr (std::is_same<real, float>::value && GenPk<Sys>::supported) {
    const bool pk_ok = !cand && !generic && !tgt && c.gamma == 1.0 && Sys::ZW_PRESET != 0u &&
                       (P.zero_w & Sys::ZW_PRESET) == Sys::ZW_PRESET && K >= 256 && A.n_tiles % 4 == 0 && A.grid_g > 0 &&
                       (64 % A.grid_g) == 0 && !A.no_multi && !knobs.no_pk;
    if (pk_ok) {
      RCG_LAUNCH(h, (k_actor<Sys, real, false, false, false, true>), dim3(blocks), dim3(64 * wpb), lds, A, P);
      note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR, 8, A.G);  // variant bit 3: the packed instance
      HIPCHK(h, hipGetLastError());
      return RCG_OK;
    }
  }
  if (long_row) {
    if (tgt)
      RCG_LAUNCH(h, (k_actor<Sys, real, true, true, true, false, true>), dim3(blocks), dim3(64 * wpb), 0, A, P);
    else
      RCG_LAUNCH(h, (k_actor<Sys, real, true, false, true, false, true>), dim3(blocks), dim3(64 * wpb), 0, A, P);
    note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR, 1 | (tgt ? 2 : 0) | 4 | 16, A.G);  // variant bit 4: DIRECT rows
    HIPCHK(h, hipGetLastError());
    return RCG_OK;
  }
#define RCG_LAUNCH_ACTOR(GEN, TGT, STR) \
  RCG_LAUNCH(h, (k_actor<Sys, real, GEN, TGT, STR>), dim3(blocks), dim3(64 * wpb), lds, A, P)
#define RCG_LAUNCH_ACTOR2(GEN, TGT)      \
  do {                                   \
    if (cand)                            \
      RCG_LAUNCH_ACTOR(GEN, TGT, true);  \
    else                                 \
      RCG_LAUNCH_ACTOR(GEN, TGT, false); \
  } while (0)
  if (generic) {
    if (tgt)
      RCG_LAUNCH_ACTOR2(true, true);
    else
      RCG_LAUNCH_ACTOR2(true, false);
  } else {
    if (tgt)
      RCG_LAUNCH_ACTOR2(false, true);
    else
      RCG_LAUNCH_ACTOR2(false, false);
  }
#undef RCG_LAUNCH_ACTOR2
#undef RCG_LAUNCH_ACTOR
  note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR, (generic ? 1 : 0) | (tgt ? 2 : 0) | (cand ? 4 : 0), A.G);
  HIPCHK(h, hipGetLastError());
  return RCG_OK;
}

template <typename Sys>
int op_actor(rcg_handle* h, const char* who, const void* cand, int K, const void* obs, const void* state_sys,
                    const void* w, void* J, void* action, void* best_J, int32_t* best_idx, bool tick, bool sim_first) {
  return by_dtype(h, [&](auto r) {
    return launch_actor<Sys, decltype(r)>(h, who, cand, K, obs, state_sys, w, J, action, best_J, best_idx, tick,
                                          sim_first);
  });
}

template <typename Sys>
int op_optimize(rcg_handle* h, int32_t iters, const void* obs, const void* state_sys, const void* u_init,
                       int shift, void* u_opt, void* action, void* best_J, int32_t* n_iter, bool tick, bool sim_first) {
  const rcg_cfg& c = h->cfg;
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    const KParams<real>& P = params<real>(h);
    OptArgs<real> A;
    memset(&A, 0, sizeof A);
    A.obs = obs ? (const real*)obs : (const real*)h->f[RCG_FIELD_STATE];
    if (state_sys)
      A.state_sys = (const real*)state_sys;
    else if (obs)
      A.state_sys = (const real*)obs;
    else
      A.state_sys = (const real*)h->f[(tick && (c.flags & RCG_FLAG_REF_LAG)) ? RCG_FIELD_STATE_PREV : RCG_FIELD_STATE];
    A.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    A.w = (const real*)h->f[RCG_FIELD_W_CRITIC];
    if (c.mode != RCG_MODE_MPC && !A.w)
      return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_actor_optimize: RQL/SQL need critic weights (buffer_size > 0)");
    A.u_init = (const real*)u_init;
    A.u_opt = (real*)u_opt;
    A.action_out = (real*)action;
    A.best_J = (real*)best_J;
    A.n_iter = n_iter;
    A.accum = (tick && !(c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP)) ? (real*)h->f[RCG_FIELD_ACCUM] : nullptr;
    A.step_idx = tick ? (int32_t*)h->f[RCG_FIELD_STEP_IDX] : nullptr;
    for (int i = 0; i < Sys::DU; ++i) A.u0[i] = (real)c.action_init[i];
    A.iters = iters;
    A.shift = shift;
    A.memory = opt_memory_of(h);
    A.ftol = (real)h->opt_ftol;
    const bool loop = h->loop_io.on;  // rcg_loop_step's one-launch sample: head and tail of the loop iteration in this launch
    if (loop)
      fill_loop_args<real>(h, A.loop, h->loop_io.act_in, h->loop_io.n_substeps, 1, 1, 1, h->loop_io.dc, h->loop_io.out,
                           h->loop_io.flag, h->loop_io.seq);
    const bool generic = !(c.mode == RCG_MODE_MPC && P.stage_kind == 0);
    A.dcw = c.mode != RCG_MODE_MPC ? h->dc : 0;
    // waves per block: the waves of a block do not cooperate, so the block size only decides how many waves of LDS fit a CU's
    // 160 KB: 4 (one per SIMD) unless 2 or 1 bring more waves onto the CU (quad-mix on the 3-wheel robot with 4 pairs: 20.4 KB
    // per wave = ONE block of four, but seven blocks of one; long horizons in f64: N = 20, 4 pairs needs 70 KB per wave)
    const size_t lds_wave = opt_wave_lds_bytes(h);
    int wpb = 4;
    size_t on_cu = 0;
    for (int cand_wpb = 4; cand_wpb >= 1; cand_wpb >>= 1) {
      const size_t fit = lds_wave * cand_wpb ? ((size_t)160 * 1024 / (lds_wave * cand_wpb)) * cand_wpb : 0;
      if (fit * 4 > on_cu * 5) {  // a smaller block must bring a quarter more waves: 9 single-wave blocks against 2 x 4 measured 10 % SLOWER
                                  // (MPC with 4 pairs, 18.1 KB per wave: one SIMD carries three waves and the second round is ragged)
        on_cu = fit;
        wpb = cand_wpb;
      }
    }
    const size_t lds = lds_wave * wpb;
    if (lds > (size_t)160 * 1024)
      return rcg_fail(h, RCG_ERR_UNSUPPORTED,
                      "rcg_actor_optimize: horizon %d with %d curvature pairs needs %zu B of LDS per wave (rcg_set_optimizer)",
                      c.n_actor, opt_memory_of(h), lds_wave);
    const dim3 grid(blocks_for(c.batch, wpb * OPT_G)), block(64 * wpb);  // a wave owns OPT_G envs
    const bool tgt = c.flags & RCG_FLAG_HAS_TARGET;
    if (tick && sim_first) {  // rcg_control_tick_opt (MPC): the env step of the tick, once every argument check has passed
      const int rc = op_sim_step<Sys>(h, c.substeps_per_tick);
      if (rc) return rc;
    }
    const bool pairs = A.memory > 0;  // the instance with the curvature pairs and the four-lanes-per-env phase 1b
    ProfScope prof_scope(h, RCG_KERNEL_ACTOR);
#define RCG_OPT_LAUNCH(T, GEN, PR)                                                                                  \
  do {                                                                                                              \
    auto fn = k_actor_opt<Sys, real, T, GEN, PR>;                                                                   \
    if (lds > 64 * 1024) /* beyond the default dynamic-LDS limit (the CU has 160 KB) */                             \
      HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                                     \
    RCG_LAUNCH(h, fn, grid, block, lds, A, P);                                                                      \
  } while (0)
    const int sel = (generic ? 4 : 0) | (tgt ? 2 : 0) | (pairs ? 1 : 0);
    if (loop) {  // (rcg_loop_step asks only where opt_plain_instance() holds)
      if (generic || pairs || obs || state_sys != h->f[RCG_FIELD_STATE_PREV] || u_init || tick)
        return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_loop_step: the one-launch sample is the plain MPC instance on the handle's own fields");
      auto fn = tgt ? k_actor_opt<Sys, real, true, false, false, true> : k_actor_opt<Sys, real, false, false, false, true>;
      if (lds > 64 * 1024)
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      RCG_LAUNCH(h, fn, grid, block, lds, A, P);
    } else
    switch (sel) {
      case 0: RCG_OPT_LAUNCH(false, false, false); break;
      case 1: RCG_OPT_LAUNCH(false, false, true); break;
      case 2: RCG_OPT_LAUNCH(true, false, false); break;
      case 3: RCG_OPT_LAUNCH(true, false, true); break;
      case 4: RCG_OPT_LAUNCH(false, true, false); break;
      case 5: RCG_OPT_LAUNCH(false, true, true); break;
      case 6: RCG_OPT_LAUNCH(true, true, false); break;
      default: RCG_OPT_LAUNCH(true, true, true); break;
    }
#undef RCG_OPT_LAUNCH
    note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR_OPT, (generic ? 1 : 0) | (tgt ? 2 : 0) | (pairs ? 4 : 0) | (loop ? 8 : 0), OPT_G);
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// rcg_actor_search / rcg_control_tick_search: `rounds` rounds of K generated candidates per env, evaluated where they are
// generated (k_actor_search, rcg_search.hpp).  The caller (rcg_api.hip) has checked K and the critic weights.
template <typename Sys>
int op_search(rcg_handle* h, int32_t K, int32_t rounds, int32_t round0, const void* obs, const void* state_sys,
                     const void* centre, int shift, void* u_best, void* action, void* best_J, int32_t* best_idx, bool tick,
                     bool sim_first) {
  const rcg_cfg& c = h->cfg;
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    const KParams<real>& P = params<real>(h);
    SearchArgs<real> A;
    memset(&A, 0, sizeof A);
    A.obs = obs ? (const real*)obs : (const real*)h->f[RCG_FIELD_STATE];
    if (state_sys)
      A.state_sys = (const real*)state_sys;
    else if (obs)
      A.state_sys = (const real*)obs;
    else
      A.state_sys = (const real*)h->f[(tick && (c.flags & RCG_FLAG_REF_LAG)) ? RCG_FIELD_STATE_PREV : RCG_FIELD_STATE];
    A.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    A.w = (const real*)h->f[RCG_FIELD_W_CRITIC];
    A.centre_in = (const real*)centre;
    A.u_best = (real*)u_best;
    A.action_out = (real*)action;
    A.best_J = (real*)best_J;
    A.best_idx = best_idx;
    A.accum = (tick && !(c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP)) ? (real*)h->f[RCG_FIELD_ACCUM] : nullptr;
    A.step_rw = tick ? (int32_t*)h->f[RCG_FIELD_STEP_IDX] : nullptr;
    A.episode_idx = (const int32_t*)h->f[RCG_FIELD_EPISODE_IDX];
    A.step_idx = (const int32_t*)h->f[RCG_FIELD_STEP_IDX];
    for (int i = 0; i < Sys::DU; ++i) A.u0[i] = (real)c.action_init[i];
    A.K = K;
    A.rounds = rounds;
    A.round0 = round0;
    A.shift = shift;
    A.seed = c.seed;
    A.env_id_base = c.env_id_base;
    const int R = c.n_actor * Sys::DU;
    const bool generic = !(c.mode == RCG_MODE_MPC && P.stage_kind == 0);
    const bool tgt = (c.flags & RCG_FLAG_HAS_TARGET) != 0;
    // register rows (compile-time horizon): MPC with a diagonal stage cost, the preset's target setting, Nactor 3 / 5 / 10
// longest horizon whose FLOAT64 search keeps its rows in registers.  Round 6, interleaved A/B on one device
// (profiles/r06_ab_search_f64_rows.txt): at Nactor = 10 the register-row instance needs 256 VGPRs + 19 AGPRs (one wave per SIMD) and
// takes 690 us per C2-shape round of 256, the LDS-row instance (124 VGPRs, four waves) 525 us; at Nactor = 5 registers win, 257 against 268.
#ifndef RCG_SEARCH_F64_REG_ROWS
#define RCG_SEARCH_F64_REG_ROWS 5
#endif
    int nc = (!generic && tgt == Sys::TGT && (c.n_actor == 3 || c.n_actor == 5 || c.n_actor == 10)) ? c.n_actor : 0;
    if (sizeof(real) == 8 && nc > RCG_SEARCH_F64_REG_ROWS) nc = 0;  // (float64 rows of 20 reals: 256 VGPRs + AGPRs, one wave per SIMD)
    int wpb = 4;  // waves (= envs) per workgroup
    const size_t lds_wave = (size_t)search_lds_reals(R, nc > 0) * sizeof(real);
    while (wpb > 1 && lds_wave * wpb > (size_t)64 * 1024) wpb >>= 1;
    const size_t lds = lds_wave * wpb;
    if (lds > (size_t)64 * 1024)  // (65 rows per wave: 126 doubles / 252 floats per row)
      return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_actor_search: rows of %d reals need %zu B of LDS per wave (at most 65536)", R,
                      lds_wave);
    const dim3 grid(blocks_for(c.batch, wpb)), block(64 * wpb);
    if (tick && sim_first) {
      const int rc = op_sim_step<Sys>(h, c.substeps_per_tick);
      if (rc) return rc;
    }
    ProfScope prof_scope(h, RCG_KERNEL_ACTOR);
    if (nc == 3)
      RCG_LAUNCH(h, (k_actor_search<Sys, real, false, Sys::TGT, 3>), grid, block, lds, A, P);
    else if (nc == 5)
      RCG_LAUNCH(h, (k_actor_search<Sys, real, false, Sys::TGT, 5>), grid, block, lds, A, P);
    else if (nc == 10)
      RCG_LAUNCH(h, (k_actor_search<Sys, real, false, Sys::TGT, 10>), grid, block, lds, A, P);
    else if (generic && tgt)
      RCG_LAUNCH(h, (k_actor_search<Sys, real, true, true, 0>), grid, block, lds, A, P);
    else if (generic)
      RCG_LAUNCH(h, (k_actor_search<Sys, real, true, false, 0>), grid, block, lds, A, P);
    else if (tgt)
      RCG_LAUNCH(h, (k_actor_search<Sys, real, false, true, 0>), grid, block, lds, A, P);
    else
      RCG_LAUNCH(h, (k_actor_search<Sys, real, false, false, 0>), grid, block, lds, A, P);
    note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_ACTOR_SEARCH, (generic ? 1 : 0) | (tgt ? 2 : 0) | (nc > 0 ? 4 : 0), 1);
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// rcg_control_ticks / rcg_control_tick_n: T MPC ticks in one launch (k_ticks, k_ticks_pk; rcg_ticks.hpp), generated grid
// (cand == nullptr) or the caller's candidate tensor, with or without the disturbance model.  The caller has checked
// mode / K.
template <typename Sys>
int op_ticks(rcg_handle* h, int32_t T, int32_t K, const void* cand) {
  constexpr int DU = Sys::DU;
  const rcg_cfg& c = h->cfg;
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    const KParams<real>& P = params<real>(h);
    TicksArgs<real> A;
    memset(&A, 0, sizeof A);
    A.state = (real*)h->f[RCG_FIELD_STATE];
    A.state_prev = (real*)h->f[RCG_FIELD_STATE_PREV];
    A.action = (real*)h->f[RCG_FIELD_ACTION];
    A.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    A.accum = (real*)h->f[RCG_FIELD_ACCUM];
    A.step_idx = (int32_t*)h->f[RCG_FIELD_STEP_IDX];
    A.status = (uint32_t*)h->f[RCG_FIELD_STATUS];
    A.best_J = (real*)h->f[RCG_FIELD_BEST_J];
    A.best_idx = (int32_t*)h->f[RCG_FIELD_BEST_IDX];
    A.cand = (const real*)cand;
    A.dist = (c.flags & RCG_FLAG_DISTURB) ? 1 : 0;
    if (A.dist) {
      A.disturb = (real*)h->f[RCG_FIELD_DISTURB];
      A.substep_idx = (int32_t*)h->f[RCG_FIELD_SUBSTEP_IDX];
      A.episode_idx = (const int32_t*)h->f[RCG_FIELD_EPISODE_IDX];
      A.D = disturb_pars(h);
    }
    A.T = T;
    A.n_sub = c.substeps_per_tick;
    A.K = K;
    if (K >= 64) {  // the tiling of launch_actor
      A.Kp = 64;
      A.G = 1;
      A.n_tiles = (K + 63) / 64;
    } else {
      int kp = 1;
      while (kp < K) kp <<= 1;
      A.Kp = kp;
      A.G = 64 / kp;
      A.n_tiles = 1;
    }
    A.grid_g = cand ? 0 : (DU == 1 ? K : (int)std::floor(std::sqrt((double)K) + 1e-9));
    A.no_multi = dev_knobs().no_gen_multi ? 1 : 0;
    const long n_waves = (c.batch + A.G - 1) / A.G;
    const bool generic = P.stage_kind != 0;
    const bool tgt = (c.flags & RCG_FLAG_HAS_TARGET) != 0;
    // streamed candidates: the wave's rows stay in LDS for all T ticks when they fit (32 KB per wave: four waves per block,
    // one block per CU at worst), else they are re-staged tile by tile every tick (served by L2 / Infinity Cache at the
    // batch sizes this entry point is for)
    const int R = c.n_actor * DU;
    const size_t row_bytes = (size_t)R * sizeof(real);
    size_t lds = 0;
    if (cand) {
      const size_t rows_wave = K >= 64 ? (size_t)K : (size_t)A.G * K;
      A.vec_ok = (row_bytes % 16 == 0 && ((uintptr_t)cand % 16) == 0) ? 1 : 0;
      A.stage_once = rows_wave * row_bytes <= (size_t)32 * 1024 ? 1 : 0;
      const size_t per_wave = (A.stage_once ? rows_wave : (size_t)64) * row_bytes;
      A.lds_reals = (int)((per_wave + 15) / 16 * 16 / sizeof(real));
      lds = (size_t)A.lds_reals * sizeof(real) * 4;
    }
    const dim3 grid((unsigned)((n_waves + 3) / 4)), block(256);
    ProfScope prof_scope(h, RCG_KERNEL_ACTOR);
    if constexpr (std::is_same<real, float>::value && GenPk<Sys>::supported) {
      const bool pk_ok = !cand && !A.dist && !generic && !tgt && c.gamma == 1.0 && Sys::ZW_PRESET != 0u &&
                         (P.zero_w & Sys::ZW_PRESET) == Sys::ZW_PRESET && K >= 256 && A.n_tiles % 4 == 0 &&
                         (64 % A.grid_g) == 0 && !A.no_multi && !dev_knobs().no_pk;
      if (pk_ok) {  // the kernel around the hand-packed rollout (k_ticks_pk): several envs per wave
        // envs per wave: a power of two <= 8 that leaves >= 2048 waves (two per SIMD), so that a wave's loads, the env step
        // (one RK4 for all of its envs) and its stores are amortised without unbalancing the launch (65 536 envs, one tick, us
        // per launch by envs per wave: 1: 93.5, 2: 75.6, 4: 67.0, 8: 63.8, 16: 65.1 - profiles/r04_ab_ticks_pk.txt).  Alone, a
        // launch of 8192 .. 32768 envs lasts the same within 2-4 % for 2, 4 or 8 envs per wave; next to OTHER handles' kernels
        // (configs[4]'s pool: three handles of 21 846 envs on three streams) fewer, longer waves leave the co-running kernels
        // room: 93 us per pool tick at 2 envs per wave (the former rule: >= 8192 waves), 76 at 4, 67 at 8
        // (profiles/r04_ab_pool_gpw.txt)
        int gpw = 1;
        while (gpw < 8 && c.batch / (gpw * 2) >= 2048) gpw *= 2;
        if (dev_knobs().gpw > 0 && dev_knobs().gpw <= 64) gpw = (int)dev_knobs().gpw;
        A.gpw = gpw;
        const long pw = ((long)c.batch + gpw - 1) / gpw;
        RCG_LAUNCH(h, (k_ticks_pk<Sys>), dim3((unsigned)((pw + 3) / 4)), block, 0, A, P);
        note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_TICKS, 8, gpw);
        HIPCHK(h, hipGetLastError());
        return (int)RCG_OK;
      }
    }
#define RCG_TICKS(GEN, TGT)                                                             \
  do {                                                                                  \
    if (cand)                                                                           \
      RCG_LAUNCH(h, (k_ticks<Sys, real, GEN, TGT, true>), grid, block, lds, A, P);     \
    else                                                                                \
      RCG_LAUNCH(h, (k_ticks<Sys, real, GEN, TGT, false>), grid, block, 0, A, P);      \
  } while (0)
    if (lds > 64 * 1024) {
      const void* fn = generic ? (tgt ? (const void*)&k_ticks<Sys, real, true, true, true> : (const void*)&k_ticks<Sys, real, true, false, true>)
                               : (tgt ? (const void*)&k_ticks<Sys, real, false, true, true> : (const void*)&k_ticks<Sys, real, false, false, true>);
      HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (generic && tgt)
      RCG_TICKS(true, true);
    else if (generic)
      RCG_TICKS(true, false);
    else if (tgt)
      RCG_TICKS(false, true);
    else
      RCG_TICKS(false, false);
#undef RCG_TICKS
    note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_TICKS, (generic ? 1 : 0) | (tgt ? 2 : 0) | (cand ? 4 : 0), A.G);
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// rcg_control_ticks on an RQL / SQL handle: T ticks in one launch (k_ticks_mem), generated candidates.  The caller has
// checked K, the critic buffers, 1 <= Ncritic - 1 <= kFitMaxRows, no disturbance model, and that an instance exists
// (ticks_mem_ok).
template <typename Sys>
static bool ticks_mem_ok(const rcg_handle* h) {
  const bool tgt = (h->cfg.flags & RCG_FLAG_HAS_TARGET) != 0;
  return tgt == Sys::TGT || (!tgt && Sys::TGT);  // instances exist for the preset's target setting (zeros serve "no target")
}

template <typename Sys>
int op_ticks_mem(rcg_handle* h, int32_t T, int32_t K, const void* cand) {
  constexpr int DU = Sys::DU;
  const rcg_cfg& c = h->cfg;
  if (!ticks_mem_ok<Sys>(h)) return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_control_ticks: no persistent RQL/SQL instance for this observation target");
  // (structures with >= 20 weights fit with four lanes per env: the wave's envs must fit its 16 quads)
  if (dma_dc(c.critic_struct, Sys::DS, Sys::DU) >= kFitLanesMinDc && c.n_critic - 1 <= 3 && K < 4)
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_control_ticks: K >= 4 with this critic structure (four lanes per env in the fit)");
  return by_dtype(h, [&](auto r) {
    using real = decltype(r);
    const KParams<real>& P = params<real>(h);
    TicksMemArgs<real> M;
    memset(&M, 0, sizeof M);
    ActorArgs<real>& A = M.A;
    A.cand = (const real*)cand;  // nullptr: the generated grid
    A.obs = (const real*)h->f[RCG_FIELD_STATE];
    A.state_sys = (const real*)h->f[(c.flags & RCG_FLAG_REF_LAG) ? RCG_FIELD_STATE_PREV : RCG_FIELD_STATE];
    A.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    A.w = (const real*)h->f[RCG_FIELD_W_CRITIC];
    A.action_out = (real*)h->f[RCG_FIELD_ACTION];
    A.best_J = (real*)h->f[RCG_FIELD_BEST_J];
    A.best_idx = (int32_t*)h->f[RCG_FIELD_BEST_IDX];
    A.accum = !(c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP) ? (real*)h->f[RCG_FIELD_ACCUM] : nullptr;
    A.step_idx = (int32_t*)h->f[RCG_FIELD_STEP_IDX];
    A.K = K;
    if (K >= 64) {
      A.Kp = 64;
      A.G = 1;
      A.n_tiles = (K + 63) / 64;
    } else {
      int kp = 1;
      while (kp < K) kp <<= 1;
      A.Kp = kp;
      A.G = 64 / kp;
      A.n_tiles = 1;
    }
    A.grid_g = cand ? 0 : (DU == 1 ? K : (int)std::floor(std::sqrt((double)K) + 1e-9));
    A.no_multi = dev_knobs().no_gen_multi ? 1 : 0;
    const size_t row_bytes = (size_t)c.n_actor * DU * sizeof(real);
    A.vec_ok = (cand && row_bytes % 16 == 0 && ((uintptr_t)cand % 16) == 0) ? 1 : 0;
    const size_t lds = cand ? (size_t)4 * 64 * row_bytes : 0;  // four waves, a 64-row tile each
    if (lds > (size_t)64 * 1024)
      return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_control_tick_n: rows of %zu bytes do not fit the persistent kernel's tiles", row_bytes);
    FitArgs<real>& F = M.F;
    F.w_critic = (real*)h->f[RCG_FIELD_W_CRITIC];
    F.w_prev = (real*)h->f[RCG_FIELD_W_PREV];
    F.obs_buf = (real*)h->f[RCG_FIELD_OBS_BUF];
    F.act_buf = (real*)h->f[RCG_FIELD_ACT_BUF];
    F.wcfg = reinterpret_cast<const double*>((unsigned char*)h->d_const + kConstW);
    F.do_sim = 1;
    F.do_push = 1;
    F.state = (const real*)h->f[RCG_FIELD_STATE];
    F.action = (const real*)h->f[RCG_FIELD_ACTION];
    F.sim.state = (real*)h->f[RCG_FIELD_STATE];
    F.sim.state_prev = (real*)h->f[RCG_FIELD_STATE_PREV];
    F.sim.action = (const real*)h->f[RCG_FIELD_ACTION];
    F.sim.pars_env = (const real*)h->f[RCG_FIELD_PARS];
    F.sim.accum = (real*)h->f[RCG_FIELD_ACCUM];
    F.sim.status = (uint32_t*)h->f[RCG_FIELD_STATUS];
    F.sim.n_sub = c.substeps_per_tick;
    M.T = T;
    M.tick0 = (int)h->tick_count;
    M.every = c.critic_every_ticks > 1 ? c.critic_every_ticks : 1;
    const int m = c.n_critic - 1;
    const long n_waves = (c.batch + A.G - 1) / A.G;
    const dim3 grid((unsigned)((n_waves + 3) / 4)), block(256);
    ProfScope prof_scope(h, RCG_KERNEL_ACTOR);
#define RCG_TM(CS)                                                                                              \
  do {                                                                                                          \
    constexpr bool ml_ = CriticDim<CS, Sys::DS, Sys::DU>::value >= kFitLanesMinDc;                              \
    if (m <= 3 && cand)                                                                                         \
      RCG_LAUNCH(h, (k_ticks_mem<Sys, real, CS, 3, Sys::TGT, ml_, true>), grid, block, lds, M, h->p64, P);      \
    else if (m <= 3)                                                                                            \
      RCG_LAUNCH(h, (k_ticks_mem<Sys, real, CS, 3, Sys::TGT, ml_, false>), grid, block, 0, M, h->p64, P);       \
    else if (cand)                                                                                              \
      RCG_LAUNCH(h, (k_ticks_mem<Sys, real, CS, kFitMaxRows, Sys::TGT, false, true>), grid, block, lds, M, h->p64, P); \
    else                                                                                                        \
      RCG_LAUNCH(h, (k_ticks_mem<Sys, real, CS, kFitMaxRows, Sys::TGT>), grid, block, 0, M, h->p64, P);        \
  } while (0)
    switch (c.critic_struct) {
      case RCG_CRITIC_QUAD_LIN: RCG_TM(RCG_CRITIC_QUAD_LIN); break;
      case RCG_CRITIC_QUADRATIC: RCG_TM(RCG_CRITIC_QUADRATIC); break;
      case RCG_CRITIC_QUAD_NOMIX: RCG_TM(RCG_CRITIC_QUAD_NOMIX); break;
      default: RCG_TM(RCG_CRITIC_QUAD_MIX); break;
    }
#undef RCG_TM
    note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_TICKS, 16 | 1 | (Sys::TGT ? 2 : 0) | (cand ? 4 : 0), A.G);  // variant bit 4: k_ticks_mem
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// CtrlNominal3WRobot / CtrlNominal3WRobotNI for n points (tick: the handle's envs, with the tick epilogue)
template <typename Sys>
static int op_nominal(rcg_handle* h, const void* obs, void* action, void* lyap, void* theta, int32_t n, double gain,
                      const double* ctrl_pars, int32_t clip, bool tick) {
  if constexpr (!Nominal<Sys>::supported) {
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "nominal controller: the reference defines none for this system");
  } else {
    const rcg_cfg& c = h->cfg;
    return by_dtype(h, [&](auto r) {
      using real = decltype(r);
      NomArgs<real> A;
      A.obs = (const real*)obs;
      A.action = (real*)action;
      A.lyap = (real*)lyap;
      A.theta = (real*)theta;
      A.accum = (tick && !(c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP)) ? (real*)h->f[RCG_FIELD_ACCUM] : nullptr;
      A.step_idx = tick ? (int32_t*)h->f[RCG_FIELD_STEP_IDX] : nullptr;
      A.n = n;
      A.gain = gain;
      A.m = ctrl_pars ? ctrl_pars[0] : c.pars[0];
      A.I = ctrl_pars ? ctrl_pars[1] : c.pars[1];
      A.clip = clip;
      ProfScope prof_scope(h, RCG_KERNEL_ACTOR);
      RCG_LAUNCH(h, (k_nominal<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, A, params<real>(h));
      note_launch(h, RCG_KERNEL_ACTOR, RCG_KID_NOMINAL, 0, 64);
      HIPCHK(h, hipGetLastError());
      return (int)RCG_OK;
    });
  }
}

// The launchers above are instantiated per environment by rcg_sys_inst.hip, which is compiled once per (system, part): the
// heavy ones (op_actor, op_ticks, op_ticks_mem, op_optimize + op_search, op_sim_step + op_critic_update) each in a part of
// their own - `extern template` everywhere else - so that no object takes more than about a minute to build; part 0 holds
// the table below and the light launchers.  Through their launch expressions the launchers pull in every kernel.
template <typename Sys>
struct SysInstances {
  static SysVTable table() {
    return SysVTable{&op_rhs<Sys>,   &op_stage_obj<Sys>, &op_critic<Sys>,        &op_critic_cost<Sys>, &op_actor<Sys>,
                     &op_sim_step<Sys>, &op_critic_update<Sys>, &op_optimize<Sys>, &op_nominal<Sys>,
                     &op_ticks<Sys>,  &op_rhs_full<Sys>, &op_search<Sys>, &op_ticks_mem<Sys>, &op_loop<Sys>};
  }
};

}  // namespace rcg
