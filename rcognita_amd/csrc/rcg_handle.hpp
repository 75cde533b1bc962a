// rcg_handle.hpp - host-side state shared by the translation units of librcg.so.
//
// The library is split so that hipcc can build it in parallel: rcg_api.hip holds the C ABI and the
// system-independent kernels; rcg_sys_<system>.hip each instantiate every system-templated kernel and
// launcher for one environment (rcg_sysops.hpp) and export them through a SysVTable.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rcg_actor_opt.hpp"
#include "rcg_kernels.hpp"

struct rcg_handle {
  rcg_cfg cfg;
  int ds, du, np, dc, nchi;
  size_t esz;  // sizeof(real)
  hipStream_t stream;
  hipStream_t own_stream;  // created by rcg_use_own_stream, destroyed with the handle (nullptr: none)
  void* f[RCG_FIELD_COUNT_];
  size_t fbytes[RCG_FIELD_COUNT_];
  double* d_summary;
  long tick_count;  // control ticks issued through rcg_control_tick (drives the critic period)
  int opt_memory;   // curvature pairs of k_actor_opt (rcg_set_optimizer); -1: the default of opt_memory_of()
  double opt_ftol;  // k_actor_opt: an env stops after an accepted step that gained <= opt_ftol (rcg_set_optimizer_tol; 0: never)
  void* d_const;    // constant block in HBM, layout kConst* below
  // rcg_loop_step -> op_optimize: when `on`, the decision's launch also does the loop iteration's head and tail (k_actor_opt LOOP)
  struct {
    bool on;
    const double* act_in;
    int n_substeps, dc;
    double *out, *flag, seq;
  } loop_io;
  bool loop_pending;         // rcg_loop_step_begin has enqueued a step that rcg_loop_step_end has not collected
  int loop_pending_row;      //   doubles per env of its rows
  bool loop_pending_decided; //   it carries a decision (its sequence sits in sqn_alt until the step is collected)
  void* sqn_alt;             // spare ACTION_SQN buffer: rcg_loop_step_end swaps it in when a deciding step is collected
  void* loop_pin;            // rcg_loop_step's own pinned buffer (kBounceBytes, first use)
  uint64_t loop_seq;         // rcg_loop_step: sequence number of the last call (the glue kernel hands it back through pinned memory)
  void* fit_scratch;         // k_critic_fit_gen (Ncritic - 1 > kFitMaxRows): per-env stack / factor, allocated on first use
  size_t fit_scratch_bytes;
  rcg::KParams<float> p32;
  rcg::KParams<double> p64;
  const struct SysVTable* sys;
  std::string err;
  // measurement (rcg_profile): event pairs recorded on `stream`, drained into totals on demand
  unsigned prof_mask;  // bit k: bracket launches of rcg_kernel k
  unsigned prof_stride;                    // ... every prof_stride-th launch (>= 1)
  uint64_t prof_seen[RCG_KERNEL_COUNT_];   // launches seen since rcg_profile()
  std::vector<hipEvent_t> ev_free;
  struct Pending {
    hipEvent_t a, b;
    int kernel;
  };
  std::vector<Pending> ev_pending;
  double prof_ms[RCG_KERNEL_COUNT_];
  int64_t prof_n[RCG_KERNEL_COUNT_];
  std::vector<float> prof_samples[RCG_KERNEL_COUNT_];  // per-launch durations (ms), launch order, first kProfMaxSamples
  hipEvent_t cur_a, cur_b;  // event pair of the ProfScope whose sample is due: the next launch carries it (prof_take)
  bool scope_due;           // a due ProfScope is alive (dev build: a second launch inside it aborts)
  hipEvent_t order_ev;      // rcg_wait_stream's event (created on first use)
  hipEvent_t release_ev;    // rcg_release_stream's event (created on first use)
  // A tick in two halves (round 5; rcg_control_tick of an RQL / SQL handle with a caller's tensor on k_actor_dma): the envs
  // [0, half) and [half, B) run their [env step + push + fit] -> decision on two internal streams, so that - exactly as with
  // two handles on two streams, MixedPool(parts=2) - the latency-bound fit of one half runs under the streaming kernel of the
  // other, tick after tick (the halves never meet: envs are independent).  The handle's own stream rejoins them the next
  // time anything else is asked of the handle (join_split, called by DeviceGuard).
  void* bounce;                // pinned host buffer (kBounceBytes, allocated on first use): small device-to-host reads land here
  int tick_parts;              // rcg_set_tick_parts: 0 auto (2 from kSplitMinBatch envs), 1 never, 2 whenever the tick is eligible
  hipStream_t split_stream[2];  // created on first use
  hipEvent_t split_fork, split_join[2];
  bool split_pending;          // work of a split tick is queued on split_stream[] and not yet joined
  int sub_lo, sub_hi;          // the launchers' current sub-range of the batch (sub_hi == 0: all envs)
  int probe;                   // launch_actor: 1 = only report whether this (cand, K) goes to k_actor_dma -> probe = 2 / 3 (no / yes)
  // rcg_last_launch: which kernel served the last launch of each kind
  struct LastLaunch {
    int32_t kernel_id, variant, envs_per_wave;
  } last[RCG_KERNEL_COUNT_];
};
static constexpr size_t kProfMaxSamples = 65536;

static constexpr size_t kBounceBytes = 16384;  // host reads up to this size go through the handle's pinned buffer
static constexpr int kSplitMinBatch = 65536;   // envs from which an eligible tick is split by default
static constexpr int kVariantSplitBit = 4096;  // rcg_last_launch: the launch served one half of a split tick
static inline void note_launch(rcg_handle* h, int kind, int kernel_id, int variant, int envs_per_wave) {
  h->last[kind].kernel_id = kernel_id;
  h->last[kind].variant = variant | (h->sub_hi > 0 ? kVariantSplitBit : 0);
  h->last[kind].envs_per_wave = envs_per_wave;
}

// [0,392) R1|R2 as f32, [512,1296) R1|R2 as f64, [1296,2256) w_init|w_min|w_max as f64
static constexpr size_t kConstR64 = 512, kConstW = 1296, kConstBytes = 2256;
static constexpr int kFitMaxRows = 8;  // Ncritic - 1 <= 8: the register kernels (k_critic_fit); beyond: k_critic_fit_gen (HBM scratch)

// sets the handle's (or, for h == nullptr, the thread's) error text and returns `code`
int rcg_fail(rcg_handle* h, int code, const char* fmt, ...);

#define HIPCHK(h, call)                                                                                 \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return rcg_fail((h), RCG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                      __LINE__);                                                                        \
  } while (0)

// Measurement (rcg_profile).  A sampled launch carries an event pair IN its dispatch (hipExtLaunchKernelGGL): the
// events read the dispatch packet's own start / end stamps, the figures a rocprofv3 kernel trace shows.  Rounds 1-2
// recorded marker events before and after the launch instead; the bracket then contained the drain of the stream in
// front of the kernel (tools/event_probe.hip on MI355X, 1.35 GB streaming kernel: markers 192.4 us, dispatch stamps
// 190.6 us).  Either way a timed launch costs the stream ~7 us (the completion signal is written and waited for before
// the next dispatch), so callers sample with a stride.
// RAII: while a ProfScope whose sample is due is alive, the handle's next RCG_LAUNCH takes its events.
struct ProfScope {
  rcg_handle* h;
  hipEvent_t a, b;
  int kernel;
  bool on;
  ProfScope(rcg_handle* h_, int kernel_)
      : h(h_), a(nullptr), b(nullptr), kernel(kernel_), on((h_->prof_mask >> kernel_) & 1u) {
    if (!on) return;
    if ((h->prof_seen[kernel]++ % h->prof_stride) != 0) {
      on = false;
      return;
    }
    for (hipEvent_t* e : {&a, &b}) {
      if (!h->ev_free.empty()) {
        *e = h->ev_free.back();
        h->ev_free.pop_back();
      } else if (hipEventCreate(e) != hipSuccess) {
        if (a) h->ev_free.push_back(a);
        a = b = nullptr;
        on = false;
        return;
      }
    }
    h->cur_a = a;
    h->cur_b = b;
    h->scope_due = true;
  }
  ~ProfScope() {
    if (!on) return;
    h->scope_due = false;
    if (h->cur_a == nullptr) {  // a launch took the pair
      h->ev_pending.push_back({a, b, kernel});
    } else {  // no launch was made inside the scope (an argument check failed): nothing to time
      h->cur_a = h->cur_b = nullptr;
      h->ev_free.push_back(a);
      h->ev_free.push_back(b);
    }
  }
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
};

// The event pair of the due ProfScope, handed to ONE dispatch: every launcher - RCG_LAUNCH below, launch_dma and
// launch_dma_packed through launch_actor - takes it through prof_take, so the hand-off exists once.  A launcher that did
// not launch after all (no kernel instance) gives the pair back.  A scope times one launch: a second launch inside a
// due scope would go untimed and rcg_profile_read would under-report "the summed time of the kernel" - the -DRCG_DEV
// build aborts on it.
struct ProfPair {
  hipEvent_t a, b;
};
static inline ProfPair prof_take(rcg_handle* h) {
  const ProfPair p{h->cur_a, h->cur_b};
  h->cur_a = h->cur_b = nullptr;
#ifdef RCG_DEV
  if (h->scope_due) {
    if (!p.a) {
      fprintf(stderr, "librcg (dev): a second launch inside one due ProfScope would go untimed\n");
      abort();
    }
  }
#endif
  return p;
}
static inline void prof_give_back(rcg_handle* h, const ProfPair& p) {
  h->cur_a = p.a;
  h->cur_b = p.b;
}

// Launch on the handle's stream; inside a due ProfScope the launch carries the scope's event pair.
#define RCG_LAUNCH(h, kern, grid, block, lds, ...)                                                                 \
  do {                                                                                                             \
    rcg_handle* h__ = (h);                                                                                         \
    const ProfPair pp__ = prof_take(h__);                                                                          \
    if (pp__.a) {                                                                                                  \
      hipExtLaunchKernelGGL(kern, grid, block, (std::uint32_t)(lds), h__->stream, pp__.a, pp__.b, 0, __VA_ARGS__); \
    } else {                                                                                                       \
      hipLaunchKernelGGL(kern, grid, block, lds, h__->stream, __VA_ARGS__);                                        \
    }                                                                                                              \
  } while (0)

template <typename real>
inline const rcg::KParams<real>& params(const rcg_handle* h);
template <>
inline const rcg::KParams<float>& params<float>(const rcg_handle* h) {
  return h->p32;
}
template <>
inline const rcg::KParams<double>& params<double>(const rcg_handle* h) {
  return h->p64;
}

// LDS bytes one wave of k_actor_opt needs for this handle (the launcher sizes its blocks with it; rcg_control_tick_opt
// refuses a tick whose optimiser cannot be launched BEFORE it steps the env)
// Curvature pairs k_actor_opt keeps for this handle.  Default: 4 where the problem needs them - the critic modes and the
// non-diagonal stage costs, whose terminal / coupled terms carry 1e4 times the curvature of the rest (steepest descent
// stalls 3-14 % above SLSQP there, fixtures F8c) - and none for MPC with a diagonal R1, where box-scaled steepest descent
// reaches SLSQP's optimum on every decision of the reference's own loops (fixtures F8, mpc_tick_* of F7c: <= 3.5e-4 after
// 10 iterations, 0 after 30) and the pairs would cost 2.4 x the time (LDS footprint: 77 KB per block instead of 26).
// A long horizon (the reference's is unbounded) keeps fewer pairs by default: as many of the 4 as leave a wave's working set
// inside the CU's 160 KB of LDS (3-wheel robot, float64, critic modes: 4 pairs up to Nactor = 41, none from 63).
static inline int opt_memory_of(const rcg_handle* h) {
  if (h->opt_memory >= 0) return h->opt_memory;
  const bool generic = !(h->cfg.mode == RCG_MODE_MPC && h->p32.stage_kind == 0);
  const int dcw = h->cfg.mode != RCG_MODE_MPC ? h->dc : 0;
  int mem = generic ? 4 : 0;
  while (mem > 0 && (size_t)rcg::opt_lds_reals(h->cfg.n_actor, h->ds, h->du, h->np, dcw, mem) * h->esz > (size_t)160 * 1024) --mem;
  return mem;
}
static inline size_t opt_wave_lds_bytes(const rcg_handle* h) {
  const int dcw = h->cfg.mode != RCG_MODE_MPC ? h->dc : 0;
  return (size_t)rcg::opt_lds_reals(h->cfg.n_actor, h->ds, h->du, h->np, dcw, opt_memory_of(h)) * h->esz;
}

static inline unsigned blocks_for(long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// Everything that depends on the system type, one table per environment (rcg_sys_inst.hip, part 0).
struct SysVTable {
  int (*rhs)(rcg_handle*, const void* state, const void* action, void* dstate, void* clipped, int32_t n, int32_t clip);
  int (*stage_obj)(rcg_handle*, const void* obs, const void* act, void* out, int32_t n);
  int (*critic)(rcg_handle*, const void* obs, const void* act, const void* w, void* out, int32_t n);
  int (*critic_cost)(rcg_handle*, const void* w, void* Jc);
  int (*actor)(rcg_handle*, const char* who, const void* cand, int K, const void* obs, const void* state_sys,
               const void* w, void* J, void* action, void* best_J, int32_t* best_idx, bool tick, bool sim_first);
  int (*sim_step)(rcg_handle*, int32_t n_substeps);
  // RQL / SQL bookkeeping between two decisions in ONE launch: [sim_step x n_substeps] -> [push] -> [fit]
  int (*critic_update)(rcg_handle*, int32_t n_substeps /* 0: no env step */, int32_t do_push, int32_t do_fit);
  int (*optimize)(rcg_handle*, int32_t iters, const void* obs, const void* state_sys, const void* u_init, int shift,
                  void* u_opt, void* action, void* best_J, int32_t* n_iter, bool tick, bool sim_first);
  int (*nominal)(rcg_handle*, const void* obs, void* action, void* lyap, void* theta, int32_t n, double gain,
                 const double* ctrl_pars, int32_t clip, bool tick);
  int (*ticks)(rcg_handle*, int32_t T, int32_t K, const void* cand);
  int (*rhs_full)(rcg_handle*, const void* state, const void* disturb, const void* action, const void* xi, void* dstate,
                  void* ddisturb, void* clipped, int32_t n, int32_t clip);
  int (*search)(rcg_handle*, int32_t K, int32_t rounds, int32_t round0, const void* obs, const void* state_sys,
                const void* centre, int shift, void* u_best, void* action, void* best_J, int32_t* best_idx, bool tick,
                bool sim_first);
  int (*ticks_mem)(rcg_handle*, int32_t T, int32_t K, const void* cand);  // RQL / SQL: T ticks in one launch (k_ticks_mem)
  // rcg_loop_step's glue kernel (k_loop): [ACTION := act_in] -> [sim step] -> [stage cost + pack into `out`]; act_in / out: pinned host
  int (*loop)(rcg_handle*, const double* act_in, int32_t n_substeps, int32_t do_sim, int32_t do_tail, int32_t decided, int32_t dc,
              double* out, double* flag, double seq);
};
extern const SysVTable kVt3WRobot, kVt3WRobotNI, kVt2Tank;
