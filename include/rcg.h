/* rcg.h - C ABI of the MI355X-native rcognita hot path (librcg.so).
 *
 * The reference (AIDynamicAction/rcognita, pure Python) has no FFI: its "plugin API" for this path
 * is the duck-typed class surface System / Simulator / CtrlOptPred.  This header is the boundary a
 * binding for that surface calls into; every entry point names the reference method(s) it
 * replaces (paths relative to the reference checkout).  The Python mirror of the class surface that
 * binds these symbols with ctypes lives in rcognita_amd/ (see INTEGRATION.md).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every function returns RCG_OK (0) or a negative rcg_status; rcg_last_error() gives the text.
 *   - Batched data is struct-of-arrays, env index innermost: a tensor written "[d][B]" stores
 *     component c of env b at element c*B + b.  Element type is the handle's dtype (f32 or f64).
 *   - Candidate action sequences are "[B][K][N][du]" (env-major, then candidate, then the
 *     reference's own flat step-major action_sqn layout, rcognita/controllers.py:1284).
 *   - "dev" pointers are HIP device pointers (e.g. torch.Tensor.data_ptr()); arguments documented
 *     with a `where` flag accept host or device memory.
 *   - One handle <-> one device <-> one HIP stream; calls on one handle are not re-entrant.
 *     All launches are asynchronous on the handle's stream unless the call returns host data.
 */
#ifndef RCG_H
#define RCG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCG_VERSION 119 /* 117 + rcg_loop_step (_begin / _end), rcg_set_optimizer_tol (round 6) */

/* ---- limits ------------------------------------------------------------------------------- */
#define RCG_MAX_DS 5    /* largest dim_state of the built-in systems            */
#define RCG_MAX_DU 2    /* largest dim_input                                    */
#define RCG_MAX_CHI 7   /* RCG_MAX_DS + RCG_MAX_DU                               */
#define RCG_MAX_PARS 5  /* largest parameter vector (Sys2Tank)                  */
#define RCG_MAX_DC 35   /* quad-lin critic on chi of length 7                   */
#define RCG_MAX_ROW 64  /* N*du reals per candidate row that the decision kernels stage in LDS tiles; longer rows (the
                           reference's horizon is unbounded, controllers.py:965) are walked straight from HBM by the generic
                           kernel (k_actor, DIRECT form): any Nactor runs, the short rows of every preset run fastest */
#define RCG_MAX_NACTOR 4096 /* sanity bound of rcg_create (int32 index arithmetic on [B][K][N][du]) */

/* ---- enums --------------------------------------------------------------------------------- */
typedef enum rcg_status {
  RCG_OK = 0,
  RCG_ERR_BAD_ARG = -1,
  RCG_ERR_HIP = -2,       /* a HIP runtime call failed; text in rcg_last_error           */
  RCG_ERR_NO_DEVICE = -3, /* no gfx950 device visible: the library has no CPU fallback   */
  RCG_ERR_UNSUPPORTED = -4,
  RCG_ERR_NONFINITE = -5  /* returned by rcg_episode_stats when any env is flagged       */
} rcg_status;

/* rcognita/systems.py:255 (Sys3WRobot), :353 (Sys3WRobotNI), :401 (Sys2Tank) */
typedef enum rcg_system { RCG_SYS_3WROBOT = 0, RCG_SYS_3WROBOT_NI = 1, RCG_SYS_2TANK = 2 } rcg_system;
/* CtrlOptPred modes, rcognita/controllers.py:1304-1326 */
typedef enum rcg_mode { RCG_MODE_MPC = 0, RCG_MODE_RQL = 1, RCG_MODE_SQL = 2 } rcg_mode;
/* stage_obj_struct, rcognita/controllers.py:1076-1082 */
typedef enum rcg_stage { RCG_STAGE_QUADRATIC = 0, RCG_STAGE_BIQUADRATIC = 1 } rcg_stage;
/* critic_struct, rcognita/controllers.py:1024-1039, 1204-1212 */
typedef enum rcg_critic_struct {
  RCG_CRITIC_QUAD_LIN = 0,
  RCG_CRITIC_QUADRATIC = 1,
  RCG_CRITIC_QUAD_NOMIX = 2,
  RCG_CRITIC_QUAD_MIX = 3
} rcg_critic_struct;
/* Element type of a handle's tensors and arithmetic.  The reference computes in float64 throughout (RCG_F64 agrees with it
 * to 1e-11 and is the default of the drop-in classes).  RCG_F32, measured against numbers the reference produced at robot
 * headings |alpha| of 30 ... 1e4 rad from float32-exact inputs (fixture F13, tests/test_hip_large_heading.py,
 * profiles/r06_large_heading.txt):
 *   - every operator as a MAP from the same inputs - rcg_rhs, rcg_actor_cost / rcg_actor_argmin / the decision of a tick on every
 *     kernel, one rcg_sim_step - stays inside 1e-5 relative (measured <= 4.7e-7 for the costs, <= 2.8e-6 for one env step) at
 *     EVERY heading: the float32 trig is v_sin / v_cos behind an exact two-constant reduction (4.5e-7 absolute out to 1e6 rad),
 *     the simulator's a three-constant reduction + minimax polynomials (1.2e-7);
 *   - a FREE-RUNNING float32 trajectory stores its heading to 2^-24 |alpha| per step, which no kernel can undo: after n steps the
 *     heading may have drifted n * 2^-24 * |alpha| and the position speed * time * that much.  Measured over 104 steps: inside
 *     3e-5 of the reference for |alpha| <= 72 rad (400 steps to -72 rad: 9.6e-6), 2.2e-4 at 1e3 rad, 2.3e-2 at 1e4 rad (the NI
 *     robot at 8 m/s).  Loops that spin the robot beyond ~100 rad and need trajectory-level agreement use RCG_F64. */
typedef enum rcg_dtype { RCG_F32 = 0, RCG_F64 = 1 } rcg_dtype;
typedef enum rcg_where { RCG_HOST = 0, RCG_DEVICE = 1 } rcg_where;

/* rcg_cfg.flags */
#define RCG_FLAG_HAS_TARGET 0x1          /* observation_target != [] (controllers.py:1069)              */
#define RCG_FLAG_PER_ENV_PARS 0x2        /* pars is a [np][B] device tensor (RCG_FIELD_PARS), not cfg.pars */
#define RCG_FLAG_REF_LAG 0x4             /* rollout starts from the state before the last substep
                                            (reference loop order, presets/main_3wrobot.py:425-428)    */
#define RCG_FLAG_ACCUM_EVERY_SUBSTEP 0x8 /* upd_accum_obj every sim step (controllers.py:1093 quirk)    */
#define RCG_FLAG_NO_CLIP 0x10            /* ctrl_bnds all zero <=> unconstrained (systems.py:241)       */
#define RCG_FLAG_DISTURB 0x20            /* System(is_disturb=1): full state [state, disturb] (systems.py:140-145) */

/* per-env tensors owned by the handle (rcg_set_field / rcg_get_field) */
typedef enum rcg_field {
  RCG_FIELD_STATE = 0,       /* [ds][B]  real    System._state / Simulator.state_full            */
  RCG_FIELD_ACTION = 1,      /* [du][B]  real    System.action == CtrlOptPred.action_curr (ZOH)   */
  RCG_FIELD_ACCUM = 2,       /* [B]      real    CtrlOptPred.accum_obj_val                        */
  RCG_FIELD_STEP_IDX = 3,    /* [B]      int32   control ticks done in the current episode        */
  RCG_FIELD_EPISODE_IDX = 4, /* [B]      int32                                                    */
  RCG_FIELD_STATUS = 5,      /* [B]      uint32  bit0: non-finite state seen, env frozen          */
  RCG_FIELD_PARS = 6,        /* [np][B]  real    only with RCG_FLAG_PER_ENV_PARS                  */
  RCG_FIELD_STATE_INIT = 7,  /* [ds][B]  real    Simulator.state_full_init                        */
  RCG_FIELD_STATE_PREV = 8,  /* [ds][B]  real    state before the last RK4 substep                */
  RCG_FIELD_BEST_J = 9,      /* [B]      real    J of the last argmin                             */
  RCG_FIELD_BEST_IDX = 10,   /* [B]      int32   winning candidate index of the last argmin       */
  RCG_FIELD_W_CRITIC = 11,   /* [dc][B]  real    CtrlOptPred.w_critic                             */
  RCG_FIELD_W_PREV = 12,     /* [dc][B]  real    CtrlOptPred.w_critic_prev                        */
  RCG_FIELD_OBS_BUF = 13,    /* [buffer_size][dy][B] real, newest row last (utilities.py:78)      */
  RCG_FIELD_ACT_BUF = 14,    /* [buffer_size][du][B] real                                         */
  RCG_FIELD_RETURNS = 15,    /* [B]      real    accum_obj of the last finished episode           */
  RCG_FIELD_ACTION_SQN = 16, /* [B][N][du] real  last optimised action sequence, one row per env  */
  RCG_FIELD_DISTURB = 17,    /* [dd][B] real  disturbance part of the full state (RCG_FLAG_DISTURB); dd = 2, 2, 1 */
  RCG_FIELD_SUBSTEP_IDX = 18, /* [B] int32  simulation substeps since the episode began: noise counter word 3  */
  RCG_FIELD_COUNT_ = 19
} rcg_field;

/* Plain-old-data configuration.  All reals are double here and are converted to the handle's
 * dtype once, at rcg_create.  Matrices are row-major n x n with n = ds + du. */
typedef struct rcg_cfg {
  int32_t struct_size;      /* = sizeof(rcg_cfg); checked by rcg_create                          */
  int32_t sys_id;           /* rcg_system                                                         */
  int32_t batch;            /* B >= 1                                                             */
  int32_t dtype;            /* rcg_dtype                                                          */
  int32_t device;           /* HIP device ordinal                                                 */
  int32_t n_actor;          /* Nactor (controllers.py:965), 1 <= N <= RCG_MAX_NACTOR; the on-device optimiser / search hold a
                               wave's working set in LDS and refuse (RCG_ERR_UNSUPPORTED, nothing touched) a horizon that
                               does not fit 160 KB / 64 KB: e.g. 3-wheel robot, float64, MPC: Nactor <= 95                */
  int32_t mode;             /* rcg_mode                                                           */
  int32_t stage_obj_struct; /* rcg_stage                                                          */
  int32_t critic_struct;    /* rcg_critic_struct                                                  */
  int32_t n_critic;         /* Ncritic; clipped to buffer_size-1 as controllers.py:1015          */
  int32_t buffer_size;      /* controllers.py:980-981                                             */
  int32_t substeps_per_tick; /* RK4 substeps of dt_sim per controller sampling period            */
  int32_t flags;            /* RCG_FLAG_*                                                         */
  int32_t critic_every_ticks; /* refit the critic every this many control ticks (critic_period /
                                 sampling_time, controllers.py:1466); 0 or 1: every tick          */
  double dt_sim;            /* RK4 step                                                           */
  double sampling_time;     /* controller sampling time (controllers.py:962)                      */
  double pred_step_size;    /* Euler step of the rollout (controllers.py:966)                     */
  double gamma;             /* discount (controllers.py:1012)                                     */
  double pars[8];           /* system parameters: 3wrobot (m, I); 2tank (tau1,tau2,K1,K2,K3)      */
  double ctrl_bnds[4];      /* [du][2] = (lo, hi) per input (systems.py:241-243)                  */
  double R1[49];            /* stage_obj_pars[0], leading dimension n                             */
  double R2[49];            /* stage_obj_pars[1] (biquadratic only)                               */
  double target[8];         /* observation_target (used iff RCG_FLAG_HAS_TARGET)                  */
  double action_init[4];    /* action applied before the first tick (controllers.py:973-978)      */
  double w_init[40];        /* w_critic_init (controllers.py:1041-1042: ones)                     */
  double w_min[40];         /* Wmin / Wmax (controllers.py:1026-1039)                             */
  double w_max[40];
  /* Disturbance model (RCG_FLAG_DISTURB), System.pars_disturb = [sigma, mu, tau] (systems.py:303-306, 337):
   * dq_k/dt = -tau_k (q_k + sigma_k (xi_k + mu_k)) (systems.py:343), xi ~ N(0,1) drawn once per RK4 substep from
   * Philox4x32-10 with counter (env id lo, env id hi, EPISODE_IDX, SUBSTEP_IDX) and key (seed lo, seed hi), env id =
   * env_id_base + index in this handle, so a shard reproduces exactly its slice of the unsharded run. */
  double pars_disturb[6];  /* sigma[2], mu[2], tau[2] */
  double disturb_init[2];  /* Simulator(disturb_init=...) (simulator.py:131-134) */
  uint64_t seed;
  int64_t env_id_base;
} rcg_cfg;

/* Per-shard summary of episode returns: what ranks exchange. */
typedef struct rcg_summary {
  double count;    /* envs in the shard                              */
  double sum;      /* sum of returns                                 */
  double sumsq;    /* sum of squared returns                         */
  double min;
  double max;
  double n_failed; /* envs whose status bit0 is set                  */
} rcg_summary;

typedef struct rcg_handle rcg_handle;

/* ---- library ------------------------------------------------------------------------------- */
int rcg_version(void);
/* Text of the last error on this handle; h == NULL: last error of a failed rcg_create in this
 * thread.  Never NULL. */
const char* rcg_last_error(const rcg_handle* h);
/* Number of visible HIP devices (0 if none / no driver).  Does not create a context. */
int rcg_device_count(void);

/* ---- life cycle: System.__init__ (systems.py:69-145), Simulator.__init__ (simulator.py:71-154),
 *      CtrlOptPred.__init__ (controllers.py:811-1044) ------------------------------------------ */
int rcg_create(const rcg_cfg* cfg, rcg_handle** out);
int rcg_destroy(rcg_handle* h);
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default. */
int rcg_set_stream(rcg_handle* h, void* hip_stream);
/* Give the handle a non-blocking HIP stream of its own (created here, destroyed by rcg_destroy) and switch to it: for
 * callers without a stream library that drive several independent handles - the segments of a mixed pool - side by side. */
int rcg_use_own_stream(rcg_handle* h);
/* Order everything this handle launches from now on AFTER the work queued on `producer_stream` so far (an event recorded
 * there, waited for on the handle's stream; NULL = the legacy default stream): for device-resident inputs - candidate
 * tensors, states - written on another stream than the handle's, e.g. torch's current stream when the handle runs on a
 * stream of its own.  No host synchronisation.  No-op when `producer_stream` is the handle's stream. */
int rcg_wait_stream(rcg_handle* h, void* producer_stream);
/* The reverse edge: order everything queued on `consumer_stream` from now on AFTER the work this handle has launched so far
 * (an event recorded on the handle's stream, waited for on `consumer_stream`).  For a producer that REWRITES or frees a
 * device-resident input - the candidate tensor of the next tick - on its own stream while this handle's kernels may still be
 * reading the previous contents.  No host synchronisation.  No-op when `consumer_stream` is the handle's stream. */
int rcg_release_stream(rcg_handle* h, void* consumer_stream);
int rcg_synchronize(rcg_handle* h);

/* Device-memory helpers so that a host language without a GPU array library can drive the ABI. */
int rcg_dev_alloc(rcg_handle* h, uint64_t bytes, void** dev_out);
int rcg_dev_free(rcg_handle* h, void* dev);
int rcg_memcpy_h2d(rcg_handle* h, void* dev_dst, const void* host_src, uint64_t bytes);
int rcg_memcpy_d2h(rcg_handle* h, void* host_dst, const void* dev_src, uint64_t bytes);

/* ---- per-env tensors ----------------------------------------------------------------------- */
/* Copy a whole field in or out (layout and element type: rcg_field).  Synchronous for RCG_HOST. */
int rcg_set_field(rcg_handle* h, int field, const void* src, int where);
int rcg_get_field(rcg_handle* h, int field, void* dst, int where);
/* Size in bytes of a field for this handle (0 if the field is not allocated). */
int64_t rcg_field_bytes(const rcg_handle* h, int field);
/* Raw device pointer of a field (zero-copy access; valid until rcg_destroy). */
int rcg_field_ptr(rcg_handle* h, int field, void** dev_out);

/* ---- stateless operators (unit parity; all pointers are device pointers) ------------------- */
/* Sys*._state_dyn (systems.py:308-323, 370-382, 412-419) when clip == 0;
 * System.closed_loop_rhs (systems.py:213-253) when clip != 0: the action is clipped to ctrl_bnds
 * first and the clipped value is written to clipped_action (may be NULL).
 * state [ds][n], action [du][n] -> dstate [ds][n].  Uses cfg.pars (or, with
 * RCG_FLAG_PER_ENV_PARS and n == B, the per-env parameters). */
int rcg_rhs(rcg_handle* h, const void* state, const void* action, void* dstate, void* clipped_action,
            int32_t n, int32_t clip);
/* System.closed_loop_rhs on the FULL state of a system with is_disturb = 1 (systems.py:213-253 with :308-345,
 * :370-394, :412-426), for n points: state [ds][n], disturb [dd][n], action [du][n], xi [dd][n] = the value randn()
 * returns for component k (the reference draws it inside the right-hand side; here it is an input) -> dstate [ds][n],
 * ddisturb [dd][n]; clipped_action (may be NULL) [du][n].  Needs a handle created with RCG_FLAG_DISTURB. */
int rcg_rhs_full(rcg_handle* h, const void* state, const void* disturb, const void* action, const void* xi,
                 void* dstate, void* ddisturb, void* clipped_action, int32_t n, int32_t clip);
/* The draw rcg_sim_step will use for each env of the handle at its current (EPISODE_IDX, SUBSTEP_IDX):
 * bits_out (device, [4][B] uint32, may be NULL) the four Philox words, xi_out (device, [2][B] real, may be NULL) the
 * two normals (Box-Muller on 24-bit uniforms of words 0 and 1).  Integer output is bit-exact by contract. */
int rcg_disturb_noise(rcg_handle* h, void* bits_out, void* xi_out);
/* CtrlOptPred.stage_obj (controllers.py:1063-1084): obs [dy][n], act [du][n] -> out [n]. */
int rcg_stage_obj(rcg_handle* h, const void* obs, const void* act, void* out, int32_t n);
/* CtrlOptPred._critic (controllers.py:1192-1214): w [dc][n] -> out [n]. */
int rcg_critic(rcg_handle* h, const void* obs, const void* act, const void* w, void* out, int32_t n);
/* CtrlOptPred._actor_cost (controllers.py:1273-1328) for K candidate sequences per env.
 * cand [B][K][N][du]; obs [dy][B] and state_sys [ds][B] (NULL: the handle's STATE for both);
 * w [dc][B] (NULL: the handle's W_CRITIC; ignored in MPC mode) -> J [B][K]. */
int rcg_actor_cost(rcg_handle* h, const void* cand, int32_t K, const void* obs, const void* state_sys,
                   const void* w, void* J);
/* CtrlOptPred._critic_cost (controllers.py:1216-1245) on the handle's buffers and W_PREV.
 * w [dc][B] (NULL: the handle's W_CRITIC) -> Jc [B]. */
int rcg_critic_cost(rcg_handle* h, const void* w, void* Jc);

/* ---- stateful steps ------------------------------------------------------------------------ */
/* Simulator.sim_step (simulator.py:156-168) x n_substeps: fixed-step RK4 of closed_loop_rhs with
 * the held ACTION, clipped (systems.py:241-243).  Updates STATE, STATE_PREV, STATUS. */
int rcg_sim_step(rcg_handle* h, int32_t n_substeps);
/* The same with the step length given by the caller: ONE simulation step of length `step` (> 0, finite), integrated as
 * n_substeps RK4 substeps of step / n_substeps; the handle's dt_sim is untouched.  The reference's Simulator.sim_step
 * advances by whatever its adaptive solver chose (simulator.py:156-168: `self.ODE_solver.step()`); this entry lets a
 * caller walk a recorded time grid of the reference - tests/test_hip_ref_traces.py replays the reference's closed loops at
 * the reference's own step and decision instants.  SUBSTEP_IDX / the disturbance draw advance as in rcg_sim_step. */
int rcg_sim_step_h(rcg_handle* h, int32_t n_substeps, double step);
/* One iteration of the reference's headless loop body (presets/main_3wrobot.py:419-429) in ONE call with ONE host wait, for the
 * drop-in classes at small batch (round 6; the separate calls cost three device-to-host round trips per simulation step):
 *   System.receive_action       ACTION := action_in (host, [B][du] doubles; NULL: the handle's own)
 *   Simulator.sim_step          one step of length step_h in n_substeps RK4 substeps; STATE_PREV := the state before it
 *   CtrlOptPred.compute_action  RCG_LOOP_PUSH (RQL / SQL): push_vec(action_curr, observation) on both buffers (utilities.py:78,
 *                               controllers.py:1463-1464); RCG_LOOP_FIT: the critic fit (controllers.py:1466-1471);
 *                               RCG_LOOP_DECIDE: rcg_actor_optimize, `iters` iterations from action_sqn_init, the rollout from
 *                               STATE_PREV - the loop hands the controller System._state one iteration late (controllers.py:
 *                               1056-1061) - with STATE as the observation; ACTION := the optimum's first action, ACTION_SQN := it
 *   CtrlOptPred.stage_obj       rho(STATE, ACTION)
 * out (host): [B][ds + du + 2 (+ dc in RQL / SQL)] doubles per env: state, action, stage_obj, best_J (NaN without
 * RCG_LOOP_DECIDE), W_CRITIC.  ACCUM / STEP_IDX are not touched (upd_accum_obj is the caller's multiply-add; the caller decides
 * from its clock which steps are samples).  Every number equals what the separate calls (rcg_set_field, rcg_sim_step_h,
 * rcg_critic_update, rcg_actor_optimize, rcg_stage_obj) leave, bit for bit.  Handles whose rows fit the 16-KB pinned buffer
 * (3-wheel robot, MPC: up to 200 envs); no disturbance model.  Launches: ONE for a step that is no sample (k_loop), ONE for a
 * sample with the plain MPC decision (diagonal stage cost, no curvature pairs - every MPC preset: k_actor_opt does the
 * iteration's head and tail itself, rcg_last_launch variant bit 3), three otherwise. */
#define RCG_LOOP_DECIDE 1
#define RCG_LOOP_PUSH 2
#define RCG_LOOP_FIT 4
int rcg_loop_step(rcg_handle* h, const double* action_in, double step_h, int32_t n_substeps, int32_t flags, int32_t iters,
                  double* out);
/* The same in two halves, so that the host can do its own bookkeeping of iteration i while the device already runs iteration
 * i + 1: rcg_loop_step_begin enqueues the iteration on the handle's stream and returns (action_in is copied before it returns);
 * rcg_loop_step_end waits for it - polling the sequence numbers the last kernel writes into the handle's pinned buffer, no stream
 * wait - and hands over the rows (out == NULL: the iteration is dropped - STATE, ACTION, the critic buffers and weights keep its
 * effects and the caller is expected to set them anew, ACTION_SQN stays the last collected decision's; a collected deciding step
 * makes its sequence ACTION_SQN by swapping two buffers, so a pointer from rcg_field_ptr(ACTION_SQN) is stale after it).  One iteration
 * may be pending per handle (a second begin: RCG_ERR_BAD_ARG); every other entry point is ordered behind a pending iteration on
 * the stream as usual.  rcg_loop_step is begin + end.  The drop-in classes use the pair to start the next simulation step at the
 * end of compute_action - the action it returns is the one the system will hold - while the loop body still logs the current
 * one (rcognita_amd/controllers.py `_speculate`). */
int rcg_loop_step_begin(rcg_handle* h, const double* action_in, double step_h, int32_t n_substeps, int32_t flags, int32_t iters);
int rcg_loop_step_end(rcg_handle* h, double* out);
/* Replacement of CtrlOptPred._actor_optimizer (controllers.py:1330-1427): evaluate _actor_cost for
 * K candidates per env and take the argmin (lower J wins, ties -> lower index, NaN = +inf).
 * cand [B][K][N][du], or NULL for the generated level grid (K levels for du = 1, g*g for du = 2).
 * obs / state_sys as rcg_actor_cost.  Outputs (each may be NULL): action [du][B] = first du
 * entries of the winner (controllers.py:1427), best_J [B], best_idx [B] int32.  Does not modify
 * the handle's ACTION.
 * Non-finite corner: the stage cost is the full sum chi' R1 chi as numpy evaluates it, so a component that overflows under a
 * ZERO weight makes J NaN (0 * inf) and the candidate counts as +inf, exactly as in the reference.  The generated grid with
 * the presets' R1 does not accumulate its zero-weighted terms (v, omega, F, M of the robots) step by step; it tests the
 * zero-weighted state components once, on the observation and on the last rolled-out state (a non-finite value is sticky
 * under the Euler step), with the same outcome (tests/test_hip_reset_and_guards.py). */
int rcg_actor_argmin(rcg_handle* h, const void* cand, int32_t K, const void* obs, const void* state_sys,
                     void* action, void* best_J, int32_t* best_idx);
/* One env.control-step for every env (the loop body of presets/main_3wrobot.py:419-429):
 * sim_step x substeps_per_tick -> [RQL/SQL: buffer push + critic fit] -> actor argmin over K
 * candidates -> ACTION := winner's first action -> ACCUM += stage_obj(obs, action)*sampling_time ->
 * STEP_IDX += 1.  cand as rcg_actor_argmin. */
int rcg_control_tick(rcg_handle* h, const void* cand, int32_t K);
/* T consecutive rcg_control_tick(h, NULL, K) - the loop of presets/main_3wrobot.py:415-468 for T sampling periods with
 * the generated candidate grid - in ONE kernel launch: each env's wave keeps state, held action, ACCUM and STEP_IDX (and,
 * with the disturbance model, the disturbance state and the noise counter) in registers and loops over {sim_step, K x
 * _actor_cost, argmin, upd_accum_obj}.  Every field ends bit-identical to T single ticks (same source expressions;
 * tests/test_hip_ticks.py checks it for every system, element type and shape) - with ONE stated exception, Sys2Tank's BEST_J:
 * its rollout right-hand side leaves the choice of fused multiply-adds to the compiler (18-26 % faster than the written-out
 * form), and the persistent kernels and the per-tick kernels inline it into different surroundings, so a cost may differ by a
 * rounding of its terms (measured on the library as built, float32, streamed candidates: BEST_J of one env in about 4 000, by 1 to
 * 40 ulp where the critic's signed terms cancel; tools/fuzz_parity.py).
 * The decision - and with it STATE, ACTION, ACCUM and everything downstream - differs only where two candidates' costs tie to the
 * last bit.  The robots' right-hand sides and every simulator step (one set of bits in every kernel: rk4_step) write their fusions
 * out.  A checkpoint taken under one entry point and resumed under the other inherits the same caveat.  BEST_J / BEST_IDX are the last tick's.  MPC handles: any stage-cost structure,
 * with or without RCG_FLAG_DISTURB.  RQL / SQL handles (1 <= Ncritic - 1 <= 8, no disturbance model, the preset's observation
 * target setting): the two launches of a tick - env step + buffer push + critic fit, then the decision - run as phases of
 * one persistent launch (k_ticks_mem), same functions on the same memory, bit-identical as well (critic structures with 20 or
 * more weights, whose single ticks fit with four lanes per env, run that four-lane walk as their critic phase; they need
 * K >= 4); other RQL / SQL handles get RCG_ERR_UNSUPPORTED and loop rcg_control_tick.  Removes the launch-bound regime of
 * small batches. */
int rcg_control_ticks(rcg_handle* h, int32_t T, int32_t K);
/* T consecutive rcg_control_tick(h, cand, K) issued by ONE call: the loop of presets/main_3wrobot.py:415-468 for T sampling
 * periods with the SAME candidate tensor (or the generated grid, cand == NULL) at every tick, any mode.  Handles of up to
 * 16384 envs run them as ONE launch: MPC on k_ticks (a caller's tensor: the wave's candidate rows are staged into LDS once
 * and re-walked T times; a tensor whose rows do not fit a wave's 32 KB AND that exceeds 128 MB - half the Infinity Cache -
 * would be re-staged from HBM by plain loads every tick and loops single ticks on k_actor_dma instead), RQL / SQL on k_ticks_mem
 * - with the generated grid, and since round 5 with a caller's tensor as well (diagonal stage cost; the decision phase walks the
 * tensor with the accumulation order of k_actor_dma / k_actor_dma_packed, the kernels of the single ticks); larger batches issue
 * the launches of T single ticks without T trips through the
 * caller's FFI (a Python caller needs ~12 us per call, and a GPU that idles between short ticks clocks down).  Either way
 * every field ends as T single calls leave it, bit for bit (Sys2Tank's BEST_J: to a rounding of its terms - see
 * rcg_control_ticks); stops at the first error. */
int rcg_control_tick_n(rcg_handle* h, const void* cand, int32_t K, int32_t T);
/* A tick in two halves.  An RQL / SQL handle whose decision streams a caller's tensor through k_actor_dma (no disturbance
 * model, 1 .. 8 TD rows) runs rcg_control_tick for the envs [0, B / 2) and [B / 2, B) on two internal streams: the critic fit of
 * one half (bound by the latency of its longest active-set walk, 63-76 us at configs[2] whatever the batch) then runs under the
 * streaming kernel of the other, tick after tick - what two handles on two streams did from outside (the controller
 * loop being replaced, controllers.py:1458-1477, is per env: the halves never meet).  Every field ends bit-identical to the
 * unsplit tick.  WHICH MODE PIPELINES: a split tick returns with half the batch still on the two internal streams, and the
 * handle's stream does not wait for them until the next entry point is called (reads, writes, rcg_synchronize,
 * rcg_release_stream ... every entry point rejoins; rcg_join does only that, without a host wait).  Work a caller enqueues
 * itself on the stream it gave rcg_set_stream - a kernel reading rcg_field_ptr(ACTION) - would therefore see an unfinished
 * tick.  So:
 *   parts = 0 (default)  never splits on a caller's stream (rcg_set_stream) or the null stream: everything the tick launched is
 *                        on that stream when rcg_control_tick returns, at every batch size.  On a stream the handle owns
 *                        (rcg_use_own_stream; the caller has no handle on it and orders against it through rcg_wait_stream /
 *                        rcg_release_stream / rcg_synchronize, which rejoin) an eligible tick is split from 65 536 envs;
 *   parts = 1            never;
 *   parts = 2            the explicit opt-in on any stream: whenever the tick is eligible.  Contract: call rcg_join (or any
 *                        other entry point) before work of your own on the handle's stream touches the handle's fields.
 * rcg_last_launch reports a half-batch launch with bit 12 (4096) of `variant`. */
int rcg_set_tick_parts(rcg_handle* h, int32_t parts);
int rcg_join(rcg_handle* h);
/* On-device replacement of the SLSQP call of CtrlOptPred._actor_optimizer (controllers.py:1373-1398) for every mode
 * (MPC / RQL / SQL, controllers.py:1304-1326; RQL / SQL read the handle's W_CRITIC), stage-cost structure (diagonal or
 * full R1, biquadratic) and critic structure: `iters` iterations of projected limited-memory quasi-Newton descent -
 * adjoint gradient of _actor_cost w.r.t. the whole sequence (closed-form gradients of the stage cost and of w . phi), L-BFGS
 * direction over the last `memory` accepted steps on the free coordinates (box-scaled steepest descent without pairs), 16
 * trial step lengths evaluated in parallel (quasi-Newton: 4 .. 2^-13 times the unit step; steepest descent: 4 box widths down
 * to 2^-28), best feasible trial kept if it lowers J.  Deterministic, no finite differences.  obs / state_sys as
 * rcg_actor_cost; u_init [B][N][du] (NULL: action_sqn_init = action_init tiled, as the reference starts every call);
 * outputs, each may be NULL:
 * u_opt [B][N][du], action [du][B] (first du entries, controllers.py:1427), best_J [B], n_iter [B] int32 (accepted steps).
 * On the reference's own decisions (fixtures F8 / F8c: SLSQP's result at 12-32 states per system, mode and critic
 * structure) 30 iterations end within 0.5 % of SLSQP's cost. */
int rcg_actor_optimize(rcg_handle* h, int32_t iters, const void* obs, const void* state_sys, const void* u_init,
                       void* u_opt, void* action, void* best_J, int32_t* n_iter);
/* Curvature pairs rcg_actor_optimize keeps per env: 0 (projected steepest descent, round 3's optimiser) .. 8, or -1 for the
 * default: 4 for RQL / SQL and for non-diagonal stage costs, 0 for MPC with a diagonal R1 (steepest descent reaches SLSQP's
 * optimum on every MPC decision of the reference's loops; the pairs cost 2.4 x the time).  Each pair costs 2 * N * du reals
 * of LDS per env. */
int rcg_set_optimizer(rcg_handle* h, int32_t memory);
/* Stopping tolerance of rcg_actor_optimize, the counterpart of the accuracy the reference hands SLSQP (`tol=1e-7`,
 * controllers.py:1396 -> SciPy's `ftol`: SLSQP ends once an iteration changes J by less than that): an env is done after an
 * ACCEPTED step that lowered J by no more than `ftol` (absolute, in the units of J; the step is kept).  0 (the default of a
 * handle): no such test - an env runs until `iters` steps are done or a steepest-descent line search finds nothing better.  The
 * drop-in CtrlOptPred sets the reference's 1e-7. */
int rcg_set_optimizer_tol(rcg_handle* h, double ftol);
/* rcg_control_tick with rcg_actor_optimize as the decision: sim_step -> [RQL/SQL: buffer push + critic fit] -> optimise ->
 * ACTION, ACTION_SQN, BEST_J -> ACCUM, STEP_IDX.  warm_start != 0: start from the previous tick's optimum shifted by one
 * step (the reference always restarts from action_sqn_init: warm_start = 0). */
int rcg_control_tick_opt(rcg_handle* h, int32_t iters, int32_t warm_start);
/* Device-side candidate search, the sampling counterpart of rcg_actor_optimize (replacement of controllers.py:1330-1427 that
 * needs no gradient): `rounds` rounds of K candidate sequences per env, GENERATED on the device, evaluated with
 * _actor_cost where they are generated (they never exist in HBM) and refined around the round's winner - one launch, every
 * mode and cost structure.  Round r (r = 0 .. rounds - 1) perturbs its centre - round 0: `centre` [B][N][du] (NULL:
 * action_sqn_init) - by sigma_r xi with sigma_r = 0.5 (hi - lo) 2^-r, clipped to the box; candidate 0 is the centre itself
 * (the cost never increases from round to round), candidate 1 of round 0 is action_sqn_init, candidates below K / 2 hold one
 * draw per input over the horizon, the others draw per input and step.  xi: Philox4x32-10 keyed by (cfg.seed, global env id,
 * EPISODE_IDX, STEP_IDX) and indexed by (candidate, chunk, round) - independent of batch size, sharding and launch geometry;
 * Box-Muller in float32 (rcg_search.hpp).  K >= 64.  obs / state_sys as rcg_actor_cost; outputs, each may be NULL:
 * u_best [B][N][du] (the last round's winner), action [du][B] (its first du entries), best_J [B], best_idx [B] int32 (the
 * winner's index in the last round; 0 = the incumbent was kept). */
int rcg_actor_search(rcg_handle* h, int32_t K, int32_t rounds, const void* obs, const void* state_sys, const void* centre,
                     void* u_best, void* action, void* best_J, int32_t* best_idx);
/* rcg_control_tick with rcg_actor_search as the decision: sim_step -> [RQL/SQL: buffer push + critic fit] -> search ->
 * ACTION, ACTION_SQN, BEST_J, BEST_IDX -> ACCUM, STEP_IDX.  warm_start != 0: round 0 is centred on the previous tick's
 * optimum shifted by one step (from the episode's second tick on). */
int rcg_control_tick_search(rcg_handle* h, int32_t K, int32_t rounds, int32_t warm_start);
/* The producer alone: the K candidate rows rcg_actor_search evaluates in round `round` around `centre` [B][N][du] (NULL:
 * action_sqn_init) at the handle's current (EPISODE_IDX, STEP_IDX), written to cand [B][K][N][du] (device) - for callers
 * that stream candidates through rcg_actor_cost / rcg_actor_argmin / rcg_control_tick, and for tests. */
int rcg_candidates_sample(rcg_handle* h, void* cand, int32_t K, int32_t round, const void* centre);
/* Nominal (benchmark / safe-fallback) controllers for n points, replacing CtrlNominal3WRobot.compute_action_vanila /
 * compute_action / compute_LF (rcognita/controllers.py:1495-1755) and CtrlNominal3WRobotNI's
 * (controllers.py:1757-1956); the handle's system selects which.  obs: device [ds][n]; action (may be NULL): device
 * [du][n]; lyap (may be NULL): device [n], the Lyapunov function value of compute_LF.  ctrl_pars: host (m, I) of the
 * 3wrobot controller's constructor, NULL = the handle's pars; ignored for 3wrobotNI.  clip != 0: clip to ctrl_bnds as
 * compute_action does.  theta* of the 3wrobot controller (SciPy trust-constr in the reference) is build-defined:
 * a compass search from theta = 0 (step 0.25, halved whenever neither neighbour is lower, until 1e-9; round 6): the reference's
 * minimiser on 94.8 % of its fixture, Fc(theta*) never above the reference's (rounds 2-5: grid walk + golden section, 92.7 %).
 * Sys2Tank has no nominal controller: RCG_ERR_UNSUPPORTED. */
int rcg_nominal_action(rcg_handle* h, const void* obs, void* action, void* lyap, int32_t n, double ctrl_gain,
                       const double* ctrl_pars, int32_t clip);
/* theta* = argmin_theta Fc of CtrlNominal3WRobot._minimizer_theta (controllers.py:1618-1634; SciPy trust-constr there, the
 * build-defined local search here) for n points: obs device [ds][n] -> theta device [n] (the handle's real), wrapped into
 * [-pi, pi].  What rcg_nominal_action uses internally; exposed so that the search and the control law can be checked
 * separately (the law is not Lipschitz in theta).  Sys3WRobot handles only. */
int rcg_nominal_theta(rcg_handle* h, const void* obs, void* theta, int32_t n);
/* One control tick under the nominal controller ('--ctrl_mode nominal', presets/main_3wrobot.py:425 through
 * ctrl_selector, controllers.py:58-59): sim_step -> ACTION := clipped nominal action of STATE -> ACCUM, STEP_IDX. */
int rcg_control_tick_nominal(rcg_handle* h, double ctrl_gain, const double* ctrl_pars);
/* RQL/SQL bookkeeping of CtrlOptPred.compute_action (controllers.py:1458-1477): push (ACTION, obs)
 * into the buffers and, if do_fit != 0, refit W_CRITIC by bounded least squares on the TD stack of
 * _critic_cost (replacement of _critic_optimizer, controllers.py:1248-1271); W_PREV := W_CRITIC. */
int rcg_critic_update(rcg_handle* h, int32_t do_fit);
/* Episode boundary (Simulator.reset, simulator.py:197-204; CtrlOptPred.reset, controllers.py:1046):
 * RETURNS := ACCUM; ACCUM := 0; STATE := STATE_INIT; ACTION := action_init; STEP_IDX := 0;
 * EPISODE_IDX += 1.  Critic weights and buffers are retained, as in the reference. */
int rcg_episode_reset(rcg_handle* h);
/* Summary of `RETURNS` (from_accum == 0) or of the running ACCUM (from_accum != 0) over this
 * handle's envs; returns_out (host, [B] real, may be NULL) receives the per-env values. */
int rcg_episode_stats(rcg_handle* h, int32_t from_accum, void* returns_out, rcg_summary* out);

/* ---- checkpoint / resume --------------------------------------------------------------------- */
/* Everything a handle carries between ticks is the per-env tensors above (rcg_get_field / rcg_set_field) plus ONE host
 * counter: the control ticks issued in the current episode, which drives the critic period (fits on ticks k-1, 2k-1, ...
 * of an episode) and the warm start of rcg_control_tick_opt.  A checkpoint = all allocated fields + this counter; a
 * handle created with the same rcg_cfg and restored from it continues bit-identically
 * (rcognita_amd.Engine.checkpoint / restore).  The reference has no checkpointing: learned parameters persist in the
 * controller object (controllers.py:1046-1054). */
int64_t rcg_tick_count(const rcg_handle* h);
int rcg_set_tick_count(rcg_handle* h, int64_t ticks);

/* ---- measurement ---------------------------------------------------------------------------- */
typedef enum rcg_kernel { RCG_KERNEL_ACTOR = 0, RCG_KERNEL_SIM = 1, RCG_KERNEL_CRITIC = 2, RCG_KERNEL_COUNT_ = 3 } rcg_kernel;
/* kernel_mask bits 0..7: bit k set = bracket launches of rcg_kernel k with HIP events recorded on the
 * handle's own stream (1 = the actor kernel only, 7 = all); bits 8..19: sampling stride n (0/1 = every
 * launch, n = every n-th launch of each kernel); bits 20..: launches to let pass before the first sample
 * (< n).  A non-zero mask also resets the totals; 0 stops (and waits for the samples still in flight);
 * RCG_PROFILE_PAUSE stops sampling without waiting for anything or resetting anything, so that a measured region can be
 * closed from the host while its launches are still queued. */
#define RCG_PROFILE_PAUSE 0x80
int rcg_profile(rcg_handle* h, int32_t kernel_mask);
/* Synchronise, then return the summed device time (ms) and number of launches of one kernel since
 * the last rcg_profile(h, 1). */
int rcg_profile_read(rcg_handle* h, int32_t kernel, double* total_ms, int64_t* launches);
/* Synchronise, then copy the duration (ms) of each sampled launch of one kernel since the last rcg_profile(h, mask != 0),
 * in launch order, into ms_out[0 .. min(cap, n)) and return n in *n_out (the library keeps the first 65536 samples): the
 * median / min that SURVEY.md 8d asks for are computed by the caller.  A sample is the dispatch's own start / end stamps
 * (hipExtLaunchKernelGGL events, the figures a rocprofv3 kernel trace shows), not a pair of markers around it. */
int rcg_profile_samples(rcg_handle* h, int32_t kernel, double* ms_out, int64_t cap, int64_t* n_out);

/* Which kernel served the handle's last launch of a kind (rcg_kernel: the decision, the env step, the critic
 * bookkeeping) - the dispatch rule lives in the library (rcg_sysops.hpp::launch_actor) and callers that claim "the
 * production kernel" or label a roofline ask instead of re-deriving it. */
typedef enum rcg_kernel_id {
  RCG_KID_NONE = 0,       /* nothing of that kind launched yet                                                       */
  RCG_KID_ACTOR = 1,      /* k_actor: tile kernel, streamed (VGPR -> LDS staging) or generated candidates             */
  RCG_KID_ACTOR_DMA = 2,  /* k_actor_dma: the production streamed kernel (LDS-DMA tiles, unrolled register rollout)   */
  RCG_KID_TICKS = 3,      /* k_ticks: T ticks per launch (rcg_control_ticks)                                          */
  RCG_KID_ACTOR_OPT = 4,  /* k_actor_opt (rcg_actor_optimize / rcg_control_tick_opt)                                  */
  RCG_KID_NOMINAL = 5,    /* k_nominal                                                                                */
  RCG_KID_SIM = 6,        /* k_sim: lane = env                                                                        */
  RCG_KID_SIM_V = 7,      /* k_sim_v: lane = 16 bytes of consecutive envs                                             */
  RCG_KID_SIM_DIST = 8,   /* k_sim_dist: env step on the full state [state, disturb]                                  */
  RCG_KID_CRITIC_FIT = 9, /* k_critic_fit: [env step] + push + [fit]                                                  */
  RCG_KID_ACTOR_DMA_PACKED = 10, /* k_actor_dma_packed: k_actor_dma's data path with 64 / K envs per tile (4 <= K <= 32) */
  RCG_KID_ACTOR_SEARCH = 11, /* k_actor_search: candidates generated, evaluated and refined in one launch (rcg_actor_search) */
  RCG_KID_COUNT_ = 12
} rcg_kernel_id;
/* variant: k_actor_dma, k_actor_dma_packed: 0 MPC gamma = 1, 1 MPC discounted, 2 + critic_struct RQL, 6 + critic_struct SQL (k_actor_dma_packed: + 16 when the env step of the tick ran inside the launch);
 * k_actor_dma, round 6: 10 / 11 MPC with a biquadratic (diagonal) / full-matrix stage cost, 12 + critic_struct RQL with such a cost (SQL: its ordinary instances); k_actor / k_ticks: bit 0 generic
 * stage cost / critic modes, bit 1 observation target, bit 2 streamed candidates, bit 3 the hand-packed generated-grid instance, bit 4 (k_actor) rows beyond RCG_MAX_ROW walked straight from HBM; k_critic_fit: critic_struct + 16 * (rows
 * the instance is compiled for; 0: any) + 256 * do_sim + 512 * do_fit + 1024 * (the four-lanes-per-env form, structures with >= 9 weights) + 2048 * (the any-number-of-rows form, Ncritic - 1 > 8); k_actor_opt: bit 0 generic
 * stage cost / critic modes, bit 1 observation target, bit 2 curvature pairs, bit 3 the launch also did rcg_loop_step's head and tail; others 0.  envs_per_wave: envs a wave owns (k_actor_dma, k_actor_dma_packed) or
 * packs into one 64-row tile (k_actor, k_ticks); 64 for lane = env kernels.  Bit 12 (4096) of variant: the launch served one half
 * of a split tick (rcg_set_tick_parts).  Each out pointer may be NULL. */
int rcg_last_launch(const rcg_handle* h, int32_t kind, int32_t* kernel_id, int32_t* variant, int32_t* envs_per_wave);
/* "k_actor_dma", ... ; "?" for an unknown id.  Never NULL. */
const char* rcg_kernel_name(int32_t kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* RCG_H */
