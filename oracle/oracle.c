/* oracle.c - plain-C (float64) restatement of the rcognita hot path, OpenMP over envs.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/liboracle.so, loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg (kind "port").  The product library
 * (librcg.so) never links or calls it.
 *
 * Parity status: pinned against the golden vectors generated from the reference
 * (tests/golden/F*.npz, tests/test_c_oracle.py) and against the numpy oracle (oracle/rcg_oracle.py).
 * Citations are to the reference checkout (/root/reference).  Layout here is array-of-structs,
 * row-major [B][d], the natural numpy layout of the reference's vectors with a batch axis in front.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXS 5
#define MAXU 2
#define MAXC 7

typedef struct orc_cfg {
  int32_t sys_id;        /* 0 3wrobot, 1 3wrobotNI, 2 2tank                      */
  int32_t n_actor;
  int32_t mode;          /* 0 MPC, 1 RQL, 2 SQL                                  */
  int32_t biquad;        /* stage_obj_struct == 'biquadratic'                    */
  int32_t critic_struct; /* 0 quad-lin, 1 quadratic, 2 quad-nomix, 3 quad-mix    */
  int32_t has_target;
  int32_t clip;          /* ctrl_bnds.any()                                      */
  int32_t substeps_per_tick;
  double gamma, h_pred, dt_sim, sampling_time;
  double pars[8];
  double lo[2], hi[2];
  double R1[49], R2[49]; /* row-major n x n, n = ds + du                         */
  double target[8];
} orc_cfg;

static const int DIMS[3][2] = {{5, 2}, {3, 2}, {2, 1}};

/* Sys*._state_dyn: rcognita/systems.py:308-323 (3wrobot), 370-382 (3wrobotNI), 412-419 (2tank) */
void orc_state_dyn(const orc_cfg* c, const double* x, const double* u, const double* p, double* d) {
  switch (c->sys_id) {
    case 0:
      d[0] = x[3] * cos(x[2]);
      d[1] = x[3] * sin(x[2]);
      d[2] = x[4];
      d[3] = 1.0 / p[0] * u[0];
      d[4] = 1.0 / p[1] * u[1];
      break;
    case 1:
      d[0] = u[0] * cos(x[2]);
      d[1] = u[0] * sin(x[2]);
      d[2] = u[1];
      break;
    default:
      d[0] = 1.0 / p[0] * (-x[0] + p[2] * u[0]);
      d[1] = 1.0 / p[1] * (-x[1] + p[3] * x[0] + p[4] * (x[1] * x[1]));
  }
}

/* the clip of System.closed_loop_rhs, rcognita/systems.py:241-243 */
static void clip_action(const orc_cfg* c, const double* u, double* uc) {
  const int du = DIMS[c->sys_id][1];
  for (int k = 0; k < du; ++k) {
    double v = u[k];
    if (c->clip) v = v < c->lo[k] ? c->lo[k] : (v > c->hi[k] ? c->hi[k] : v);
    uc[k] = v;
  }
}

/* classical RK4 of closed_loop_rhs under a held action (build-defined) */
void orc_rk4(const orc_cfg* c, double* x, const double* u, const double* p, double h) {
  const int ds = DIMS[c->sys_id][0];
  double uc[MAXU], k1[MAXS], k2[MAXS], k3[MAXS], k4[MAXS], t[MAXS];
  clip_action(c, u, uc);
  orc_state_dyn(c, x, uc, p, k1);
  for (int i = 0; i < ds; ++i) t[i] = x[i] + (0.5 * h) * k1[i];
  orc_state_dyn(c, t, uc, p, k2);
  for (int i = 0; i < ds; ++i) t[i] = x[i] + (0.5 * h) * k2[i];
  orc_state_dyn(c, t, uc, p, k3);
  for (int i = 0; i < ds; ++i) t[i] = x[i] + h * k3[i];
  orc_state_dyn(c, t, uc, p, k4);
  for (int i = 0; i < ds; ++i) x[i] = x[i] + (h / 6.0) * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
}

static void make_chi(const orc_cfg* c, const double* y, const double* u, double* chi) {
  const int ds = DIMS[c->sys_id][0], du = DIMS[c->sys_id][1];
  for (int i = 0; i < ds; ++i) chi[i] = c->has_target ? y[i] - c->target[i] : y[i];
  for (int k = 0; k < du; ++k) chi[ds + k] = u[k];
}

/* CtrlOptPred.stage_obj, rcognita/controllers.py:1063-1084 */
double orc_stage_obj(const orc_cfg* c, const double* y, const double* u) {
  const int n = DIMS[c->sys_id][0] + DIMS[c->sys_id][1];
  double chi[MAXC], q = 0.0;
  make_chi(c, y, u, chi);
  for (int j = 0; j < n; ++j) { /* (chi @ R1) @ chi */
    double v = 0.0;
    for (int i = 0; i < n; ++i) v += chi[i] * c->R1[i * n + j];
    q += v * chi[j];
  }
  if (c->biquad) {
    double c2[MAXC], q4 = 0.0;
    for (int i = 0; i < n; ++i) c2[i] = chi[i] * chi[i];
    for (int j = 0; j < n; ++j) {
      double v = 0.0;
      for (int i = 0; i < n; ++i) v += c2[i] * c->R2[i * n + j];
      q4 += v * c2[j];
    }
    q = q4 + q;
  }
  return q;
}

/* CtrlOptPred._critic, rcognita/controllers.py:1192-1214 (uptria2vec: utilities.py:81-96) */
double orc_critic(const orc_cfg* c, const double* y, const double* u, const double* w) {
  const int ds = DIMS[c->sys_id][0], du = DIMS[c->sys_id][1], n = ds + du;
  double chi[MAXC], acc = 0.0;
  int idx = 0;
  make_chi(c, y, u, chi);
  switch (c->critic_struct) {
    case 0:
    case 1:
      for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) acc += w[idx++] * (chi[i] * chi[j]);
      if (c->critic_struct == 0)
        for (int i = 0; i < n; ++i) acc += w[idx++] * chi[i];
      break;
    case 2:
      for (int i = 0; i < n; ++i) acc += w[i] * (chi[i] * chi[i]);
      break;
    default: /* quad-mix: raw observation, target ignored (controllers.py:1212) */
      for (int i = 0; i < ds; ++i) acc += w[idx++] * (y[i] * y[i]);
      for (int i = 0; i < ds; ++i)
        for (int k = 0; k < du; ++k) acc += w[idx++] * (y[i] * u[k]);
      for (int k = 0; k < du; ++k) acc += w[idx++] * (u[k] * u[k]);
  }
  return acc;
}

/* CtrlOptPred._actor_cost, rcognita/controllers.py:1273-1328 (is_est_model = 0) */
double orc_actor_cost(const orc_cfg* c, const double* useq, const double* obs, const double* state_sys,
                      const double* p, const double* w) {
  const int ds = DIMS[c->sys_id][0], du = DIMS[c->sys_id][1], N = c->n_actor;
  double x[MAXS], y[MAXS], d[MAXS], J = 0.0, g = 1.0;
  memcpy(x, state_sys, ds * sizeof(double));
  memcpy(y, obs, ds * sizeof(double));
  for (int k = 0; k < N; ++k) {
    const double* u = useq + k * du;
    if (k > 0) {
      orc_state_dyn(c, x, useq + (k - 1) * du, p, d); /* Euler, unclipped (controllers.py:1294) */
      for (int i = 0; i < ds; ++i) {
        x[i] = x[i] + c->h_pred * d[i];
        y[i] = x[i];
      }
    }
    if (c->mode == 0)
      J += g * orc_stage_obj(c, y, u);
    else if (c->mode == 1)
      J += (k < N - 1) ? g * orc_stage_obj(c, y, u) : orc_critic(c, y, u, w);
    else
      J += orc_critic(c, y, u, w);
    g *= c->gamma;
  }
  return J;
}

/* batched operator: cand [B][K][N][du] -> J [B][K] */
void orc_actor_cost_batch(const orc_cfg* c, int B, int K, const double* cand, const double* obs,
                          const double* state_sys, const double* pars, int per_env_pars, const double* w, int dc,
                          double* J, int nthreads) {
  const int ds = DIMS[c->sys_id][0], du = DIMS[c->sys_id][1], R = c->n_actor * du;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int b = 0; b < B; ++b) {
    const double* p = per_env_pars ? pars + (size_t)b * 8 : pars;
    for (int k = 0; k < K; ++k)
      J[(size_t)b * K + k] = orc_actor_cost(c, cand + ((size_t)b * K + k) * R, obs + (size_t)b * ds,
                                            state_sys + (size_t)b * ds, p, w ? w + (size_t)b * dc : 0);
  }
}

/* One env.control-step for B envs (loop body of presets/main_3wrobot.py:419-429 with the build's
 * RK4 + K-candidate argmin): sim -> argmin_k _actor_cost -> action -> accum -> step_idx. */
void orc_control_tick(const orc_cfg* c, int B, int K, const double* cand, double* state, double* action,
                      double* accum, int32_t* step_idx, const double* pars, int per_env_pars, const double* w,
                      int dc, double* best_J, int32_t* best_idx, int nthreads) {
  const int ds = DIMS[c->sys_id][0], du = DIMS[c->sys_id][1], R = c->n_actor * du;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int b = 0; b < B; ++b) {
    const double* p = per_env_pars ? pars + (size_t)b * 8 : pars;
    double* x = state + (size_t)b * ds;
    double* u = action + (size_t)b * du;
    for (int s = 0; s < c->substeps_per_tick; ++s) orc_rk4(c, x, u, p, c->dt_sim);
    double bj = INFINITY;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      double J = orc_actor_cost(c, cand + ((size_t)b * K + k) * R, x, x, p, w ? w + (size_t)b * dc : 0);
      if (J != J) J = INFINITY;
      if (J < bj) {
        bj = J;
        bi = k;
      }
    }
    for (int i = 0; i < du; ++i) u[i] = cand[((size_t)b * K + bi) * R + i];
    accum[b] += orc_stage_obj(c, x, u) * c->sampling_time;
    step_idx[b] += 1;
    if (best_J) best_J[b] = bj;
    if (best_idx) best_idx[b] = bi;
  }
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
