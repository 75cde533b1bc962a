/* asan_driver.c - CPU-only sanitizer run (`make asan`: clang -fsanitize=address,undefined) of
 *   (1) the C oracle (oracle/oracle.c, included below) on exactly-sized heap buffers: every system x mode x critic
 *       structure x stage structure, operator and tick entry points, shared and per-env parameters;
 *   (2) the host side of librcg's C ABI as far as it goes without a GPU: argument validation of rcg_create, the
 *       no-device path, every entry point with a NULL handle, error strings.
 * TEST INFRASTRUCTURE: built into build/asan/abi_asan and run by tests/test_asan.py (never on the GPU box's device:
 * GPU AddressSanitizer is not available on this pool).  Exit code 0 = no finding; ASan/UBSan abort otherwise. */
#include <stdio.h>
#include <stdlib.h>

#include "../oracle/oracle.c"
#include "rcg.h"

static int failures = 0;
#define EXPECT(cond)                                              \
  do {                                                            \
    if (!(cond)) {                                                \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                 \
    }                                                             \
  } while (0)

static double* rnd(size_t n, double lo, double hi, unsigned* seed) {
  double* p = (double*)malloc((n ? n : 1) * sizeof(double));
  for (size_t i = 0; i < n; ++i) {
    *seed = *seed * 1664525u + 1013904223u;
    p[i] = lo + (hi - lo) * ((*seed >> 8) / 16777216.0);
  }
  return p;
}

static int dim_critic_of(int cs, int ds, int du) {
  const int n = ds + du;
  return cs == 0 ? n * (n + 1) / 2 + n : (cs == 1 ? n * (n + 1) / 2 : (cs == 2 ? n : ds + ds * du + du));
}

static void oracle_sweep(void) {
  unsigned seed = 12345u;
  const int B = 3, K = 5, N = 4;
  for (int sys = 0; sys < 3; ++sys)
    for (int mode = 0; mode < 3; ++mode)
      for (int cs = 0; cs < 4; ++cs)
        for (int biq = 0; biq < 2; ++biq)
          for (int pe = 0; pe < 2; ++pe) {
            orc_cfg c;
            memset(&c, 0, sizeof c);
            const int ds = DIMS[sys][0], du = DIMS[sys][1], n = ds + du, dc = dim_critic_of(cs, ds, du);
            c.sys_id = sys, c.n_actor = N, c.mode = mode, c.biquad = biq, c.critic_struct = cs;
            c.has_target = sys == 2, c.clip = 1, c.substeps_per_tick = 2;
            c.gamma = 0.95, c.h_pred = 0.02, c.dt_sim = 0.01, c.sampling_time = 0.01;
            const double pars[8] = {10.0, 1.0, 1.3, 1.0, 0.2, 0, 0, 0};
            memcpy(c.pars, pars, sizeof pars);
            if (sys == 2) c.pars[0] = 18.4, c.pars[1] = 24.4;
            for (int k = 0; k < du; ++k) c.lo[k] = -1.0 - k, c.hi[k] = 1.0 + k;
            for (int i = 0; i < n; ++i)
              for (int j = 0; j < n; ++j) {
                c.R1[i * n + j] = i == j ? 1.0 + i : 0.01 * (i + j);
                c.R2[i * n + j] = i == j ? 0.5 : 0.0;
              }
            c.target[0] = c.target[1] = 0.5;
            double* cand = rnd((size_t)B * K * N * du, -2, 2, &seed);
            double* state = rnd((size_t)B * ds, -1, 1, &seed);
            double* action = rnd((size_t)B * du, -3, 3, &seed);
            double* w = rnd((size_t)B * dc, 0, 2, &seed);
            double* pv = rnd((size_t)B * 8, 1, 20, &seed);
            double* accum = (double*)calloc(B, sizeof(double));
            int32_t* step = (int32_t*)calloc(B, sizeof(int32_t));
            double* J = (double*)malloc((size_t)B * K * sizeof(double));
            double* bj = (double*)malloc(B * sizeof(double));
            int32_t* bi = (int32_t*)malloc(B * sizeof(int32_t));
            orc_actor_cost_batch(&c, B, K, cand, state, state, pe ? pv : c.pars, pe, w, dc, J, 2);
            for (int i = 0; i < B * K; ++i) EXPECT(J[i] == J[i]);
            for (int t = 0; t < 2; ++t)
              orc_control_tick(&c, B, K, cand, state, action, accum, step, pe ? pv : c.pars, pe, w, dc, bj, bi, 2);
            for (int b = 0; b < B; ++b) EXPECT(step[b] == 2 && bi[b] >= 0 && bi[b] < K && accum[b] == accum[b]);
            free(cand), free(state), free(action), free(w), free(pv), free(accum), free(step), free(J), free(bj), free(bi);
          }
  EXPECT(orc_max_threads() >= 1);
}

static rcg_cfg good_cfg(void) {
  rcg_cfg c;
  memset(&c, 0, sizeof c);
  c.struct_size = (int32_t)sizeof(rcg_cfg);
  c.sys_id = RCG_SYS_3WROBOT, c.batch = 4, c.dtype = RCG_F32, c.device = 0, c.n_actor = 5, c.mode = RCG_MODE_MPC;
  c.substeps_per_tick = 1, c.dt_sim = c.sampling_time = 0.01, c.pred_step_size = 0.02, c.gamma = 1.0;
  c.pars[0] = 10, c.pars[1] = 1;
  return c;
}

static void abi_host_side(void) {
  rcg_handle* h = (rcg_handle*)0x1;
  EXPECT(rcg_version() == RCG_VERSION);
  EXPECT(rcg_last_error(NULL) != NULL);
  EXPECT(rcg_create(NULL, &h) == RCG_ERR_BAD_ARG);
  rcg_cfg c = good_cfg();
  EXPECT(rcg_create(&c, NULL) == RCG_ERR_BAD_ARG);
#define BAD(stmt, needle)                                  \
  do {                                                     \
    c = good_cfg();                                        \
    stmt;                                                  \
    h = (rcg_handle*)0x1;                                  \
    EXPECT(rcg_create(&c, &h) == RCG_ERR_BAD_ARG);         \
    EXPECT(h == NULL);                                     \
    EXPECT(strstr(rcg_last_error(NULL), needle) != NULL);  \
  } while (0)
  BAD(c.struct_size = 12, "struct_size");
  BAD(c.sys_id = 3, "sys_id");
  BAD(c.sys_id = -1, "sys_id");
  BAD(c.batch = 0, "batch");
  BAD(c.dtype = 2, "dtype");
  BAD(c.mode = 7, "mode");
  BAD(c.stage_obj_struct = 2, "stage_obj_struct");
  BAD(c.critic_struct = -1, "critic_struct");
  BAD(c.n_actor = 0, "Nactor");
  BAD(c.n_actor = RCG_MAX_NACTOR + 1, "Nactor"); /* (round 6: any horizon up to the sanity bound; 1000 is a legal Nactor) */
  BAD(c.substeps_per_tick = 0, "substeps_per_tick");
  BAD(c.buffer_size = -1, "buffer_size");
  BAD((c.mode = RCG_MODE_RQL, c.buffer_size = 1), "buffer_size");
#undef BAD
  const int ndev = rcg_device_count();
  EXPECT(ndev >= 0);
  c = good_cfg();
  if (ndev == 0) { /* the build container: the product has no CPU fallback and must say so */
    h = (rcg_handle*)0x1;
    EXPECT(rcg_create(&c, &h) == RCG_ERR_NO_DEVICE);
    EXPECT(h == NULL);
    EXPECT(strstr(rcg_last_error(NULL), "no CPU fallback") != NULL);
  } else {
    c.device = ndev; /* out of range: refused before anything is allocated */
    EXPECT(rcg_create(&c, &h) == RCG_ERR_BAD_ARG);
  }
  /* every entry point with a NULL handle: an error code, never a crash */
  double x[8] = {0};
  int32_t i32[2] = {0};
  void* p = NULL;
  rcg_summary s;
  double ms;
  int64_t nl;
  EXPECT(rcg_destroy(NULL) == RCG_OK);
  EXPECT(rcg_set_stream(NULL, NULL) < 0);
  EXPECT(rcg_synchronize(NULL) < 0);
  EXPECT(rcg_use_own_stream(NULL) < 0);
  EXPECT(rcg_dev_alloc(NULL, 16, &p) < 0);
  EXPECT(rcg_dev_free(NULL, NULL) < 0);
  EXPECT(rcg_memcpy_h2d(NULL, x, x, 8) < 0);
  EXPECT(rcg_memcpy_d2h(NULL, x, x, 8) < 0);
  EXPECT(rcg_set_field(NULL, 0, x, RCG_HOST) < 0);
  EXPECT(rcg_get_field(NULL, 0, x, RCG_HOST) < 0);
  EXPECT(rcg_field_bytes(NULL, 0) == 0);
  EXPECT(rcg_field_ptr(NULL, 0, &p) < 0);
  EXPECT(rcg_rhs(NULL, x, x, x, x, 1, 0) < 0);
  EXPECT(rcg_rhs_full(NULL, x, x, x, x, x, x, x, 1, 0) < 0);
  EXPECT(rcg_disturb_noise(NULL, x, x) < 0);
  EXPECT(rcg_stage_obj(NULL, x, x, x, 1) < 0);
  EXPECT(rcg_critic(NULL, x, x, x, x, 1) < 0);
  EXPECT(rcg_actor_cost(NULL, x, 1, x, x, x, x) < 0);
  EXPECT(rcg_critic_cost(NULL, x, x) < 0);
  EXPECT(rcg_sim_step(NULL, 1) < 0);
  EXPECT(rcg_actor_argmin(NULL, x, 1, x, x, x, x, i32) < 0);
  EXPECT(rcg_control_tick(NULL, x, 1) < 0);
  EXPECT(rcg_control_ticks(NULL, 2, 16) < 0);
  EXPECT(rcg_control_tick_n(NULL, x, 1, 2) < 0);
  EXPECT(rcg_actor_optimize(NULL, 1, x, x, x, x, x, x, i32) < 0);
  EXPECT(rcg_control_tick_opt(NULL, 1, 0) < 0);
  EXPECT(rcg_nominal_action(NULL, x, x, x, 1, 1.0, NULL, 0) < 0);
  EXPECT(rcg_control_tick_nominal(NULL, 1.0, NULL) < 0);
  EXPECT(rcg_critic_update(NULL, 1) < 0);
  EXPECT(rcg_loop_step(NULL, x, 0.01, 1, RCG_LOOP_DECIDE, 3, x) < 0);
  EXPECT(rcg_loop_step_begin(NULL, x, 0.01, 1, RCG_LOOP_DECIDE, 3) < 0);
  EXPECT(rcg_loop_step_end(NULL, x) < 0);
  EXPECT(rcg_set_optimizer_tol(NULL, 1e-7) < 0);
  EXPECT(rcg_sim_step_h(NULL, 1, 0.01) < 0);
  EXPECT(rcg_set_tick_parts(NULL, 2) < 0);
  EXPECT(rcg_join(NULL) < 0);
  EXPECT(rcg_episode_reset(NULL) < 0);
  EXPECT(rcg_episode_stats(NULL, 0, NULL, &s) < 0);
  EXPECT(rcg_profile(NULL, 1) < 0);
  EXPECT(rcg_profile_read(NULL, 0, &ms, &nl) < 0);
  EXPECT(rcg_tick_count(NULL) < 0);
  EXPECT(rcg_set_tick_count(NULL, 3) < 0);
  EXPECT(rcg_last_error(NULL)[0] != '\0');
}

int main(void) {
  oracle_sweep();
  abi_host_side();
  if (failures) {
    fprintf(stderr, "asan_driver: %d expectation(s) failed\n", failures);
    return 1;
  }
  printf("asan_driver ok: oracle sweep + C ABI host side, no sanitizer finding\n");
  return 0;
}
