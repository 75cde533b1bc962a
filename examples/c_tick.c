/* c_tick.c - a plain-C client of the drop-in boundary (include/rcg.h, librcg.so): no Python, no torch.
 *
 *   gcc -O2 -Iinclude examples/c_tick.c -Lrcognita_amd/lib -lrcg -Wl,-rpath,$PWD/rcognita_amd/lib -lm -o c_tick
 *   ./c_tick [B] [ticks]
 *
 * Runs the Sys3WRobot preset (presets/main_3wrobot.py:45-47, 177, 207-215 of the reference) for `B` envs: every tick
 * is Simulator.sim_step + CtrlOptPred.compute_action (MPC, Nactor = 10, K = 256 generated candidate sequences) +
 * upd_accum_obj, i.e. one rcg_control_tick.  Prints one line a test can parse. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcg.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != RCG_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rcg_last_error(h));                  \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int ticks = argc > 2 ? atoi(argv[2]) : 50;
  rcg_handle* h = NULL;
  rcg_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = (int32_t)sizeof cfg;
  cfg.sys_id = RCG_SYS_3WROBOT;
  cfg.batch = B;
  cfg.dtype = RCG_F32;
  cfg.n_actor = 10;
  cfg.mode = RCG_MODE_MPC;
  cfg.substeps_per_tick = 1;
  cfg.critic_every_ticks = 1;
  cfg.dt_sim = 0.01;
  cfg.sampling_time = 0.01;
  cfg.pred_step_size = 0.02;
  cfg.gamma = 1.0;
  cfg.pars[0] = 10.0; /* m */
  cfg.pars[1] = 1.0;  /* I */
  cfg.ctrl_bnds[0] = -300.0, cfg.ctrl_bnds[1] = 300.0, cfg.ctrl_bnds[2] = -100.0, cfg.ctrl_bnds[3] = 100.0;
  {
    const double r1[7] = {1, 10, 1, 0, 0, 0, 0}; /* R1_diag of the preset */
    for (int i = 0; i < 7; ++i) cfg.R1[i * 7 + i] = r1[i];
  }
  cfg.action_init[0] = -30.0, cfg.action_init[1] = -10.0; /* action_min / 10 (controllers.py:973-975) */

  if (rcg_device_count() < 1) {
    fprintf(stderr, "no HIP device: librcg has no CPU fallback\n");
    return 2;
  }
  CHECK(rcg_create(&cfg, &h));

  /* initial states, struct-of-arrays [ds][B]: a ring of robots around the origin, heading tangentially */
  float* x0 = (float*)malloc(sizeof(float) * 5 * (size_t)B);
  for (int b = 0; b < B; ++b) {
    const double a = 6.283185307179586 * b / B, r = 3.0 + 5.0 * (b % 7) / 7.0;
    x0[0 * (size_t)B + b] = (float)(r * cos(a));
    x0[1 * (size_t)B + b] = (float)(r * sin(a));
    x0[2 * (size_t)B + b] = (float)(a + 1.5707963267948966);
    x0[3 * (size_t)B + b] = 0.0f;
    x0[4 * (size_t)B + b] = 0.0f;
  }
  CHECK(rcg_set_field(h, RCG_FIELD_STATE, x0, RCG_HOST));
  CHECK(rcg_set_field(h, RCG_FIELD_STATE_INIT, x0, RCG_HOST));

  for (int t = 0; t < ticks; ++t) CHECK(rcg_control_tick(h, NULL, 256)); /* NULL: generated 16 x 16 level grid */

  float* x1 = (float*)malloc(sizeof(float) * 5 * (size_t)B);
  int32_t* steps = (int32_t*)malloc(sizeof(int32_t) * (size_t)B);
  CHECK(rcg_get_field(h, RCG_FIELD_STATE, x1, RCG_HOST));
  CHECK(rcg_get_field(h, RCG_FIELD_STEP_IDX, steps, RCG_HOST));
  rcg_summary s;
  CHECK(rcg_episode_stats(h, 1 /* from the running ACCUM */, NULL, &s));
  double d0 = 0, d1 = 0;
  int steps_ok = 1;
  for (int b = 0; b < B; ++b) {
    d0 += hypot(x0[b], x0[(size_t)B + b]);
    d1 += hypot(x1[b], x1[(size_t)B + b]);
    steps_ok &= steps[b] == ticks;
  }
  printf("c_tick B=%d ticks=%d mean_dist0=%.6f mean_dist1=%.6f accum_sum=%.9e count=%.0f n_failed=%.0f steps_ok=%d\n", B,
         ticks, d0 / B, d1 / B, s.sum, s.count, s.n_failed, steps_ok);
  CHECK(rcg_destroy(h));
  free(x0);
  free(x1);
  free(steps);
  return 0;
}
