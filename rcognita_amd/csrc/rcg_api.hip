// rcg_api.hip - C ABI (include/rcg.h) over the gfx950 kernels of rcg_kernels.hpp.
//
// There is no CPU fallback in this library: without a HIP device rcg_create fails with
// RCG_ERR_NO_DEVICE.  The CPU restatement used for parity lives under oracle/ and is never linked
// or called from here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rcg_critic_fit.hpp"
#include "rcg_kernels.hpp"

using namespace rcg;

struct rcg_handle {
  rcg_cfg cfg;
  int ds, du, np, dc, nchi;
  size_t esz;  // sizeof(real)
  hipStream_t stream;
  void* f[RCG_FIELD_COUNT_];
  size_t fbytes[RCG_FIELD_COUNT_];
  double* d_summary;
  long tick_count;  // control ticks issued through rcg_control_tick (drives the critic period)
  void* d_rfull;  // [2][49] real: full R1, R2 (read by the non-diagonal stage cost only)
  KParams<float> p32;
  KParams<double> p64;
  std::string err;
  // measurement (rcg_profile): event pairs recorded on `stream`, drained into totals on demand
  bool prof;
  std::vector<hipEvent_t> ev_free;
  struct Pending {
    hipEvent_t a, b;
    int kernel;
  };
  std::vector<Pending> ev_pending;
  double prof_ms[RCG_KERNEL_COUNT_];
  int64_t prof_n[RCG_KERNEL_COUNT_];
};

// RAII bracket: records start/stop events around the launches made while it is alive
struct ProfScope {
  rcg_handle* h;
  hipEvent_t a, b;
  int kernel;
  bool on;
  ProfScope(rcg_handle* h_, int kernel_) : h(h_), a(nullptr), b(nullptr), kernel(kernel_), on(h_->prof) {
    if (!on) return;
    for (hipEvent_t* e : {&a, &b}) {
      if (!h->ev_free.empty()) {
        *e = h->ev_free.back();
        h->ev_free.pop_back();
      } else if (hipEventCreate(e) != hipSuccess) {
        on = false;
        return;
      }
    }
    (void)hipEventRecord(a, h->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, h->stream);
    h->ev_pending.push_back({a, b, kernel});
  }
};

static void prof_drain(rcg_handle* h) {
  (void)hipStreamSynchronize(h->stream);
  for (auto& p : h->ev_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      h->prof_ms[p.kernel] += ms;
      h->prof_n[p.kernel] += 1;
    }
    h->ev_free.push_back(p.a);
    h->ev_free.push_back(p.b);
  }
  h->ev_pending.clear();
}

static thread_local std::string g_err = "";

static int fail(rcg_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h)
    h->err = buf;
  else
    g_err = buf;
  return code;
}

#define HIPCHK(h, call)                                                                             \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail((h), RCG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                        \
  } while (0)

static const int kDims[3][3] = {{5, 2, 2}, {3, 2, 0}, {2, 1, 5}};  // ds, du, np
// layout of the per-handle constant block in HBM (see rcg_create)
static constexpr size_t kConstR64 = 512, kConstW = 1296, kConstBytes = 2256;
static constexpr int kFitMaxRows = 8;  // Ncritic - 1 <= 8 for the native critic fit

static int dim_critic(int cs, int dy, int du) {
  const int n = dy + du;
  switch (cs) {
    case RCG_CRITIC_QUAD_LIN: return n * (n + 1) / 2 + n;
    case RCG_CRITIC_QUADRATIC: return n * (n + 1) / 2;
    case RCG_CRITIC_QUAD_NOMIX: return n;
    case RCG_CRITIC_QUAD_MIX: return dy + dy * du + du;
  }
  return -1;
}

template <typename real>
static void build_params(const rcg_handle* h, KParams<real>* P, real* rfull_host) {
  const rcg_cfg& c = h->cfg;
  memset(P, 0, sizeof *P);
  memset(rfull_host, 0, 2 * 49 * sizeof(real));
  P->Rfull = (const real*)h->d_rfull;
  const int n = h->nchi;
  for (int i = 0; i < RCG_MAX_PARS; ++i) P->pars[i] = (real)c.pars[i];
  for (int i = 0; i < h->du; ++i) {
    P->lo[i] = (real)c.ctrl_bnds[2 * i];
    P->hi[i] = (real)c.ctrl_bnds[2 * i + 1];
  }
  bool full = false;
  const bool biq = c.stage_obj_struct == RCG_STAGE_BIQUADRATIC;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      rfull_host[i * n + j] = (real)c.R1[i * n + j];
      rfull_host[49 + i * n + j] = biq ? (real)c.R2[i * n + j] : (real)0;
      if (i != j && (c.R1[i * n + j] != 0.0 || (biq && c.R2[i * n + j] != 0.0))) full = true;
    }
  for (int i = 0; i < n; ++i) {
    P->R1d[i] = (real)c.R1[i * n + i];
    P->R2d[i] = biq ? (real)c.R2[i * n + i] : (real)0;
  }
  for (int i = 0; i < h->ds; ++i) P->target[i] = (c.flags & RCG_FLAG_HAS_TARGET) ? (real)c.target[i] : (real)0;
  P->gamma = (real)c.gamma;
  P->h_pred = (real)c.pred_step_size;
  P->dt_sim = (real)c.dt_sim;
  P->sampling_time = (real)c.sampling_time;
  P->B = c.batch;
  P->n_actor = c.n_actor;
  P->mode = c.mode;
  P->critic_struct = c.critic_struct;
  P->dc = h->dc;
  P->n_critic = c.n_critic;
  P->buffer_size = c.buffer_size;
  P->stage_kind = (full ? STAGE_FULL : 0) | (biq ? STAGE_BIQUAD : 0);
  P->has_target = (c.flags & RCG_FLAG_HAS_TARGET) ? 1 : 0;
  P->clip = (c.flags & RCG_FLAG_NO_CLIP) ? 0 : 1;
  P->per_env_pars = (c.flags & RCG_FLAG_PER_ENV_PARS) ? 1 : 0;
  P->ref_lag = (c.flags & RCG_FLAG_REF_LAG) ? 1 : 0;
  P->accum_every_substep = (c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP) ? 1 : 0;
}

template <typename real>
static const KParams<real>& params(const rcg_handle* h);
template <>
const KParams<float>& params<float>(const rcg_handle* h) {
  return h->p32;
}
template <>
const KParams<double>& params<double>(const rcg_handle* h) {
  return h->p64;
}

// f(SysTag{}, real{}) for the handle's system and dtype
template <typename F>
static int dispatch(rcg_handle* h, F&& f) {
  const bool d = h->cfg.dtype == RCG_F64;
  switch (h->cfg.sys_id) {
    case RCG_SYS_3WROBOT: return d ? f(Sys3WRobot{}, double{}) : f(Sys3WRobot{}, float{});
    case RCG_SYS_3WROBOT_NI: return d ? f(Sys3WRobotNI{}, double{}) : f(Sys3WRobotNI{}, float{});
    case RCG_SYS_2TANK: return d ? f(Sys2Tank{}, double{}) : f(Sys2Tank{}, float{});
  }
  return fail(h, RCG_ERR_BAD_ARG, "unknown sys_id %d", h->cfg.sys_id);
}

template <typename real>
__global__ void k_fill(real* p, long n, real v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

static inline unsigned blocks_for(long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

template <typename real>
static int fill_rows(rcg_handle* h, void* base, int rows, const double* vals) {
  const long B = h->cfg.batch;
  for (int r = 0; r < rows; ++r) {
    hipLaunchKernelGGL(k_fill<real>, dim3(blocks_for(B)), dim3(256), 0, h->stream, (real*)base + (long)r * B, B,
                       (real)vals[r]);
  }
  HIPCHK(h, hipGetLastError());
  return RCG_OK;
}

// shared launcher of k_actor
template <typename Sys, typename real>
static int launch_actor(rcg_handle* h, const char* who, const void* cand, int K, const void* obs,
                        const void* state_sys, const void* w, void* J, void* action, void* best_J, int32_t* best_idx,
                        bool tick) {
  constexpr int DU = Sys::DU;
  const rcg_cfg& c = h->cfg;
  if (K < 1) return fail(h, RCG_ERR_BAD_ARG, "%s: K must be >= 1", who);
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
    return fail(h, RCG_ERR_BAD_ARG, "%s: RQL/SQL need critic weights (buffer_size > 0 or an explicit w)", who);
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
  if (!cand) {
    if (DU == 1) {
      A.grid_g = K;
    } else {
      int g = (int)std::floor(std::sqrt((double)K) + 1e-9);
      if (g * g != K) return fail(h, RCG_ERR_BAD_ARG, "%s: generated grid for du = 2 needs a square K (got %d)", who, K);
      A.grid_g = g;
    }
  }
  const int R = c.n_actor * DU;
  const size_t row_bytes = (size_t)R * sizeof(real);
  A.vec_ok = (cand && row_bytes % 16 == 0 && ((uintptr_t)cand % 16) == 0) ? 1 : 0;
  const long B = c.batch;
  const long n_waves = (B + A.G - 1) / A.G;
  int wpb = 4;  // waves per workgroup
  size_t lds_per_wave = cand ? 64 * row_bytes : 0;
  while (wpb > 1 && lds_per_wave * wpb > 64 * 1024) wpb >>= 1;
  const size_t lds = lds_per_wave * wpb;
  const unsigned blocks = (unsigned)((n_waves + wpb - 1) / wpb);
  const bool generic = !(c.mode == RCG_MODE_MPC && params<real>(h).stage_kind == 0);
  const bool tgt = (c.flags & RCG_FLAG_HAS_TARGET) != 0;
  const KParams<real>& P = params<real>(h);
  ProfScope prof_scope(h, RCG_KERNEL_ACTOR);

  // Production shape (f32, MPC + diagonal R1, K a multiple of 64, 16-B granular rows of <= 8 KiB per tile)
  // -> k_actor_dma.  Development knobs, read per launch: RCG_ACTOR_KERNEL=plain forces k_actor,
  // RCG_GPW=<n> sets the envs per persistent wave, RCG_DBG=1 selects the timing-only variant.
  const int nrow = (int)(row_bytes / 16);
  if (const char* e = getenv("RCG_DBG")) A.dbg = atoi(e);
  const char* ksel = getenv("RCG_ACTOR_KERNEL");
  const bool force_plain = ksel && !strcmp(ksel, "plain");
  if constexpr (std::is_same<real, float>::value) {
    if (cand && A.vec_ok && K >= 64 && (K % 64) == 0 && nrow >= 1 && nrow <= 8 && !generic && !force_plain) {
      long gpw = B / (256L * 20);  // about one round of fully resident waves on 256 CUs
      if (const char* e = getenv("RCG_GPW")) gpw = atol(e);
      gpw = gpw < 1 ? 1 : (gpw > 8 ? 8 : gpw);
      A.gpw = (int)gpw;
      const long pw = (B + gpw - 1) / gpw;
      const unsigned pblocks = (unsigned)((pw + wpb - 1) / wpb);
      const bool same = A.obs == A.state_sys;  // tick mode without ref_lag: one state array
#define RCG_LAUNCH_DMA2(TG, NR)                                                                                   \
  do {                                                                                                            \
    if (same)                                                                                                     \
      hipLaunchKernelGGL((k_actor_dma<Sys, TG, NR, true>), dim3(pblocks), dim3(64 * wpb), lds, h->stream, A, P);  \
    else                                                                                                          \
      hipLaunchKernelGGL((k_actor_dma<Sys, TG, NR, false>), dim3(pblocks), dim3(64 * wpb), lds, h->stream, A, P); \
  } while (0)
#define RCG_LAUNCH_DMA(NR)     \
  case NR:                     \
    if (tgt)                   \
      RCG_LAUNCH_DMA2(true, NR);  \
    else                       \
      RCG_LAUNCH_DMA2(false, NR); \
    break;
      switch (nrow) {
        RCG_LAUNCH_DMA(1)
        RCG_LAUNCH_DMA(2)
        RCG_LAUNCH_DMA(3)
        RCG_LAUNCH_DMA(4)
        RCG_LAUNCH_DMA(5)
        RCG_LAUNCH_DMA(6)
        RCG_LAUNCH_DMA(7)
        RCG_LAUNCH_DMA(8)
      }
#undef RCG_LAUNCH_DMA
#undef RCG_LAUNCH_DMA2
      HIPCHK(h, hipGetLastError());
      return RCG_OK;
    }
  }
#define RCG_LAUNCH_ACTOR(GEN, TGT, STR) \
  hipLaunchKernelGGL((k_actor<Sys, real, GEN, TGT, STR>), dim3(blocks), dim3(64 * wpb), lds, h->stream, A, P)
#define RCG_LAUNCH_ACTOR2(GEN, TGT) \
  do {                              \
    if (cand)                       \
      RCG_LAUNCH_ACTOR(GEN, TGT, true);  \
    else                            \
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
  HIPCHK(h, hipGetLastError());
  return RCG_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

int rcg_version(void) { return RCG_VERSION; }

const char* rcg_last_error(const rcg_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int rcg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rcg_create(const rcg_cfg* cfg, rcg_handle** out) {
  if (!cfg || !out) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(rcg_cfg))
    return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: struct_size %d != sizeof(rcg_cfg) %zu (ABI mismatch)",
                cfg->struct_size, sizeof(rcg_cfg));
  if (cfg->sys_id < 0 || cfg->sys_id > 2) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad sys_id %d", cfg->sys_id);
  if (cfg->batch < 1) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: batch must be >= 1");
  if (cfg->dtype != RCG_F32 && cfg->dtype != RCG_F64) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad dtype");
  if (cfg->mode < 0 || cfg->mode > 2) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad mode %d", cfg->mode);
  if (cfg->stage_obj_struct < 0 || cfg->stage_obj_struct > 1)
    return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad stage_obj_struct");
  if (cfg->critic_struct < 0 || cfg->critic_struct > 3) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad critic_struct");
  const int ds = kDims[cfg->sys_id][0], du = kDims[cfg->sys_id][1], np = kDims[cfg->sys_id][2];
  if (cfg->n_actor < 1 || cfg->n_actor * du > RCG_MAX_ROW)
    return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: need 1 <= Nactor and Nactor*du <= %d", RCG_MAX_ROW);
  if (cfg->substeps_per_tick < 1) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: substeps_per_tick must be >= 1");
  if (cfg->buffer_size < 0) return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: buffer_size < 0");
  if (cfg->mode != RCG_MODE_MPC && cfg->buffer_size < 2)
    return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: RQL/SQL need buffer_size >= 2");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, RCG_ERR_NO_DEVICE,
                "rcg_create: no HIP device visible; librcg has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: device %d out of range (%d visible)", cfg->device, ndev);
  HIPCHK(nullptr, hipSetDevice(cfg->device));

  rcg_handle* h = new rcg_handle();
  h->cfg = *cfg;
  h->ds = ds;
  h->du = du;
  h->np = np;
  h->nchi = ds + du;
  h->dc = dim_critic(cfg->critic_struct, ds, du);
  h->esz = cfg->dtype == RCG_F64 ? 8 : 4;
  h->stream = nullptr;
  h->d_summary = nullptr;
  h->prof = false;
  h->tick_count = 0;
  memset(h->prof_ms, 0, sizeof h->prof_ms);
  memset(h->prof_n, 0, sizeof h->prof_n);
  if (h->cfg.buffer_size > 0) {
    // Ncritic = min(Ncritic, buffer_size - 1)  (controllers.py:1015)
    if (h->cfg.n_critic > h->cfg.buffer_size - 1) h->cfg.n_critic = h->cfg.buffer_size - 1;
  }
  bool any_bnd = false;
  for (int i = 0; i < 2 * du; ++i) any_bnd = any_bnd || cfg->ctrl_bnds[i] != 0.0;
  if (!any_bnd) h->cfg.flags |= RCG_FLAG_NO_CLIP;  // `if self.ctrl_bnds.any()` (systems.py:241)
  h->d_rfull = nullptr;
  {
    // one small constant block: [0,392) R1|R2 as f32, [512,1296) R1|R2 as f64, [1296,2256) w_init|w_min|w_max
    if (hipMalloc(&h->d_rfull, kConstBytes) != hipSuccess) {
      fail(nullptr, RCG_ERR_HIP, "rcg_create: allocating the constant block");
      delete h;
      return RCG_ERR_HIP;
    }
    unsigned char blk[kConstBytes];
    memset(blk, 0, sizeof blk);
    build_params<float>(h, &h->p32, reinterpret_cast<float*>(blk));
    build_params<double>(h, &h->p64, reinterpret_cast<double*>(blk + kConstR64));
    h->p64.Rfull = reinterpret_cast<const double*>((unsigned char*)h->d_rfull + kConstR64);
    double* wc = reinterpret_cast<double*>(blk + kConstW);
    for (int i = 0; i < 40; ++i) {
      wc[i] = cfg->w_init[i];
      wc[40 + i] = cfg->w_min[i];
      wc[80 + i] = cfg->w_max[i];
    }
    hipError_t er = hipMemcpy(h->d_rfull, blk, sizeof blk, hipMemcpyHostToDevice);
    if (er != hipSuccess) {
      fail(nullptr, RCG_ERR_HIP, "rcg_create: uploading the stage-cost matrices: %s", hipGetErrorString(er));
      (void)hipFree(h->d_rfull);
      delete h;
      return RCG_ERR_HIP;
    }
  }

  const size_t B = (size_t)cfg->batch, e = h->esz;
  memset(h->f, 0, sizeof h->f);
  memset(h->fbytes, 0, sizeof h->fbytes);
  h->fbytes[RCG_FIELD_STATE] = ds * B * e;
  h->fbytes[RCG_FIELD_ACTION] = du * B * e;
  h->fbytes[RCG_FIELD_ACCUM] = B * e;
  h->fbytes[RCG_FIELD_STEP_IDX] = B * 4;
  h->fbytes[RCG_FIELD_EPISODE_IDX] = B * 4;
  h->fbytes[RCG_FIELD_STATUS] = B * 4;
  h->fbytes[RCG_FIELD_PARS] = ((cfg->flags & RCG_FLAG_PER_ENV_PARS) && np > 0) ? np * B * e : 0;
  h->fbytes[RCG_FIELD_STATE_INIT] = ds * B * e;
  h->fbytes[RCG_FIELD_STATE_PREV] = ds * B * e;
  h->fbytes[RCG_FIELD_BEST_J] = B * e;
  h->fbytes[RCG_FIELD_BEST_IDX] = B * 4;
  h->fbytes[RCG_FIELD_RETURNS] = B * e;
  if (cfg->buffer_size > 0) {
    h->fbytes[RCG_FIELD_W_CRITIC] = h->dc * B * e;
    h->fbytes[RCG_FIELD_W_PREV] = h->dc * B * e;
    h->fbytes[RCG_FIELD_OBS_BUF] = (size_t)cfg->buffer_size * ds * B * e;
    h->fbytes[RCG_FIELD_ACT_BUF] = (size_t)cfg->buffer_size * du * B * e;
  }
  for (int i = 0; i < RCG_FIELD_COUNT_; ++i) {
    if (!h->fbytes[i]) continue;
    hipError_t er = hipMalloc(&h->f[i], h->fbytes[i]);
    if (er == hipSuccess) er = hipMemsetAsync(h->f[i], 0, h->fbytes[i], h->stream);
    if (er != hipSuccess) {
      fail(nullptr, RCG_ERR_HIP, "rcg_create: allocating field %d (%zu bytes): %s", i, h->fbytes[i], hipGetErrorString(er));
      rcg_destroy(h);
      return RCG_ERR_HIP;
    }
  }
  if (hipMalloc((void**)&h->d_summary, 6 * sizeof(double)) != hipSuccess) {
    fail(nullptr, RCG_ERR_HIP, "rcg_create: allocating summary scratch");
    rcg_destroy(h);
    return RCG_ERR_HIP;
  }
  int rc = RCG_OK;
  if (cfg->dtype == RCG_F64) {
    rc = fill_rows<double>(h, h->f[RCG_FIELD_ACTION], du, cfg->action_init);
    if (rc == RCG_OK && cfg->buffer_size > 0) rc = fill_rows<double>(h, h->f[RCG_FIELD_W_CRITIC], h->dc, cfg->w_init);
    if (rc == RCG_OK && cfg->buffer_size > 0) rc = fill_rows<double>(h, h->f[RCG_FIELD_W_PREV], h->dc, cfg->w_init);
  } else {
    rc = fill_rows<float>(h, h->f[RCG_FIELD_ACTION], du, cfg->action_init);
    if (rc == RCG_OK && cfg->buffer_size > 0) rc = fill_rows<float>(h, h->f[RCG_FIELD_W_CRITIC], h->dc, cfg->w_init);
    if (rc == RCG_OK && cfg->buffer_size > 0) rc = fill_rows<float>(h, h->f[RCG_FIELD_W_PREV], h->dc, cfg->w_init);
  }
  if (rc != RCG_OK) {
    g_err = h->err;
    rcg_destroy(h);
    return rc;
  }
  *out = h;
  return RCG_OK;
}

int rcg_destroy(rcg_handle* h) {
  if (!h) return RCG_OK;
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);
  for (int i = 0; i < RCG_FIELD_COUNT_; ++i)
    if (h->f[i]) (void)hipFree(h->f[i]);
  if (h->d_summary) (void)hipFree(h->d_summary);
  if (h->d_rfull) (void)hipFree(h->d_rfull);
  for (auto& p : h->ev_pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : h->ev_free) (void)hipEventDestroy(e);
  delete h;
  return RCG_OK;
}

int rcg_set_stream(rcg_handle* h, void* hip_stream) {
  if (!h) return RCG_ERR_BAD_ARG;
  h->stream = (hipStream_t)hip_stream;
  return RCG_OK;
}

int rcg_synchronize(rcg_handle* h) {
  if (!h) return RCG_ERR_BAD_ARG;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int rcg_dev_alloc(rcg_handle* h, uint64_t bytes, void** dev_out) {
  if (!h || !dev_out) return RCG_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipMalloc(dev_out, bytes ? bytes : 16));
  return RCG_OK;
}

int rcg_dev_free(rcg_handle* h, void* dev) {
  if (!h) return RCG_ERR_BAD_ARG;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipFree(dev));
  return RCG_OK;
}

int rcg_memcpy_h2d(rcg_handle* h, void* dev_dst, const void* host_src, uint64_t bytes) {
  if (!h || !dev_dst || !host_src) return fail(h, RCG_ERR_BAD_ARG, "rcg_memcpy_h2d: null argument");
  HIPCHK(h, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int rcg_memcpy_d2h(rcg_handle* h, void* host_dst, const void* dev_src, uint64_t bytes) {
  if (!h || !host_dst || !dev_src) return fail(h, RCG_ERR_BAD_ARG, "rcg_memcpy_d2h: null argument");
  HIPCHK(h, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

static int check_field(rcg_handle* h, int field, const char* who) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (field < 0 || field >= RCG_FIELD_COUNT_) return fail(h, RCG_ERR_BAD_ARG, "%s: bad field %d", who, field);
  if (!h->f[field]) return fail(h, RCG_ERR_BAD_ARG, "%s: field %d is not allocated for this configuration", who, field);
  return RCG_OK;
}

int rcg_set_field(rcg_handle* h, int field, const void* src, int where) {
  int rc = check_field(h, field, "rcg_set_field");
  if (rc) return rc;
  if (!src) return fail(h, RCG_ERR_BAD_ARG, "rcg_set_field: null src");
  HIPCHK(h, hipMemcpyAsync(h->f[field], src, h->fbytes[field],
                           where == RCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
  if (field == RCG_FIELD_STATE)  // a freshly set state is also its own "previous" state
    HIPCHK(h, hipMemcpyAsync(h->f[RCG_FIELD_STATE_PREV], src, h->fbytes[field],
                             where == RCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
  if (where == RCG_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int rcg_get_field(rcg_handle* h, int field, void* dst, int where) {
  int rc = check_field(h, field, "rcg_get_field");
  if (rc) return rc;
  if (!dst) return fail(h, RCG_ERR_BAD_ARG, "rcg_get_field: null dst");
  HIPCHK(h, hipMemcpyAsync(dst, h->f[field], h->fbytes[field],
                           where == RCG_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
  if (where == RCG_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int64_t rcg_field_bytes(const rcg_handle* h, int field) {
  if (!h || field < 0 || field >= RCG_FIELD_COUNT_) return 0;
  return (int64_t)h->fbytes[field];
}

int rcg_field_ptr(rcg_handle* h, int field, void** dev_out) {
  int rc = check_field(h, field, "rcg_field_ptr");
  if (rc) return rc;
  if (!dev_out) return fail(h, RCG_ERR_BAD_ARG, "rcg_field_ptr: null out");
  *dev_out = h->f[field];
  return RCG_OK;
}

// ---- stateless operators --------------------------------------------------------------------
int rcg_rhs(rcg_handle* h, const void* state, const void* action, void* dstate, void* clipped_action, int32_t n,
            int32_t clip) {
  if (!h || !state || !action || !dstate || n < 1) return fail(h, RCG_ERR_BAD_ARG, "rcg_rhs: bad argument");
  return dispatch(h, [&](auto sys, auto r) {
    using Sys = decltype(sys);
    using real = decltype(r);
    const real* pe = (h->f[RCG_FIELD_PARS] && n == h->cfg.batch) ? (const real*)h->f[RCG_FIELD_PARS] : nullptr;
    hipLaunchKernelGGL((k_rhs<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)state,
                       (const real*)action, (real*)dstate, (real*)clipped_action, pe, (long)n, (int)clip,
                       params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

int rcg_stage_obj(rcg_handle* h, const void* obs, const void* act, void* out, int32_t n) {
  if (!h || !obs || !act || !out || n < 1) return fail(h, RCG_ERR_BAD_ARG, "rcg_stage_obj: bad argument");
  return dispatch(h, [&](auto sys, auto r) {
    using Sys = decltype(sys);
    using real = decltype(r);
    hipLaunchKernelGGL((k_stage_obj<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)obs,
                       (const real*)act, (real*)out, (long)n, params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

int rcg_critic(rcg_handle* h, const void* obs, const void* act, const void* w, void* out, int32_t n) {
  if (!h || !obs || !act || !w || !out || n < 1) return fail(h, RCG_ERR_BAD_ARG, "rcg_critic: bad argument");
  return dispatch(h, [&](auto sys, auto r) {
    using Sys = decltype(sys);
    using real = decltype(r);
    hipLaunchKernelGGL((k_critic<Sys, real>), dim3(blocks_for(n)), dim3(256), 0, h->stream, (const real*)obs,
                       (const real*)act, (const real*)w, (real*)out, (long)n, params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

int rcg_actor_cost(rcg_handle* h, const void* cand, int32_t K, const void* obs, const void* state_sys, const void* w,
                   void* J) {
  if (!h || !cand || !J) return fail(h, RCG_ERR_BAD_ARG, "rcg_actor_cost: cand and J are required");
  return dispatch(h, [&](auto sys, auto r) {
    return launch_actor<decltype(sys), decltype(r)>(h, "rcg_actor_cost", cand, K, obs, state_sys, w, J, nullptr,
                                                    nullptr, nullptr, false);
  });
}

int rcg_actor_argmin(rcg_handle* h, const void* cand, int32_t K, const void* obs, const void* state_sys, void* action,
                     void* best_J, int32_t* best_idx) {
  if (!h) return RCG_ERR_BAD_ARG;
  return dispatch(h, [&](auto sys, auto r) {
    return launch_actor<decltype(sys), decltype(r)>(h, "rcg_actor_argmin", cand, K, obs, state_sys, nullptr, nullptr,
                                                    action, best_J, best_idx, false);
  });
}

int rcg_critic_cost(rcg_handle* h, const void* w, void* Jc) {
  if (!h || !Jc) return fail(h, RCG_ERR_BAD_ARG, "rcg_critic_cost: Jc is required");
  if (!h->f[RCG_FIELD_OBS_BUF]) return fail(h, RCG_ERR_BAD_ARG, "rcg_critic_cost: handle has no critic buffers (buffer_size = 0)");
  return dispatch(h, [&](auto sys, auto r) {
    using Sys = decltype(sys);
    using real = decltype(r);
    hipLaunchKernelGGL((k_critic_cost<Sys, real>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, h->stream,
                       w ? (const real*)w : (const real*)h->f[RCG_FIELD_W_CRITIC], (const real*)h->f[RCG_FIELD_W_PREV],
                       (const real*)h->f[RCG_FIELD_OBS_BUF], (const real*)h->f[RCG_FIELD_ACT_BUF], (real*)Jc,
                       params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

// ---- stateful steps -------------------------------------------------------------------------
int rcg_sim_step(rcg_handle* h, int32_t n_substeps) {
  if (!h || n_substeps < 1) return fail(h, RCG_ERR_BAD_ARG, "rcg_sim_step: n_substeps must be >= 1");
  return dispatch(h, [&](auto sys, auto r) {
    using Sys = decltype(sys);
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
    if (h->cfg.flags & RCG_FLAG_HAS_TARGET)
      hipLaunchKernelGGL((k_sim<Sys, real, true>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, h->stream, A,
                         params<real>(h));
    else
      hipLaunchKernelGGL((k_sim<Sys, real, false>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, h->stream, A,
                         params<real>(h));
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

int rcg_critic_update(rcg_handle* h, int32_t do_fit) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (!h->f[RCG_FIELD_OBS_BUF]) return fail(h, RCG_ERR_BAD_ARG, "rcg_critic_update: handle has no critic buffers (buffer_size = 0)");
  const int m = h->cfg.n_critic - 1;
  if (do_fit && (m < 1 || m > kFitMaxRows))
    return fail(h, RCG_ERR_UNSUPPORTED, "rcg_critic_update: the native critic fit needs 1 <= Ncritic-1 <= %d (got %d)",
                kFitMaxRows, m);
  return dispatch(h, [&](auto sys, auto r) {
    using Sys = decltype(sys);
    using real = decltype(r);
    {
      ProfScope prof_scope(h, RCG_KERNEL_CRITIC);
      hipLaunchKernelGGL((k_critic_push<Sys, real>), dim3(blocks_for(h->cfg.batch)), dim3(256), 0, h->stream,
                         (real*)h->f[RCG_FIELD_OBS_BUF], (real*)h->f[RCG_FIELD_ACT_BUF],
                         (const real*)h->f[RCG_FIELD_STATE], (const real*)h->f[RCG_FIELD_ACTION], params<real>(h));
      if (do_fit) {
        FitArgs<real> F;
        F.w_critic = (real*)h->f[RCG_FIELD_W_CRITIC];
        F.w_prev = (real*)h->f[RCG_FIELD_W_PREV];
        F.obs_buf = (const real*)h->f[RCG_FIELD_OBS_BUF];
        F.act_buf = (const real*)h->f[RCG_FIELD_ACT_BUF];
        F.wcfg = reinterpret_cast<const double*>((unsigned char*)h->d_rfull + kConstW);
        const dim3 grid(blocks_for(h->cfg.batch, 64)), block(64);
#define RCG_FIT(CS)                                                                                         \
  do {                                                                                                      \
    if (m <= 3)                                                                                             \
      hipLaunchKernelGGL((k_critic_fit<Sys, real, CS, 3>), grid, block, 0, h->stream, F, h->p64);           \
    else                                                                                                    \
      hipLaunchKernelGGL((k_critic_fit<Sys, real, CS, kFitMaxRows>), grid, block, 0, h->stream, F, h->p64); \
  } while (0)
        switch (h->cfg.critic_struct) {
          case RCG_CRITIC_QUAD_LIN: RCG_FIT(RCG_CRITIC_QUAD_LIN); break;
          case RCG_CRITIC_QUADRATIC: RCG_FIT(RCG_CRITIC_QUADRATIC); break;
          case RCG_CRITIC_QUAD_NOMIX: RCG_FIT(RCG_CRITIC_QUAD_NOMIX); break;
          default: RCG_FIT(RCG_CRITIC_QUAD_MIX); break;
        }
#undef RCG_FIT
      }
    }
    HIPCHK(h, hipGetLastError());
    return (int)RCG_OK;
  });
}

int rcg_control_tick(rcg_handle* h, const void* cand, int32_t K) {
  if (!h) return RCG_ERR_BAD_ARG;
  int rc = rcg_sim_step(h, h->cfg.substeps_per_tick);
  if (rc) return rc;
  if (h->cfg.mode != RCG_MODE_MPC) {
    // critic_period = critic_every_ticks * sampling_time (controllers.py:1466-1477)
    const int every = h->cfg.critic_every_ticks > 1 ? h->cfg.critic_every_ticks : 1;
    rc = rcg_critic_update(h, (h->tick_count % every) == 0 ? 1 : 0);
    if (rc) return rc;
  }
  h->tick_count += 1;
  return dispatch(h, [&](auto sys, auto r) {
    return launch_actor<decltype(sys), decltype(r)>(h, "rcg_control_tick", cand, K, nullptr, nullptr, nullptr, nullptr,
                                                    h->f[RCG_FIELD_ACTION], h->f[RCG_FIELD_BEST_J],
                                                    (int32_t*)h->f[RCG_FIELD_BEST_IDX], true);
  });
}

int rcg_episode_reset(rcg_handle* h) {
  if (!h) return RCG_ERR_BAD_ARG;
  const long B = h->cfg.batch;
  if (h->cfg.dtype == RCG_F64)
    hipLaunchKernelGGL((k_episode_reset<double>), dim3(blocks_for(B)), dim3(256), 0, h->stream,
                       (double*)h->f[RCG_FIELD_STATE], (double*)h->f[RCG_FIELD_STATE_PREV],
                       (const double*)h->f[RCG_FIELD_STATE_INIT], (double*)h->f[RCG_FIELD_ACTION],
                       (double*)h->f[RCG_FIELD_ACCUM], (double*)h->f[RCG_FIELD_RETURNS],
                       (int32_t*)h->f[RCG_FIELD_STEP_IDX], (int32_t*)h->f[RCG_FIELD_EPISODE_IDX],
                       (uint32_t*)h->f[RCG_FIELD_STATUS], h->ds, h->du, h->cfg.action_init[0], h->cfg.action_init[1], B);
  else
    hipLaunchKernelGGL((k_episode_reset<float>), dim3(blocks_for(B)), dim3(256), 0, h->stream,
                       (float*)h->f[RCG_FIELD_STATE], (float*)h->f[RCG_FIELD_STATE_PREV],
                       (const float*)h->f[RCG_FIELD_STATE_INIT], (float*)h->f[RCG_FIELD_ACTION],
                       (float*)h->f[RCG_FIELD_ACCUM], (float*)h->f[RCG_FIELD_RETURNS],
                       (int32_t*)h->f[RCG_FIELD_STEP_IDX], (int32_t*)h->f[RCG_FIELD_EPISODE_IDX],
                       (uint32_t*)h->f[RCG_FIELD_STATUS], h->ds, h->du, (float)h->cfg.action_init[0],
                       (float)h->cfg.action_init[1], B);
  HIPCHK(h, hipGetLastError());
  return RCG_OK;
}

int rcg_episode_stats(rcg_handle* h, int32_t from_accum, void* returns_out, rcg_summary* out) {
  if (!h || !out) return fail(h, RCG_ERR_BAD_ARG, "rcg_episode_stats: out is required");
  const int field = from_accum ? RCG_FIELD_ACCUM : RCG_FIELD_RETURNS;
  const long B = h->cfg.batch;
  if (h->cfg.dtype == RCG_F64)
    hipLaunchKernelGGL((k_stats<double>), dim3(1), dim3(1024), 0, h->stream, (const double*)h->f[field],
                       (const uint32_t*)h->f[RCG_FIELD_STATUS], B, h->d_summary);
  else
    hipLaunchKernelGGL((k_stats<float>), dim3(1), dim3(1024), 0, h->stream, (const float*)h->f[field],
                       (const uint32_t*)h->f[RCG_FIELD_STATUS], B, h->d_summary);
  HIPCHK(h, hipGetLastError());
  double s[6];
  HIPCHK(h, hipMemcpyAsync(s, h->d_summary, sizeof s, hipMemcpyDeviceToHost, h->stream));
  if (returns_out)
    HIPCHK(h, hipMemcpyAsync(returns_out, h->f[field], h->fbytes[field], hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out->count = s[0];
  out->sum = s[1];
  out->sumsq = s[2];
  out->min = s[3];
  out->max = s[4];
  out->n_failed = s[5];
  return s[5] > 0 ? fail(h, RCG_ERR_NONFINITE, "rcg_episode_stats: %.0f env(s) hit a non-finite state", s[5]) : RCG_OK;
}

int rcg_profile(rcg_handle* h, int32_t enable) {
  if (!h) return RCG_ERR_BAD_ARG;
  prof_drain(h);
  h->prof = enable != 0;
  if (enable) {
    memset(h->prof_ms, 0, sizeof h->prof_ms);
    memset(h->prof_n, 0, sizeof h->prof_n);
  }
  return RCG_OK;
}

int rcg_profile_read(rcg_handle* h, int32_t kernel, double* total_ms, int64_t* launches) {
  if (!h || kernel < 0 || kernel >= RCG_KERNEL_COUNT_) return fail(h, RCG_ERR_BAD_ARG, "rcg_profile_read: bad kernel id");
  prof_drain(h);
  if (total_ms) *total_ms = h->prof_ms[kernel];
  if (launches) *launches = h->prof_n[kernel];
  return RCG_OK;
}

}  // extern "C"
