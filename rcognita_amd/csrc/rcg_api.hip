// rcg_api.hip - C ABI (include/rcg.h) of librcg.so: handle life cycle, per-env tensors, dispatch to the
// per-system launchers (rcg_sys_inst.hip via SysVTable) and the system-independent kernels.
//
// There is no CPU fallback in this library: without a HIP device rcg_create fails with RCG_ERR_NO_DEVICE.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rcg_disturb.hpp"
#include "rcg_handle.hpp"
#include "rcg_search.hpp"

using namespace rcg;

static thread_local std::string g_err = "";

int rcg_fail(rcg_handle* h, int code, const char* fmt, ...) {
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

// Turn the pending event pairs into totals and samples.  Returns RCG_OK, or RCG_ERR_HIP (text in the handle) when a
// pair could not be read - e.g. one that no dispatch ever recorded: the readers pass that on instead of returning
// totals that silently miss launches.
static int prof_drain(rcg_handle* h) {
  if (h->ev_pending.empty()) return RCG_OK;  // nothing recorded: do not stall the stream (rcg_profile just before a timed region)
  (void)hipStreamSynchronize(h->stream);
  int rc = RCG_OK;
  for (auto& p : h->ev_pending) {
    float ms = 0.f;
    const hipError_t er = hipEventElapsedTime(&ms, p.a, p.b);
    if (er != hipSuccess) rc = rcg_fail(h, RCG_ERR_HIP, "rcg_profile: hipEventElapsedTime: %s", hipGetErrorString(er));
    if (er == hipSuccess) {
      h->prof_ms[p.kernel] += ms;
      h->prof_n[p.kernel] += 1;
      if (h->prof_samples[p.kernel].size() < kProfMaxSamples) h->prof_samples[p.kernel].push_back(ms);
    }
    h->ev_free.push_back(p.a);
    h->ev_free.push_back(p.b);
  }
  h->ev_pending.clear();
  return rc;
}

// Every entry point that touches HIP runs on the handle's device whatever the calling thread's current device is
// (two handles on two GPUs in one process, or a handle driven from another thread), and puts the previous one back.
// A split tick's halves (rcg_handle.hpp: split_stream) rejoin the handle's own stream: everything queued on the two internal
// streams so far is ordered before whatever the handle's stream is given next.  Non-blocking for the host.
static void join_split(rcg_handle* h) {
  if (!h || !h->split_pending) return;
  for (int p = 0; p < 2; ++p) {
    (void)hipEventRecord(h->split_join[p], h->split_stream[p]);
    (void)hipStreamWaitEvent(h->stream, h->split_join[p], 0);
  }
  h->split_pending = false;
}

struct NoJoin {};
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const rcg_handle* h) {  // every entry point but the tick itself: the halves of a split tick rejoin first
    if (h) {
      enter(h->cfg.device);
      join_split(const_cast<rcg_handle*>(h));
    }
  }
  DeviceGuard(const rcg_handle* h, NoJoin) {
    if (h) enter(h->cfg.device);
  }
  explicit DeviceGuard(int device) { enter(device); }
  void enter(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

static const int kDims[3][3] = {{5, 2, 2}, {3, 2, 0}, {2, 1, 5}};  // ds, du, np

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
static void build_params(const rcg_handle* h, KParams<real>* P, real* rfull_host, const real* rfull_dev) {
  const rcg_cfg& c = h->cfg;
  memset(P, 0, sizeof *P);
  memset(rfull_host, 0, 2 * 49 * sizeof(real));
  P->Rfull = rfull_dev;
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
  P->zero_w = 0u;
  if (!full && !biq)
    for (int i = 0; i < n; ++i)
      if (c.R1[i * n + i] == 0.0) P->zero_w |= 1u << i;
  P->has_target = (c.flags & RCG_FLAG_HAS_TARGET) ? 1 : 0;
  P->clip = (c.flags & RCG_FLAG_NO_CLIP) ? 0 : 1;
  P->per_env_pars = (c.flags & RCG_FLAG_PER_ENV_PARS) ? 1 : 0;
  P->ref_lag = (c.flags & RCG_FLAG_REF_LAG) ? 1 : 0;
  P->accum_every_substep = (c.flags & RCG_FLAG_ACCUM_EVERY_SUBSTEP) ? 1 : 0;
}

template <typename real>
__global__ void k_fill(real* p, long n, real v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

template <typename real>
static int fill_rows(rcg_handle* h, void* base, int rows, const double* vals) {
  const long B = h->cfg.batch;
  for (int r = 0; r < rows; ++r)
    hipLaunchKernelGGL(k_fill<real>, dim3(blocks_for(B)), dim3(256), 0, h->stream, (real*)base + (long)r * B, B,
                       (real)vals[r]);
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
  if (!cfg || !out) return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(rcg_cfg))
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: struct_size %d != sizeof(rcg_cfg) %zu (ABI mismatch)",
                    cfg->struct_size, sizeof(rcg_cfg));
  if (cfg->sys_id < 0 || cfg->sys_id > 2) return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad sys_id %d", cfg->sys_id);
  if (cfg->batch < 1) return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: batch must be >= 1");
  if (cfg->dtype != RCG_F32 && cfg->dtype != RCG_F64) return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad dtype");
  if (cfg->mode < 0 || cfg->mode > 2) return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad mode %d", cfg->mode);
  if (cfg->stage_obj_struct < 0 || cfg->stage_obj_struct > 1)
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad stage_obj_struct");
  if (cfg->critic_struct < 0 || cfg->critic_struct > 3)
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: bad critic_struct");
  const int ds = kDims[cfg->sys_id][0], du = kDims[cfg->sys_id][1], np = kDims[cfg->sys_id][2];
  // (the reference's horizon is unbounded, controllers.py:965.  Rows of up to RCG_MAX_ROW reals are staged in LDS tiles; longer
  // ones are walked straight from HBM by the generic decision kernel, and the optimiser / search keep their per-wave LDS
  // budget: they refuse - before touching anything - a horizon their working set does not fit, rcg.h)
  if (cfg->n_actor < 1 || cfg->n_actor > RCG_MAX_NACTOR)
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: need 1 <= Nactor <= %d", RCG_MAX_NACTOR);
  if (cfg->substeps_per_tick < 1)
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: substeps_per_tick must be >= 1");
  if (cfg->buffer_size < 0) return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: buffer_size < 0");
  if (cfg->mode != RCG_MODE_MPC && cfg->buffer_size < 2)
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: RQL/SQL need buffer_size >= 2");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return rcg_fail(nullptr, RCG_ERR_NO_DEVICE, "rcg_create: no HIP device visible; librcg has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev)
    return rcg_fail(nullptr, RCG_ERR_BAD_ARG, "rcg_create: device %d out of range (%d visible)", cfg->device, ndev);
  DeviceGuard dev_guard(cfg->device);  // the caller's current device is put back on return
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device)
      return rcg_fail(nullptr, RCG_ERR_HIP, "rcg_create: cannot make device %d current", cfg->device);
  }

  rcg_handle* h = new rcg_handle();
  h->cfg = *cfg;
  h->ds = ds;
  h->du = du;
  h->np = np;
  h->nchi = ds + du;
  h->dc = dim_critic(cfg->critic_struct, ds, du);
  h->esz = cfg->dtype == RCG_F64 ? 8 : 4;
  h->stream = nullptr;
  h->own_stream = nullptr;
  h->bounce = nullptr;
  h->tick_parts = 0;
  h->split_stream[0] = h->split_stream[1] = nullptr;
  h->split_fork = h->split_join[0] = h->split_join[1] = nullptr;
  h->split_pending = false;
  h->sub_lo = h->sub_hi = 0;
  h->probe = 0;
  h->d_summary = nullptr;
  h->d_const = nullptr;
  h->fit_scratch = nullptr;
  h->fit_scratch_bytes = 0;
  h->loop_seq = 0;
  h->loop_io.on = false;
  h->loop_pending = false;
  h->loop_pin = nullptr;
  h->sqn_alt = nullptr;
  h->loop_pending_decided = false;
  h->prof_mask = 0;
  h->prof_stride = 1;
  memset(h->prof_seen, 0, sizeof h->prof_seen);
  h->tick_count = 0;
  h->opt_ftol = 0.0;
  h->opt_memory = -1;  // auto: 4 pairs for the generic instance, none for MPC with a diagonal R1 (rcg_set_optimizer)
  h->cur_a = h->cur_b = nullptr;
  h->scope_due = false;
  h->order_ev = nullptr;
  h->release_ev = nullptr;
  memset(h->last, 0, sizeof h->last);
  h->sys = cfg->sys_id == RCG_SYS_3WROBOT ? &kVt3WRobot : (cfg->sys_id == RCG_SYS_3WROBOT_NI ? &kVt3WRobotNI : &kVt2Tank);
  memset(h->prof_ms, 0, sizeof h->prof_ms);
  memset(h->prof_n, 0, sizeof h->prof_n);
  memset(h->f, 0, sizeof h->f);
  memset(h->fbytes, 0, sizeof h->fbytes);
  if (h->cfg.buffer_size > 0) {
    // Ncritic = min(Ncritic, buffer_size - 1)  (controllers.py:1015)
    if (h->cfg.n_critic > h->cfg.buffer_size - 1) h->cfg.n_critic = h->cfg.buffer_size - 1;
  }
  bool any_bnd = false;
  for (int i = 0; i < 2 * du; ++i) any_bnd = any_bnd || cfg->ctrl_bnds[i] != 0.0;
  if (!any_bnd) h->cfg.flags |= RCG_FLAG_NO_CLIP;  // `if self.ctrl_bnds.any()` (systems.py:241)

  // one small constant block (layout: rcg_handle.hpp)
  if (hipMalloc(&h->d_const, kConstBytes) != hipSuccess) {
    rcg_fail(nullptr, RCG_ERR_HIP, "rcg_create: allocating the constant block");
    delete h;
    return RCG_ERR_HIP;
  }
  {
    unsigned char blk[kConstBytes];
    memset(blk, 0, sizeof blk);
    build_params<float>(h, &h->p32, reinterpret_cast<float*>(blk), reinterpret_cast<const float*>(h->d_const));
    build_params<double>(h, &h->p64, reinterpret_cast<double*>(blk + kConstR64),
                         reinterpret_cast<const double*>((unsigned char*)h->d_const + kConstR64));
    double* wc = reinterpret_cast<double*>(blk + kConstW);
    for (int i = 0; i < 40; ++i) {
      wc[i] = cfg->w_init[i];
      wc[40 + i] = cfg->w_min[i];
      wc[80 + i] = cfg->w_max[i];
    }
    hipError_t er = hipMemcpy(h->d_const, blk, sizeof blk, hipMemcpyHostToDevice);
    if (er != hipSuccess) {
      rcg_fail(nullptr, RCG_ERR_HIP, "rcg_create: uploading the constant block: %s", hipGetErrorString(er));
      rcg_destroy(h);
      return RCG_ERR_HIP;
    }
  }

  const size_t B = (size_t)cfg->batch, e = h->esz;
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
  h->fbytes[RCG_FIELD_ACTION_SQN] = (size_t)cfg->n_actor * du * B * e;
  const int dd = cfg->sys_id == RCG_SYS_2TANK ? 1 : 2;  // dim_disturb of the presets (main_*.py dim_disturb)
  if (cfg->flags & RCG_FLAG_DISTURB) {
    h->fbytes[RCG_FIELD_DISTURB] = dd * B * e;
    h->fbytes[RCG_FIELD_SUBSTEP_IDX] = B * 4;
  }
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
      rcg_fail(nullptr, RCG_ERR_HIP, "rcg_create: allocating field %d (%zu bytes): %s", i, h->fbytes[i],
               hipGetErrorString(er));
      rcg_destroy(h);
      return RCG_ERR_HIP;
    }
  }
  if (hipMalloc((void**)&h->d_summary, 6 * sizeof(double)) != hipSuccess) {
    rcg_fail(nullptr, RCG_ERR_HIP, "rcg_create: allocating summary scratch");
    rcg_destroy(h);
    return RCG_ERR_HIP;
  }
  int rc = RCG_OK;
  const bool crit = cfg->buffer_size > 0;
  if (cfg->dtype == RCG_F64) {
    rc = fill_rows<double>(h, h->f[RCG_FIELD_ACTION], du, cfg->action_init);
    if (rc == RCG_OK && crit) rc = fill_rows<double>(h, h->f[RCG_FIELD_W_CRITIC], h->dc, cfg->w_init);
    if (rc == RCG_OK && crit) rc = fill_rows<double>(h, h->f[RCG_FIELD_W_PREV], h->dc, cfg->w_init);
  } else {
    rc = fill_rows<float>(h, h->f[RCG_FIELD_ACTION], du, cfg->action_init);
    if (rc == RCG_OK && crit) rc = fill_rows<float>(h, h->f[RCG_FIELD_W_CRITIC], h->dc, cfg->w_init);
    if (rc == RCG_OK && crit) rc = fill_rows<float>(h, h->f[RCG_FIELD_W_PREV], h->dc, cfg->w_init);
  }
  if (rc == RCG_OK && (cfg->flags & RCG_FLAG_DISTURB))
    rc = cfg->dtype == RCG_F64 ? fill_rows<double>(h, h->f[RCG_FIELD_DISTURB], dd, cfg->disturb_init)
                               : fill_rows<float>(h, h->f[RCG_FIELD_DISTURB], dd, cfg->disturb_init);
  // the memsets and fills above ran on the NULL stream: finish them here, so that a (non-blocking) stream installed
  // with rcg_set_stream cannot race with them
  if (rc == RCG_OK && hipStreamSynchronize(h->stream) != hipSuccess)
    rc = rcg_fail(h, RCG_ERR_HIP, "rcg_create: synchronising the initial fills");
  if (rc != RCG_OK) {
    g_err = h->err;
    rcg_destroy(h);
    return rc;
  }
  *out = h;
  return RCG_OK;
}

int rcg_destroy(rcg_handle* h) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_OK;
  (void)hipStreamSynchronize(h->stream);
  for (int i = 0; i < RCG_FIELD_COUNT_; ++i)
    if (h->f[i]) (void)hipFree(h->f[i]);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  for (int p = 0; p < 2; ++p) {
    if (h->split_stream[p]) {
      (void)hipStreamSynchronize(h->split_stream[p]);
      (void)hipStreamDestroy(h->split_stream[p]);
    }
    if (h->split_join[p]) (void)hipEventDestroy(h->split_join[p]);
  }
  if (h->split_fork) (void)hipEventDestroy(h->split_fork);
  if (h->bounce) (void)hipHostFree(h->bounce);
  if (h->loop_pin) (void)hipHostFree(h->loop_pin);
  if (h->sqn_alt) (void)hipFree(h->sqn_alt);
  if (h->d_summary) (void)hipFree(h->d_summary);
  if (h->d_const) (void)hipFree(h->d_const);
  if (h->fit_scratch) (void)hipFree(h->fit_scratch);
  for (auto& p : h->ev_pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : h->ev_free) (void)hipEventDestroy(e);
  if (h->order_ev) (void)hipEventDestroy(h->order_ev);
  if (h->release_ev) (void)hipEventDestroy(h->release_ev);
  delete h;
  return RCG_OK;
}

int rcg_set_stream(rcg_handle* h, void* hip_stream) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if ((hipStream_t)hip_stream == h->stream) return RCG_OK;
  // work already queued on the old stream is finished before the first launch on the new one: the two streams are
  // not ordered with respect to each other (torch.cuda.Stream() is non-blocking)
  (void)prof_drain(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)hip_stream;
  return RCG_OK;
}

int rcg_use_own_stream(rcg_handle* h) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (!h->own_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  return rcg_set_stream(h, (void*)h->own_stream);
}

int rcg_wait_stream(rcg_handle* h, void* producer_stream) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if ((hipStream_t)producer_stream == h->stream) return RCG_OK;  // same stream: already ordered
  if (!h->order_ev) HIPCHK(h, hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));
  HIPCHK(h, hipEventRecord(h->order_ev, (hipStream_t)producer_stream));
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->order_ev, 0));
  return RCG_OK;
}

int rcg_release_stream(rcg_handle* h, void* consumer_stream) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if ((hipStream_t)consumer_stream == h->stream) return RCG_OK;  // same stream: already ordered
  if (!h->release_ev) HIPCHK(h, hipEventCreateWithFlags(&h->release_ev, hipEventDisableTiming));
  HIPCHK(h, hipEventRecord(h->release_ev, h->stream));
  HIPCHK(h, hipStreamWaitEvent((hipStream_t)consumer_stream, h->release_ev, 0));
  return RCG_OK;
}

int rcg_synchronize(rcg_handle* h) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int rcg_dev_alloc(rcg_handle* h, uint64_t bytes, void** dev_out) {
  DeviceGuard dev_guard(h);
  if (!h || !dev_out) return RCG_ERR_BAD_ARG;
  HIPCHK(h, hipMalloc(dev_out, bytes ? bytes : 16));
  return RCG_OK;
}

int rcg_dev_free(rcg_handle* h, void* dev) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipFree(dev));
  return RCG_OK;
}

int rcg_memcpy_h2d(rcg_handle* h, void* dev_dst, const void* host_src, uint64_t bytes) {
  DeviceGuard dev_guard(h);
  if (!h || !dev_dst || !host_src) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_memcpy_h2d: null argument");
  HIPCHK(h, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

// Device -> pageable host memory on the handle's stream, finished on return.  Small reads (the B = 1 drop-in loop makes three
// per simulation step: the state, the stage cost, the decision) go through a pinned bounce buffer: a copy into pageable
// memory makes the runtime stage and wait by itself (30-37 us per call measured, tools/b1_profile.py), into pinned memory it is
// one DMA and one wait.
static int copy_to_host(rcg_handle* h, void* host_dst, const void* dev_src, size_t bytes) {
  if (bytes <= kBounceBytes) {
    if (!h->bounce && hipHostMalloc(&h->bounce, kBounceBytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) h->bounce = nullptr;
    if (h->bounce) {
      HIPCHK(h, hipMemcpyAsync(h->bounce, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      memcpy(host_dst, h->bounce, bytes);
      return RCG_OK;
    }
  }
  HIPCHK(h, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int rcg_memcpy_d2h(rcg_handle* h, void* host_dst, const void* dev_src, uint64_t bytes) {
  DeviceGuard dev_guard(h);
  if (!h || !host_dst || !dev_src) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_memcpy_d2h: null argument");
  return copy_to_host(h, host_dst, dev_src, (size_t)bytes);
}

static int check_field(rcg_handle* h, int field, const char* who) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (field < 0 || field >= RCG_FIELD_COUNT_) return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: bad field %d", who, field);
  if (!h->f[field])
    return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: field %d is not allocated for this configuration", who, field);
  return RCG_OK;
}

int rcg_set_field(rcg_handle* h, int field, const void* src, int where) {
  DeviceGuard dev_guard(h);
  int rc = check_field(h, field, "rcg_set_field");
  if (rc) return rc;
  if (!src) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_set_field: null src");
  const hipMemcpyKind kind = where == RCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  HIPCHK(h, hipMemcpyAsync(h->f[field], src, h->fbytes[field], kind, h->stream));
  if (field == RCG_FIELD_STATE)  // a freshly set state is also its own "previous" state
    HIPCHK(h, hipMemcpyAsync(h->f[RCG_FIELD_STATE_PREV], src, h->fbytes[field], kind, h->stream));
  if (where == RCG_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCG_OK;
}

int rcg_get_field(rcg_handle* h, int field, void* dst, int where) {
  DeviceGuard dev_guard(h);
  int rc = check_field(h, field, "rcg_get_field");
  if (rc) return rc;
  if (!dst) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_get_field: null dst");
  if (where == RCG_HOST) return copy_to_host(h, dst, h->f[field], h->fbytes[field]);
  HIPCHK(h, hipMemcpyAsync(dst, h->f[field], h->fbytes[field], hipMemcpyDeviceToDevice, h->stream));
  return RCG_OK;
}

int64_t rcg_field_bytes(const rcg_handle* h, int field) {
  if (!h || field < 0 || field >= RCG_FIELD_COUNT_) return 0;
  return (int64_t)h->fbytes[field];
}

int rcg_field_ptr(rcg_handle* h, int field, void** dev_out) {
  int rc = check_field(h, field, "rcg_field_ptr");
  if (rc) return rc;
  if (!dev_out) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_field_ptr: null out");
  *dev_out = h->f[field];
  return RCG_OK;
}

// ---- stateless operators --------------------------------------------------------------------
int rcg_rhs(rcg_handle* h, const void* state, const void* action, void* dstate, void* clipped_action, int32_t n,
            int32_t clip) {
  DeviceGuard dev_guard(h);
  if (!h || !state || !action || !dstate || n < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_rhs: bad argument");
  return h->sys->rhs(h, state, action, dstate, clipped_action, n, clip);
}

int rcg_rhs_full(rcg_handle* h, const void* state, const void* disturb, const void* action, const void* xi, void* dstate,
                 void* ddisturb, void* clipped_action, int32_t n, int32_t clip) {
  DeviceGuard dev_guard(h);
  if (!h || !state || !disturb || !action || !xi || !dstate || !ddisturb || n < 1)
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_rhs_full: bad argument");
  if (!(h->cfg.flags & RCG_FLAG_DISTURB))
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_rhs_full: the handle was created without RCG_FLAG_DISTURB");
  return h->sys->rhs_full(h, state, disturb, action, xi, dstate, ddisturb, clipped_action, n, clip);
}

int rcg_disturb_noise(rcg_handle* h, void* bits_out, void* xi_out) {
  DeviceGuard dev_guard(h);
  if (!h || (!bits_out && !xi_out)) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_disturb_noise: no output given");
  if (!(h->cfg.flags & RCG_FLAG_DISTURB))
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_disturb_noise: the handle was created without RCG_FLAG_DISTURB");
  const long B = h->cfg.batch;
  const int32_t* ep = (const int32_t*)h->f[RCG_FIELD_EPISODE_IDX];
  const int32_t* sub = (const int32_t*)h->f[RCG_FIELD_SUBSTEP_IDX];
  if (h->cfg.dtype == RCG_F64)
    hipLaunchKernelGGL((k_noise<double>), dim3(blocks_for(B)), dim3(256), 0, h->stream, ep, sub, (uint32_t*)bits_out,
                       (double*)xi_out, B, (uint64_t)h->cfg.seed, (int64_t)h->cfg.env_id_base);
  else
    hipLaunchKernelGGL((k_noise<float>), dim3(blocks_for(B)), dim3(256), 0, h->stream, ep, sub, (uint32_t*)bits_out,
                       (float*)xi_out, B, (uint64_t)h->cfg.seed, (int64_t)h->cfg.env_id_base);
  HIPCHK(h, hipGetLastError());
  return RCG_OK;
}

int rcg_stage_obj(rcg_handle* h, const void* obs, const void* act, void* out, int32_t n) {
  DeviceGuard dev_guard(h);
  if (!h || !obs || !act || !out || n < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_stage_obj: bad argument");
  return h->sys->stage_obj(h, obs, act, out, n);
}

int rcg_critic(rcg_handle* h, const void* obs, const void* act, const void* w, void* out, int32_t n) {
  DeviceGuard dev_guard(h);
  if (!h || !obs || !act || !w || !out || n < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_critic: bad argument");
  return h->sys->critic(h, obs, act, w, out, n);
}

int rcg_actor_cost(rcg_handle* h, const void* cand, int32_t K, const void* obs, const void* state_sys, const void* w,
                   void* J) {
  DeviceGuard dev_guard(h);
  if (!h || !cand || !J) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_actor_cost: cand and J are required");
  return h->sys->actor(h, "rcg_actor_cost", cand, K, obs, state_sys, w, J, nullptr, nullptr, nullptr, false, false);
}

int rcg_actor_argmin(rcg_handle* h, const void* cand, int32_t K, const void* obs, const void* state_sys, void* action,
                     void* best_J, int32_t* best_idx) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  return h->sys->actor(h, "rcg_actor_argmin", cand, K, obs, state_sys, nullptr, nullptr, action, best_J, best_idx,
                       false, false);
}

int rcg_critic_cost(rcg_handle* h, const void* w, void* Jc) {
  DeviceGuard dev_guard(h);
  if (!h || !Jc) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_critic_cost: Jc is required");
  if (!h->f[RCG_FIELD_OBS_BUF])
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_critic_cost: handle has no critic buffers (buffer_size = 0)");
  return h->sys->critic_cost(h, w, Jc);
}

// ---- stateful steps -------------------------------------------------------------------------
int rcg_sim_step(rcg_handle* h, int32_t n_substeps) {
  DeviceGuard dev_guard(h);
  if (!h || n_substeps < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_sim_step: n_substeps must be >= 1");
  return h->sys->sim_step(h, n_substeps);
}

int rcg_sim_step_h(rcg_handle* h, int32_t n_substeps, double step) {
  DeviceGuard dev_guard(h);
  if (!h || n_substeps < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_sim_step_h: n_substeps must be >= 1");
  if (!(step > 0.0) || !(step < 1e300))
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_sim_step_h: step must be positive and finite (got %g)", step);
  // the kernels take the substep from the by-value parameter block: lend it this call's length
  const float d32 = h->p32.dt_sim;
  const double d64 = h->p64.dt_sim;
  h->p64.dt_sim = step / (double)n_substeps;
  h->p32.dt_sim = (float)h->p64.dt_sim;
  const int rc = h->sys->sim_step(h, n_substeps);
  h->p32.dt_sim = d32;
  h->p64.dt_sim = d64;
  return rc;
}

// w_critic = w_prev = clip(w_init, Wmin, Wmax): what the reference's SLSQP returns when the TD stack is empty
// (Ncritic = 1: _critic_cost is identically 0, controllers.py:1227-1245, so minimize() stops at its start point)
static int critic_keep_init(rcg_handle* h) {
  double w[40];
  for (int i = 0; i < h->dc; ++i) {
    const double lo = h->cfg.w_min[i], hi = h->cfg.w_max[i], v = h->cfg.w_init[i];
    w[i] = v < lo ? lo : (v > hi ? hi : v);
  }
  for (int f : {RCG_FIELD_W_CRITIC, RCG_FIELD_W_PREV}) {
    const int rc = h->cfg.dtype == RCG_F64 ? fill_rows<double>(h, h->f[f], h->dc, w) : fill_rows<float>(h, h->f[f], h->dc, w);
    if (rc) return rc;
  }
  return RCG_OK;
}

int rcg_critic_update(rcg_handle* h, int32_t do_fit) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (!h->f[RCG_FIELD_OBS_BUF])
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_critic_update: handle has no critic buffers (buffer_size = 0)");
  const int m = h->cfg.n_critic - 1;  // any number of TD rows: <= 8 on the register kernels, beyond on k_critic_fit_gen
  if (do_fit && m < 1) {  // empty TD stack: push only, the weights stay at the (clipped) initial guess
    const int rc = h->sys->critic_update(h, 0, 1, 0);
    return rc ? rc : critic_keep_init(h);
  }
  return h->sys->critic_update(h, 0, 1, do_fit ? 1 : 0);
}

// Argument checks of the decision step, made BEFORE the tick mutates anything (env step, buffer push): a refused
// call leaves the handle exactly as it was.
static int check_candidates(rcg_handle* h, const char* who, const void* cand, int32_t K) {
  if (K < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: K must be >= 1", who);
  if (!cand && h->du == 2) {
    int g = 1;
    while ((long)(g + 1) * (g + 1) <= (long)K) ++g;
    if (g * g != K) return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: generated grid for du = 2 needs a square K (got %d)", who, K);
  }
  if (h->cfg.mode != RCG_MODE_MPC && !h->f[RCG_FIELD_W_CRITIC])
    return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: RQL/SQL need critic weights (buffer_size > 0)", who);
  return RCG_OK;
}

// RQL / SQL bookkeeping of one tick, between the env step and the decision (controllers.py:1458-1477): env step + buffer
// push + critic fit.  critic_period = critic_every_ticks * sampling_time: the reference starts critic_clock at t0 and refits
// when t - critic_clock >= critic_period; tick j of an episode happens at t0 + (j+1)*dt, so the fits fall on ticks every-1,
// 2*every-1, ...
static int tick_critic_phase(rcg_handle* h, const char* who) {
  (void)who;
  const int every = h->cfg.critic_every_ticks > 1 ? h->cfg.critic_every_ticks : 1;
  const bool do_fit = ((h->tick_count + 1) % every) == 0;
  const int m = h->cfg.n_critic - 1;
  if ((h->cfg.flags & RCG_FLAG_DISTURB) || (do_fit && m < 1)) {  // the disturbed env step has its own kernel
    const int rc = h->sys->sim_step(h, h->cfg.substeps_per_tick);
    if (rc) return rc;
    return rcg_critic_update(h, do_fit ? 1 : 0);
  }
  // env step + buffer push + fit: one launch (rcg_critic_fit.hpp)
  return h->sys->critic_update(h, h->cfg.substeps_per_tick, 1, do_fit ? 1 : 0);
}

// Is this tick one the handle runs as two halves on two internal streams?  RQL / SQL (the fit is what the split hides), a
// caller's tensor whose decision goes to k_actor_dma, the fused [env step + push + fit] launch (no disturbance model, 1 .. 8 TD
// rows), and enough envs for two launches to be worth it (rcg_set_tick_parts).
static bool tick_splits(rcg_handle* h, const void* cand, int32_t K) {
  if (h->tick_parts == 1 || h->cfg.mode == RCG_MODE_MPC || !cand) return false;
  // automatic: only on a stream the handle OWNS (rcg_use_own_stream).  On a caller's stream (rcg_set_stream) or the null stream
  // the caller may enqueue its own work behind the tick - a kernel reading rcg_field_ptr(ACTION) - and that work must find the
  // tick finished: a split tick returns with half the batch on internal streams the caller's stream does not wait for
  // (VERDICT r5 weak 6).  There the pipelining is an explicit opt-in, rcg_set_tick_parts(h, 2), whose contract is rcg_join.
  if (h->tick_parts == 0 && (h->cfg.batch < kSplitMinBatch || !h->own_stream || h->stream != h->own_stream)) return false;
  const int m = h->cfg.n_critic - 1;
  if ((h->cfg.flags & RCG_FLAG_DISTURB) || m < 1 || m > kFitMaxRows || h->cfg.batch < 2048) return false;
  h->probe = 1;
  const int rc = h->sys->actor(h, "rcg_control_tick", cand, K, nullptr, nullptr, nullptr, nullptr, h->f[RCG_FIELD_ACTION],
                               h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], true, false);
  const bool yes = rc == RCG_OK && h->probe == 3;
  h->probe = 0;
  return yes;
}

static int split_streams(rcg_handle* h) {
  for (int p = 0; p < 2; ++p) {
    if (!h->split_stream[p]) HIPCHK(h, hipStreamCreateWithFlags(&h->split_stream[p], hipStreamNonBlocking));
    if (!h->split_join[p]) HIPCHK(h, hipEventCreateWithFlags(&h->split_join[p], hipEventDisableTiming));
  }
  if (!h->split_fork) HIPCHK(h, hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
  return RCG_OK;
}

int rcg_set_tick_parts(rcg_handle* h, int32_t parts) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (parts < 0 || parts > 2) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_set_tick_parts: parts must be 0 (auto), 1 or 2");
  h->tick_parts = parts;
  return RCG_OK;
}

int rcg_join(rcg_handle* h) {
  DeviceGuard dev_guard(h);  // (the guard joins)
  return h ? RCG_OK : RCG_ERR_BAD_ARG;
}

int rcg_control_tick(rcg_handle* h, const void* cand, int32_t K) {
  DeviceGuard dev_guard(h, NoJoin{});
  if (!h) return RCG_ERR_BAD_ARG;
  int rc = check_candidates(h, "rcg_control_tick", cand, K);
  if (rc) {
    join_split(h);
    return rc;
  }
  if (tick_splits(h, cand, K)) {
    rc = split_streams(h);
    if (rc) return rc;
    // fork: whatever the handle's stream holds (the caller's candidates, rcg_wait_stream, a set_field) comes first
    HIPCHK(h, hipEventRecord(h->split_fork, h->stream));
    for (int p = 0; p < 2; ++p) HIPCHK(h, hipStreamWaitEvent(h->split_stream[p], h->split_fork, 0));
    hipStream_t const own = h->stream;
    const int half = (h->cfg.batch / 2) & ~1023;
    for (int p = 0; p < 2 && rc == RCG_OK; ++p) {
      h->stream = h->split_stream[p];
      h->sub_lo = p ? half : 0;
      h->sub_hi = p ? h->cfg.batch : half;
      rc = tick_critic_phase(h, "rcg_control_tick");
      if (rc == RCG_OK)
        rc = h->sys->actor(h, "rcg_control_tick", cand, K, nullptr, nullptr, nullptr, nullptr, h->f[RCG_FIELD_ACTION],
                           h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], true, false);
    }
    h->stream = own;
    h->sub_lo = h->sub_hi = 0;
    h->split_pending = true;
    if (rc == RCG_OK) h->tick_count += 1;
    return rc;
  }
  join_split(h);
  bool sim_first = true;  // MPC: env step, then the decision, both issued by the actor launcher
  if (h->cfg.mode != RCG_MODE_MPC) {  // RQL/SQL: the critic bookkeeping sits between the two
    rc = tick_critic_phase(h, "rcg_control_tick");
    if (rc) return rc;
    sim_first = false;
  }
  // (every argument-dependent refusal of the decision step is in check_candidates above; what can still fail below is a
  // HIP launch error, which leaves the handle unusable anyway.  The tick is counted once it has been issued whole.)
  rc = h->sys->actor(h, "rcg_control_tick", cand, K, nullptr, nullptr, nullptr, nullptr, h->f[RCG_FIELD_ACTION],
                     h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], true, sim_first);
  if (rc == RCG_OK) h->tick_count += 1;
  return rc;
}

// Batches up to this size are launch-bound under per-tick launches (two launches of a few microseconds of work each):
// rcg_control_tick_n runs their T ticks in one persistent launch when the handle's mode allows it
static const int kPersistentTicksMaxBatch = 16384;

// A caller's tensor under the persistent kernel: k_ticks keeps a wave's rows in LDS for all T ticks when they fit 32 KB
// (op_ticks: stage_once); beyond that it re-stages every tile every tick through plain loads, which only pays while the whole
// tensor stays in the Infinity Cache (256 MB; half of it granted here) - K = 1024 rows of 80 B at 16 384 envs is 1.3 GB per
// tick, and the per-tick loop on k_actor_dma streams that at 4.6-5.5 TB/s against 2.9-3.7 for plain staging.
static bool ticks_rows_stay_close(const rcg_handle* h, const void* cand, int32_t K) {
  if (cand && h->cfg.n_actor * h->du > RCG_MAX_ROW) return false;  // long rows: no LDS tile, the loop of single ticks
  if (!cand || K < 1) return true;
  const size_t row_bytes = (size_t)h->cfg.n_actor * h->du * h->esz;
  int kp = 1;
  while (kp < K) kp <<= 1;
  const size_t rows_wave = K >= 64 ? (size_t)K : (size_t)(64 / kp) * K;
  if (rows_wave * row_bytes <= (size_t)32 * 1024) return true;
  return (size_t)h->cfg.batch * K * row_bytes <= (size_t)128 << 20;
}

int rcg_control_tick_n(rcg_handle* h, const void* cand, int32_t K, int32_t T) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (T < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_control_tick_n: T must be >= 1");
  if (T > 1 && h->cfg.mode == RCG_MODE_MPC && h->cfg.batch <= kPersistentTicksMaxBatch &&
      ticks_rows_stay_close(h, cand, K)) {
    // MPC (any stage-cost structure, with or without the disturbance model): k_ticks keeps the env in registers and, for a
    // caller's tensor, the wave's candidate rows in LDS - every field ends as T single ticks leave it, bit for bit
    DeviceGuard dev_guard(h);
    int rc = check_candidates(h, "rcg_control_tick_n", cand, K);
    if (rc) return rc;
    rc = h->sys->ticks(h, T, K, cand);
    if (rc == RCG_OK) h->tick_count += T;
    return rc;
  }
  if (T > 1 && !cand && h->cfg.mode != RCG_MODE_MPC && h->cfg.batch <= kPersistentTicksMaxBatch &&
      h->cfg.n_critic - 1 >= 1 && h->cfg.n_critic - 1 <= kFitMaxRows && !(h->cfg.flags & RCG_FLAG_DISTURB)) {
    const int rc = rcg_control_ticks(h, T, K);  // RQL / SQL, generated grid: k_ticks_mem
    if (rc != RCG_ERR_UNSUPPORTED) return rc;  // (no instance for this observation target: the loop below)
    h->err.clear();  // the refusal was this function's own probe, not the caller's error (rcg_last_error after RCG_OK)
  }
  if (T > 1 && cand && h->cfg.mode != RCG_MODE_MPC && h->cfg.batch <= kPersistentTicksMaxBatch &&
      h->cfg.n_critic - 1 >= 1 && h->cfg.n_critic - 1 <= kFitMaxRows && !(h->cfg.flags & RCG_FLAG_DISTURB) &&
      h->p32.stage_kind == 0 && ticks_rows_stay_close(h, cand, K)) {
    // RQL / SQL over a caller's tensor (round 5): k_ticks_mem with the streamed decision phase - the accumulation order of
    // k_actor_dma / k_actor_dma_packed, so every field ends as T single ticks on those kernels leave it
    DeviceGuard dev_guard(h);
    int rc = check_candidates(h, "rcg_control_tick_n", cand, K);
    if (rc) return rc;
    rc = h->sys->ticks_mem(h, T, K, cand);
    if (rc == RCG_OK) {
      h->tick_count += T;
      return rc;
    }
    if (rc != RCG_ERR_UNSUPPORTED) return rc;
    h->err.clear();
  }
  for (int32_t t = 0; t < T; ++t) {
    const int rc = rcg_control_tick(h, cand, K);
    if (rc) return rc;
  }
  return RCG_OK;
}

int rcg_control_ticks(rcg_handle* h, int32_t T, int32_t K) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (T < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_control_ticks: T must be >= 1");
  int rc = check_candidates(h, "rcg_control_ticks", nullptr, K);
  if (rc) return rc;
  if (h->cfg.mode != RCG_MODE_MPC) {  // RQL / SQL: the launches of a tick as phases of one persistent launch
    const int m = h->cfg.n_critic - 1;
    if (m < 1 || m > kFitMaxRows || (h->cfg.flags & RCG_FLAG_DISTURB))
      return rcg_fail(h, RCG_ERR_UNSUPPORTED,
                      "rcg_control_ticks: RQL/SQL need 1 <= Ncritic-1 <= %d rows and no disturbance model here (loop "
                      "rcg_control_tick)", kFitMaxRows);
    rc = h->sys->ticks_mem(h, T, K, nullptr);
    if (rc == RCG_OK) h->tick_count += T;
    return rc;
  }
  rc = h->sys->ticks(h, T, K, nullptr);
  if (rc == RCG_OK) h->tick_count += T;
  return rc;
}

int rcg_set_optimizer(rcg_handle* h, int32_t memory) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (memory < -1 || memory > OPT_MAXM)
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_set_optimizer: memory must be in [0, %d], or -1 for the default", OPT_MAXM);
  h->opt_memory = memory;
  return RCG_OK;
}

int rcg_set_optimizer_tol(rcg_handle* h, double ftol) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (!(ftol >= 0.0) || !(ftol < 1e300)) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_set_optimizer_tol: ftol must be finite and >= 0");
  h->opt_ftol = ftol;
  return RCG_OK;
}

// refusals of the optimiser that depend on the handle's shape, made before a tick mutates anything
static int check_optimizer(rcg_handle* h, const char* who) {
  if (h->cfg.mode != RCG_MODE_MPC && !h->f[RCG_FIELD_W_CRITIC])
    return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: RQL/SQL need critic weights (buffer_size > 0)", who);
  const size_t lds = opt_wave_lds_bytes(h);
  if (lds > (size_t)160 * 1024)
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "%s: horizon %d with %d curvature pairs needs %zu B of LDS per wave (rcg_set_optimizer)",
                    who, h->cfg.n_actor, opt_memory_of(h), lds);
  return RCG_OK;
}

int rcg_actor_optimize(rcg_handle* h, int32_t iters, const void* obs, const void* state_sys, const void* u_init,
                       void* u_opt, void* action, void* best_J, int32_t* n_iter) {
  DeviceGuard dev_guard(h);
  if (!h || iters < 0) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_actor_optimize: iters must be >= 0");
  const int rc = check_optimizer(h, "rcg_actor_optimize");
  if (rc) return rc;
  return h->sys->optimize(h, iters, obs, state_sys, u_init, 0, u_opt, action, best_J, n_iter, false, false);
}

int rcg_control_tick_opt(rcg_handle* h, int32_t iters, int32_t warm_start) {
  DeviceGuard dev_guard(h);
  if (!h || iters < 0) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_control_tick_opt: iters must be >= 0");
  int rc = check_optimizer(h, "rcg_control_tick_opt");
  if (rc) return rc;
  bool sim_first = true;  // MPC: the env step of the tick is issued by the optimiser's launcher
  if (h->cfg.mode != RCG_MODE_MPC) {  // RQL/SQL: env step + buffer push + critic fit come first (one launch)
    rc = tick_critic_phase(h, "rcg_control_tick_opt");
    if (rc) return rc;
    sim_first = false;
  }
  const bool warm = warm_start && h->tick_count > 0;  // nothing to shift before the episode's first decision
  void* sqn = h->f[RCG_FIELD_ACTION_SQN];
  rc = h->sys->optimize(h, iters, nullptr, nullptr, warm ? sqn : nullptr, warm ? 1 : 0, sqn, h->f[RCG_FIELD_ACTION],
                        h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], true, sim_first);
  if (rc == RCG_OK) h->tick_count += 1;
  return rc;
}

// ---- rcg_loop_step: one iteration of the reference's headless loop in ONE call and ONE host wait (rcg.h) --------------------
// rcg_loop_step_begin enqueues the iteration and returns; rcg_loop_step_end waits for it and hands over the rows; rcg_loop_step is
// the two back to back.  The rows travel through a pinned buffer of the handle's own (not the one the small device-to-host reads
// of the other entry points share): [out rows | action_in | one sequence number per env].
int rcg_loop_step_begin(rcg_handle* h, const double* action_in, double step_h, int32_t n_substeps, int32_t flags, int32_t iters) {
  DeviceGuard dev_guard(h);
  if (!h || n_substeps < 1 || iters < 0) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_loop_step: bad argument");
  if (!(step_h > 0.0) || !(step_h < 1e300)) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_loop_step: step must be positive and finite");
  if (h->loop_pending) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_loop_step_begin: the previous step has not been collected (rcg_loop_step_end)");
  const bool critic = h->cfg.mode != RCG_MODE_MPC;
  const bool decide = flags & RCG_LOOP_DECIDE, push = critic && (flags & RCG_LOOP_PUSH), fit = push && (flags & RCG_LOOP_FIT);
  if (h->cfg.flags & RCG_FLAG_DISTURB) return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_loop_step: no disturbance model on this path");
  if (fit && h->cfg.n_critic - 1 < 1) return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_loop_step: an empty TD stack (Ncritic = 1) takes the separate calls");
  const int B = h->cfg.batch, ds = h->ds, du = h->du, dc = critic ? h->dc : 0, row = ds + du + 2 + dc;
  const size_t out_bytes = (size_t)B * row * sizeof(double), in_bytes = (size_t)B * du * sizeof(double);
  if (out_bytes + in_bytes + (size_t)B * sizeof(double) > kBounceBytes)
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_loop_step: %d envs do not fit the handle's %zu-byte pinned buffer (a small-batch entry point)",
                    B, kBounceBytes);
  if (decide) {
    const int rc = check_optimizer(h, "rcg_loop_step");
    if (rc) return rc;
  }
  if (!h->loop_pin) {
    if (hipHostMalloc(&h->loop_pin, kBounceBytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
      h->loop_pin = nullptr;
      return rcg_fail(h, RCG_ERR_HIP, "rcg_loop_step: cannot allocate the pinned buffer");
    }
    memset(h->loop_pin, 0, kBounceBytes);  // (sequence numbers start at 1: no env has reported yet)
  }
  // the decision's sequence goes to a spare buffer that becomes ACTION_SQN when the step is collected (rcg_loop_step_end swaps the
  // two pointers): a step that is dropped leaves ACTION_SQN - the last decision anybody took - alone
  if (decide && !h->sqn_alt) {
    HIPCHK(h, hipMalloc(&h->sqn_alt, h->fbytes[RCG_FIELD_ACTION_SQN]));
    HIPCHK(h, hipMemsetAsync(h->sqn_alt, 0, h->fbytes[RCG_FIELD_ACTION_SQN], h->stream));
  }
  double* const p_out = (double*)h->loop_pin;
  double* const p_act = p_out + (size_t)B * row;
  double* const p_flag = p_act + (size_t)B * du;
  const double seq = (double)++h->loop_seq;  // (exact up to 2^53 calls; only this handle's loop kernels write the flags)
  if (action_in) memcpy(p_act, action_in, in_bytes);
  const double* const act_dev = action_in ? p_act : nullptr;
  // the kernels take the substep from the by-value parameter block: lend it this call's length
  const float d32 = h->p32.dt_sim;
  const double d64 = h->p64.dt_sim;
  h->p64.dt_sim = step_h / (double)n_substeps;
  h->p32.dt_sim = (float)h->p64.dt_sim;
  int rc;
  // the plain MPC decision (diagonal stage cost, no curvature pairs - every MPC preset): k_actor_opt's LOOP instance does the
  // iteration's head and tail itself
  const bool one_launch = decide && !push && h->cfg.mode == RCG_MODE_MPC && h->p64.stage_kind == 0 && opt_memory_of(h) == 0;
  if (one_launch) {
    h->loop_io.on = true;
    h->loop_io.act_in = act_dev;
    h->loop_io.n_substeps = n_substeps, h->loop_io.dc = dc;
    h->loop_io.out = p_out, h->loop_io.flag = p_flag, h->loop_io.seq = seq;
    rc = h->sys->optimize(h, iters, nullptr, h->f[RCG_FIELD_STATE_PREV], nullptr, 0, h->sqn_alt,
                          h->f[RCG_FIELD_ACTION], h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], false, false);
    h->loop_io.on = false;
  } else if (!decide && !push) {
    // not a controller sample: System.receive_action, Simulator.sim_step, CtrlOptPred.stage_obj and the transfer - ONE launch
    rc = h->sys->loop(h, act_dev, n_substeps, 1, 1, 0, dc, p_out, p_flag, seq);
  } else {
    // a sample: receive_action + sim_step (the critic modes: k_critic_fit's env step + push + fit instead), then the decision -
    // the rollout starts from the state BEFORE the step (the loop hands the controller my_sys._state one iteration late,
    // controllers.py:1056-1061, presets/main_3wrobot.py:425-428), the observation is the new state - then stage_obj + transfer
    rc = h->sys->loop(h, act_dev, n_substeps, push ? 0 : 1, 0, 0, dc, p_out, p_flag, seq);
    if (rc == RCG_OK && push) rc = h->sys->critic_update(h, n_substeps, 1, fit ? 1 : 0);
    if (rc == RCG_OK && decide)
      rc = h->sys->optimize(h, iters, nullptr, h->f[RCG_FIELD_STATE_PREV], nullptr, 0, h->sqn_alt,
                            h->f[RCG_FIELD_ACTION], h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], false, false);
    if (rc == RCG_OK) rc = h->sys->loop(h, nullptr, n_substeps, 0, 1, decide ? 1 : 0, dc, p_out, p_flag, seq);
  }
  h->p32.dt_sim = d32;
  h->p64.dt_sim = d64;
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  h->loop_pending = true;
  h->loop_pending_decided = decide;
  h->loop_pending_row = row;
  return RCG_OK;
}

int rcg_loop_step_end(rcg_handle* h, double* out) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (!h->loop_pending) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_loop_step_end: no step is pending (rcg_loop_step_begin)");
  h->loop_pending = false;
  const int B = h->cfg.batch, row = h->loop_pending_row;
  double* const p_out = (double*)h->loop_pin;
  volatile double* const f = p_out + (size_t)B * row + (size_t)B * h->du;
  const double seq = (double)h->loop_seq;
  // wait for the glue kernel's sequence numbers in pinned memory instead of for the stream (tools/sync_probe.hip: 6.9 us per launch
  // + wait against 12.4): every env's row is complete once its flag shows this call's number.  Bounded: after ~50 ms of polling
  // (a fault, a hung queue) the stream is waited for the ordinary way and whatever error it carries is returned.
  bool done = false;
  for (long spin = 0; spin < 20000000L && !done; ++spin) {
    done = true;
    for (int b = 0; b < B; ++b)
      if (f[b] != seq) {
        done = false;
        break;
      }
    if (!done) __builtin_ia32_pause();
  }
  if (!done) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < B; ++b)
      if (f[b] != seq) return rcg_fail(h, RCG_ERR_HIP, "rcg_loop_step: the glue kernel did not report env %d", b);
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the rows are read after the flags, for the compiler as well
  if (out) {
    memcpy(out, p_out, (size_t)B * row * sizeof(double));
    if (h->loop_pending_decided) std::swap(h->f[RCG_FIELD_ACTION_SQN], h->sqn_alt);  // the collected decision is THE decision now
  }  // (NULL: the step is dropped - STATE, ACTION, the critic buffers and weights keep its effects, ACTION_SQN does not)
  return RCG_OK;
}

int rcg_loop_step(rcg_handle* h, const double* action_in, double step_h, int32_t n_substeps, int32_t flags, int32_t iters,
                  double* out) {
  if (!h || !out) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_loop_step: bad argument");
  const int rc = rcg_loop_step_begin(h, action_in, step_h, n_substeps, flags, iters);
  return rc ? rc : rcg_loop_step_end(h, out);
}

static int check_search(rcg_handle* h, const char* who, int32_t K, int32_t rounds) {
  if (K < 64) return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: K must be >= 64 (a wave evaluates 64 candidates at a time)", who);
  if (rounds < 1 || rounds > 64) return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: rounds must be in [1, 64]", who);
  if (h->cfg.mode != RCG_MODE_MPC && !h->f[RCG_FIELD_W_CRITIC])
    return rcg_fail(h, RCG_ERR_BAD_ARG, "%s: RQL/SQL need critic weights (buffer_size > 0)", who);
  return RCG_OK;
}

int rcg_actor_search(rcg_handle* h, int32_t K, int32_t rounds, const void* obs, const void* state_sys, const void* centre,
                     void* u_best, void* action, void* best_J, int32_t* best_idx) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  const int rc = check_search(h, "rcg_actor_search", K, rounds);
  if (rc) return rc;
  return h->sys->search(h, K, rounds, 0, obs, state_sys, centre, 0, u_best, action, best_J, best_idx, false, false);
}

int rcg_control_tick_search(rcg_handle* h, int32_t K, int32_t rounds, int32_t warm_start) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  int rc = check_search(h, "rcg_control_tick_search", K, rounds);
  if (rc) return rc;
  bool sim_first = true;
  if (h->cfg.mode != RCG_MODE_MPC) {
    rc = tick_critic_phase(h, "rcg_control_tick_search");
    if (rc) return rc;
    sim_first = false;
  }
  const bool warm = warm_start && h->tick_count > 0;
  void* sqn = h->f[RCG_FIELD_ACTION_SQN];
  rc = h->sys->search(h, K, rounds, 0, nullptr, nullptr, warm ? sqn : nullptr, warm ? 1 : 0, sqn, h->f[RCG_FIELD_ACTION],
                      h->f[RCG_FIELD_BEST_J], (int32_t*)h->f[RCG_FIELD_BEST_IDX], true, sim_first);
  if (rc == RCG_OK) h->tick_count += 1;
  return rc;
}

int rcg_candidates_sample(rcg_handle* h, void* cand, int32_t K, int32_t round, const void* centre) {
  DeviceGuard dev_guard(h);
  if (!h || !cand) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_candidates_sample: cand is required");
  if (K < 1 || round < 0 || round > 63) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_candidates_sample: need K >= 1 and 0 <= round <= 63");
  const int R = h->cfg.n_actor * h->du;
  const int32_t* ep = (const int32_t*)h->f[RCG_FIELD_EPISODE_IDX];
  const int32_t* st = (const int32_t*)h->f[RCG_FIELD_STEP_IDX];
  // a block = KB consecutive candidates (a multiple of 64) of one env; its draws (eight floats each) live in LDS: at most 48 KB
  const int n_draws = (R + 7) / 8;
  int KB = 256;
  while (KB > 64 && (size_t)rcg::cand_block_draws(KB, n_draws, h->du) * 32 > (size_t)48 * 1024) KB >>= 1;
  const size_t lds = (size_t)rcg::cand_block_draws(KB, n_draws, h->du) * 32;
  if (lds > (size_t)160 * 1024)
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_candidates_sample: rows of %d reals need %zu B of LDS per 64 candidates", R, lds);
  if ((K + KB - 1) / KB > 65535) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_candidates_sample: K too large");
  const dim3 grid((unsigned)h->cfg.batch, (unsigned)((K + KB - 1) / KB)), block(256);
#define RCG_SAMPLE(DU, real, P)                                                                                          \
  do {                                                                                                                   \
    auto fn = k_cand_sample<DU, real>;                                                                                   \
    if (lds > (size_t)64 * 1024) /* very long rows (R > 190): beyond the default dynamic-LDS limit */                    \
      HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                    (int)lds));                                                                          \
    hipLaunchKernelGGL(fn, grid, block, lds, h->stream, (real*)cand, (const real*)centre, ep, st, (int)K, (int)round, R, \
                       (uint64_t)h->cfg.seed, (int64_t)h->cfg.env_id_base, (real)h->cfg.action_init[0],                  \
                       (real)h->cfg.action_init[1], KB, P);                                                              \
  } while (0)
  if (h->cfg.dtype == RCG_F64) {
    if (h->du == 1)
      RCG_SAMPLE(1, double, h->p64);
    else
      RCG_SAMPLE(2, double, h->p64);
  } else {
    if (h->du == 1)
      RCG_SAMPLE(1, float, h->p32);
    else
      RCG_SAMPLE(2, float, h->p32);
  }
#undef RCG_SAMPLE
  HIPCHK(h, hipGetLastError());
  return RCG_OK;
}

int rcg_nominal_action(rcg_handle* h, const void* obs, void* action, void* lyap, int32_t n, double ctrl_gain,
                       const double* ctrl_pars, int32_t clip) {
  DeviceGuard dev_guard(h);
  if (!h || !obs || (!action && !lyap) || n < 1)
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_nominal_action: obs and one of action/lyap are required, n >= 1");
  return h->sys->nominal(h, obs, action, lyap, nullptr, n, ctrl_gain, ctrl_pars, clip, false);
}

int rcg_nominal_theta(rcg_handle* h, const void* obs, void* theta, int32_t n) {
  DeviceGuard dev_guard(h);
  if (!h || !obs || !theta || n < 1) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_nominal_theta: obs, theta and n >= 1 are required");
  if (h->cfg.sys_id != RCG_SYS_3WROBOT)
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_nominal_theta: only CtrlNominal3WRobot has a theta search");
  return h->sys->nominal(h, obs, nullptr, nullptr, theta, n, 1.0, nullptr, 0, false);
}

int rcg_control_tick_nominal(rcg_handle* h, double ctrl_gain, const double* ctrl_pars) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (h->cfg.sys_id == RCG_SYS_2TANK)  // refuse before the env is stepped
    return rcg_fail(h, RCG_ERR_UNSUPPORTED, "rcg_control_tick_nominal: the reference defines no nominal controller for 2tank");
  int rc = h->sys->sim_step(h, h->cfg.substeps_per_tick);
  if (rc) return rc;
  rc = h->sys->nominal(h, h->f[RCG_FIELD_STATE], h->f[RCG_FIELD_ACTION], nullptr, nullptr, h->cfg.batch, ctrl_gain,
                       ctrl_pars, 1, true);
  if (rc == RCG_OK) h->tick_count += 1;
  return rc;
}

int rcg_episode_reset(rcg_handle* h) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  const long B = h->cfg.batch;
  // the tick counter belongs to the episode: the first decision of the new episode has no previous optimum to shift
  // (rcg_control_tick_opt warm start) and the critic period restarts with the controller clock
  h->tick_count = 0;
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
  if (h->cfg.flags & RCG_FLAG_DISTURB) {  // disturbance back to disturb_init, noise counter word 3 back to 0
    const int dd = h->cfg.sys_id == RCG_SYS_2TANK ? 1 : 2;
    const int rc = h->cfg.dtype == RCG_F64 ? fill_rows<double>(h, h->f[RCG_FIELD_DISTURB], dd, h->cfg.disturb_init)
                                           : fill_rows<float>(h, h->f[RCG_FIELD_DISTURB], dd, h->cfg.disturb_init);
    if (rc) return rc;
    HIPCHK(h, hipMemsetAsync(h->f[RCG_FIELD_SUBSTEP_IDX], 0, h->fbytes[RCG_FIELD_SUBSTEP_IDX], h->stream));
  }
  return RCG_OK;
}

int rcg_episode_stats(rcg_handle* h, int32_t from_accum, void* returns_out, rcg_summary* out) {
  DeviceGuard dev_guard(h);
  if (!h || !out) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_episode_stats: out is required");
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
  return s[5] > 0 ? rcg_fail(h, RCG_ERR_NONFINITE, "rcg_episode_stats: %.0f env(s) hit a non-finite state", s[5])
                  : RCG_OK;
}

// ---- checkpoint / resume --------------------------------------------------------------------
int64_t rcg_tick_count(const rcg_handle* h) { return h ? (int64_t)h->tick_count : -1; }

int rcg_set_tick_count(rcg_handle* h, int64_t ticks) {
  if (!h) return RCG_ERR_BAD_ARG;
  if (ticks < 0) return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_set_tick_count: ticks must be >= 0");
  h->tick_count = (long)ticks;
  return RCG_OK;
}

// ---- measurement ----------------------------------------------------------------------------
int rcg_profile(rcg_handle* h, int32_t enable) {
  DeviceGuard dev_guard(h);
  if (!h) return RCG_ERR_BAD_ARG;
  if (((unsigned)enable & 0xffu) == RCG_PROFILE_PAUSE) {  // stop sampling; pending samples stay, nothing is waited for
    h->prof_mask = 0;
    return RCG_OK;
  }
  const int drain_rc = prof_drain(h);
  h->prof_mask = (unsigned)enable & 0x7fu;
  h->prof_stride = (((unsigned)enable >> 8) & 0xfffu) ? (((unsigned)enable >> 8) & 0xfffu) : 1u;
  // bits 20..: launches of each kernel to let pass before the first sample (the first launch after a synchronisation
  // starts on an idle GPU and is not representative of the stream)
  const unsigned skip = (unsigned)enable >> 20;
  for (int k = 0; k < RCG_KERNEL_COUNT_; ++k)
    h->prof_seen[k] = skip ? (uint64_t)h->prof_stride - (uint64_t)(skip % h->prof_stride) : 0;
  if (enable) {
    memset(h->prof_ms, 0, sizeof h->prof_ms);
    memset(h->prof_n, 0, sizeof h->prof_n);
    for (auto& v : h->prof_samples) v.clear();
  }
  return drain_rc;
}

int rcg_profile_read(rcg_handle* h, int32_t kernel, double* total_ms, int64_t* launches) {
  DeviceGuard dev_guard(h);
  if (!h || kernel < 0 || kernel >= RCG_KERNEL_COUNT_)
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_profile_read: bad kernel id");
  const int drain_rc = prof_drain(h);
  if (total_ms) *total_ms = h->prof_ms[kernel];
  if (launches) *launches = h->prof_n[kernel];
  return drain_rc;
}

int rcg_profile_samples(rcg_handle* h, int32_t kernel, double* ms_out, int64_t cap, int64_t* n_out) {
  DeviceGuard dev_guard(h);
  if (!h || kernel < 0 || kernel >= RCG_KERNEL_COUNT_ || cap < 0 || (cap > 0 && !ms_out))
    return rcg_fail(h, RCG_ERR_BAD_ARG, "rcg_profile_samples: bad argument");
  const int drain_rc = prof_drain(h);
  const std::vector<float>& v = h->prof_samples[kernel];
  const int64_t n = (int64_t)v.size();
  for (int64_t i = 0; i < n && i < cap; ++i) ms_out[i] = (double)v[(size_t)i];
  if (n_out) *n_out = n;
  return drain_rc;
}

int rcg_last_launch(const rcg_handle* h, int32_t kind, int32_t* kernel_id, int32_t* variant, int32_t* envs_per_wave) {
  if (!h || kind < 0 || kind >= RCG_KERNEL_COUNT_) return RCG_ERR_BAD_ARG;
  if (kernel_id) *kernel_id = h->last[kind].kernel_id;
  if (variant) *variant = h->last[kind].variant;
  if (envs_per_wave) *envs_per_wave = h->last[kind].envs_per_wave;
  return RCG_OK;
}

const char* rcg_kernel_name(int32_t kernel_id) {
  static const char* const names[RCG_KID_COUNT_] = {"none",        "k_actor",   "k_actor_dma", "k_ticks",    "k_actor_opt",
                                                    "k_nominal",   "k_sim",     "k_sim_v",     "k_sim_dist", "k_critic_fit",
                                                    "k_actor_dma_packed", "k_actor_search"};
  return (kernel_id >= 0 && kernel_id < RCG_KID_COUNT_) ? names[kernel_id] : "?";
}

}  // extern "C"
