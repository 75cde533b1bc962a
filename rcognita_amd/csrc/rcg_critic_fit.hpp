// rcg_critic_fit.hpp - k_critic_fit: replacement of CtrlOptPred._critic_optimizer
// (rcognita/controllers.py:1248-1271) for a batch of envs, lane == env.
//
// Problem per env (controllers.py:1216-1245 written as a linear least squares, see
// oracle/rcg_oracle.py::critic_td_system):   Jc(w) = 1/2 |A w - b|^2,  Wmin <= w <= Wmax,
//   row r (= the reference's term k = r + 1):  A[r] = phi(y_r, u_r),
//   b[r] = gamma * w_prev . phi(y_{r+1}, u_{r+1}) + rho(y_r, u_r),   r = 0 .. Ncritic - 2,
// on the OLDEST Ncritic buffer rows.  The reference runs SLSQP from w_init = ones; its iterates are
// path dependent, so the build defines the fit as the unique minimiser of
//   1/2 |A w - b|^2 + mu/2 |w - w_init|^2   in the box,  mu = 1e-8 * trace(A A^T) / m,
// computed by a primal active-set method (bounded-variable least squares): each iteration solves the
// problem on the free variables exactly through the m x m system (A_F A_F^T + mu I) lam = rhs
// (root-free Cholesky L D L^T), then either moves towards that point until the first bound is hit and fixes that
// variable, or accepts it and releases the bound variable with the most wrong-signed multiplier
// (none: optimal).  Feasible and monotone, so the result is never worse than w_init.  This mirrors
// oracle/rcg_oracle.py::critic_fit_single statement by statement; all arithmetic is float64 whatever
// the handle's dtype (the systems are tiny - m <= 8, dc <= 35 - and badly scaled).  The active set is
// held in 64-bit lane-private masks and every loop over variables is unrolled with a predicate, so no
// array is indexed dynamically.
//
// Cost (profiles/r02_*; tools/critic_fit_probe.py): 10 us per launch at B = 131072 while the oldest buffer rows are still
// zeros, 56-62 us in the steady state of an RQL closed loop (quadratic critic, m = 3): ~7000 VALU instructions per wave of
// which 2100 are f64 fma, 40 % of the issue slots - a wave runs as many active-set iterations as its slowest lane (11-13
// where the mean over envs is 2.2).  Tried in round 2: root-free L D L^T instead of L L^T (kept: 3 reciprocals instead
// of 12 divides / square roots per solve, 3 % faster); warm-starting the walk from the previous fit (not adopted: on
// rank-deficient stacks the end point depends on the start); running the NEXT tick's fit on a second stream while the
// actor kernel runs - legal because the TD stack reads only the oldest rows - which hid the fit but slowed the actor
// kernel by almost as much (configs[2] RQL tick 0.380 -> 0.371 ms generated, 0.549 -> 0.547 ms streamed): dropped.
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

constexpr double FIT_MU_REL = 1e-8;
constexpr double FIT_KKT_TOL = 1e-10;
__host__ __device__ constexpr int fit_max_iters(int dc) { return 3 * dc + 10; }

template <int CS, int DS, int DU>
struct CriticDim {
  static constexpr int n = DS + DU;
  static constexpr int value = CS == RCG_CRITIC_QUAD_LIN    ? n * (n + 1) / 2 + n
                               : CS == RCG_CRITIC_QUADRATIC ? n * (n + 1) / 2
                               : CS == RCG_CRITIC_QUAD_NOMIX ? n
                                                             : DS + DS * DU + DU;
};

// regressor of _critic (controllers.py:1204-1212) with compile-time structure
template <int CS, int DS, int DU>
__device__ __forceinline__ void critic_phi(const double* chi, const double* y, const double* u, double* phi) {
  constexpr int N = DS + DU;
  int idx = 0;
  if (CS == RCG_CRITIC_QUAD_LIN || CS == RCG_CRITIC_QUADRATIC) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = i; j < N; ++j) phi[idx++] = chi[i] * chi[j];
    if (CS == RCG_CRITIC_QUAD_LIN) {
#pragma unroll
      for (int i = 0; i < N; ++i) phi[idx++] = chi[i];
    }
  } else if (CS == RCG_CRITIC_QUAD_NOMIX) {
#pragma unroll
    for (int i = 0; i < N; ++i) phi[i] = chi[i] * chi[i];
  } else {
#pragma unroll
    for (int i = 0; i < DS; ++i) phi[idx++] = y[i] * y[i];
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int c = 0; c < DU; ++c) phi[idx++] = y[i] * u[c];
#pragma unroll
    for (int c = 0; c < DU; ++c) phi[idx++] = u[c] * u[c];
  }
}

template <typename real>
struct FitArgs {
  real* w_critic;         // [dc][B] out
  real* w_prev;           // [dc][B] in (TD target weights), out (:= fitted w)
  real* obs_buf;          // [buffer_size][dy][B]
  real* act_buf;          // [buffer_size][du][B]
  const double* wcfg;     // [3][40]: w_init, w_min, w_max
  // Everything an RQL / SQL tick does between two decisions is lane == env work on the same few numbers, so
  // rcg_control_tick issues it as ONE launch (three dependent launches cost ~7 us of dispatch gap each at this size,
  // plus the former push kernel's own 11 us): do_sim: Simulator.sim_step (env_substeps, the code k_sim runs); do_push:
  // push_vec of (action_curr, obs) (utilities.py:78-79); do_fit: the fit.  Same arithmetic as the separate launches.
  int do_sim, do_push, do_fit;
  SimArgs<real> sim;      // do_sim
  const real* state;      // [ds][B] do_push: the observation to push (= sim.state)
  const real* action;     // [du][B] do_push: the held action (action_curr)
  int env_lo, env_hi;     // the envs [env_lo, env_hi) this launch serves (env_hi == 0: the whole batch)
  // k_ticks_mem only (0 elsewhere): the two buffers as RINGS while the launch runs - ring = 1 + the physical row this push overwrites
  // (the oldest); logical row r after the push is physical row (ring + r) mod buffer_size.  The physical shift of a push is two
  // dependent load -> store passes over the rows (buffer_size 10: ~3 us of a 20 us tick on a wave that runs alone on its SIMD);
  // the launch rotates the rows back into place after its last tick (critic_ring_unrotate), so every launch starts and ends canonical.
  int ring;
};

// The part of an env's critic update that precedes the solver: [env step] -> [push] -> TD stack (A, b) and the box.  `store`:
// this lane writes the env's state / buffers back (false for the lanes that only help with the env's solve, k_critic_fit_ml).
// Returns false when there is nothing to fit (F.do_fit == 0).
// (`row(r, phi)` receives the regressor of TD row r, r < m: the one-lane kernels keep all of it - critic_prologue below -, the
// four-lane kernel only the columns of the calling lane, so that no lane ever holds the whole m x dc stack in registers)
template <typename Sys, typename real, int CS, int MAXM, typename RowSink>
__device__ __forceinline__ bool critic_prologue_rows(const FitArgs<real>& F, const KParams<double>& P, const KParams<real>& Pr,
                                                     const long b, const bool store, double (&bv)[MAXM], RowSink row) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, DC = CriticDim<CS, DS, DU>::value;
  // rows of the shifted buffers that stay in registers for the TD stack: new row r = old row r + 1, r = 0 .. KEEP - 1
  constexpr int KEEP = MAXM + 1 <= 4 ? MAXM + 1 : 4;
  const long B = P.B;
  const int m = P.n_critic - 1;  // rows of the TD stack, 1 <= m <= MAXM (checked on the host)
  const int bs = Pr.buffer_size;

  // ---- every load the prologue needs is requested here, before any arithmetic: the env step, the shift of the two
  // buffers and the TD stack used to be three dependent round trips to memory (and the row-by-row shift one per row:
  // the compiler cannot prove that the store to row r leaves row r + 2 alone) ----------------------------------------
  real wpr[DC];
  if (F.do_fit) {
#pragma unroll
    for (int i = 0; i < DC; ++i) wpr[i] = F.w_prev[(long)i * B + b];
  }
  real ko[KEEP][DS], ka[KEEP][DU];  // do_push: old rows 1 .. KEEP (= new rows 0 .. KEEP - 1); else rows 0 .. KEEP - 1
  const int koff = F.do_push ? 1 : 0;
  const int ring = F.ring;  // > 0: ring mode (with do_push)
  // physical row of the row that is logical row r once this call's push is done
  auto phys = [&](int r) -> int {
    if (!ring) return r;
    const int p = ring + r;
    return p >= bs ? p - bs : p;
  };
#pragma unroll
  for (int k = 0; k < KEEP; ++k)
    if (k + koff < bs) {
      const int pr = ring ? phys(k) : k + koff;
#pragma unroll
      for (int c = 0; c < DS; ++c) ko[k][c] = F.obs_buf[((long)pr * DS + c) * B + b];
#pragma unroll
      for (int c = 0; c < DU; ++c) ka[k][c] = F.act_buf[((long)pr * DU + c) * B + b];
    }

  if (F.do_sim || F.do_push) {
    real xs[DS], xp[DS], ua[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) xp[c] = xs[c] = F.state[(long)c * B + b];
#pragma unroll
    for (int c = 0; c < DU; ++c) ua[c] = F.action[(long)c * B + b];
    if (F.do_sim) {  // k_sim
      uint32_t st = F.sim.status[b];
      if (!(st & 1u)) {
        const auto pre = load_pre<Sys, real>(Pr, F.sim.pars_env, b);
        real accum = Pr.accum_every_substep ? F.sim.accum[b] : (real)0;
        const bool tgt = Pr.has_target != 0;
        const bool ok = tgt ? env_substeps<Sys, real, true>(Pr, pre, F.sim.n_sub, xs, xp, ua, st, accum)
                            : env_substeps<Sys, real, false>(Pr, pre, F.sim.n_sub, xs, xp, ua, st, accum);
        if (!ok) {
          if (store) F.sim.status[b] = st;  // became non-finite: frozen at its last finite state
        } else {
#pragma unroll
          for (int c = 0; c < DS; ++c) {
            if (store) F.sim.state[(long)c * B + b] = xs[c];
            if (store) F.sim.state_prev[(long)c * B + b] = xp[c];
          }
          if (store) if (Pr.accum_every_substep) F.sim.accum[b] = accum;
        }
      }
    }
    if (F.do_push && ring) {  // ring mode: the new row takes the oldest row's place, nothing else moves
#pragma unroll
      for (int c = 0; c < DS; ++c) if (store) F.obs_buf[((long)(ring - 1) * DS + c) * B + b] = xs[c];
#pragma unroll
      for (int c = 0; c < DU; ++c) if (store) F.act_buf[((long)(ring - 1) * DU + c) * B + b] = ua[c];
#pragma unroll
      for (int k = 0; k < KEEP; ++k)
        if (k == bs - 1) {
#pragma unroll
          for (int c = 0; c < DS; ++c) ko[k][c] = xs[c];
#pragma unroll
          for (int c = 0; c < DU; ++c) ka[k][c] = ua[c];
        }
    } else if (F.do_push) {  // push_vec on both buffers: drop row 0, append (obs, action_curr) at the bottom (utilities.py:78-79)
#pragma unroll
      for (int k = 0; k < KEEP; ++k)
        if (k + 1 < bs) {
#pragma unroll
          for (int c = 0; c < DS; ++c) if (store) F.obs_buf[((long)k * DS + c) * B + b] = ko[k][c];
#pragma unroll
          for (int c = 0; c < DU; ++c) if (store) F.act_buf[((long)k * DU + c) * B + b] = ka[k][c];
        }
      // the rest of the shift four rows at a time, all loads of a pass before its stores
      constexpr int CH = 4;
      for (int r0 = KEEP; r0 < bs - 1; r0 += CH) {
        real ro[CH][DS], ra[CH][DU];
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (r0 + k < bs - 1) {
#pragma unroll
            for (int c = 0; c < DS; ++c) ro[k][c] = F.obs_buf[((long)(r0 + k + 1) * DS + c) * B + b];
#pragma unroll
            for (int c = 0; c < DU; ++c) ra[k][c] = F.act_buf[((long)(r0 + k + 1) * DU + c) * B + b];
          }
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (r0 + k < bs - 1) {
#pragma unroll
            for (int c = 0; c < DS; ++c) if (store) F.obs_buf[((long)(r0 + k) * DS + c) * B + b] = ro[k][c];
#pragma unroll
            for (int c = 0; c < DU; ++c) if (store) F.act_buf[((long)(r0 + k) * DU + c) * B + b] = ra[k][c];
          }
      }
#pragma unroll
      for (int c = 0; c < DS; ++c) if (store) F.obs_buf[((long)(bs - 1) * DS + c) * B + b] = xs[c];
#pragma unroll
      for (int c = 0; c < DU; ++c) if (store) F.act_buf[((long)(bs - 1) * DU + c) * B + b] = ua[c];
#pragma unroll
      for (int k = 0; k < KEEP; ++k)
        if (k == bs - 1) {  // (buffer_size <= KEEP: the row just pushed is one of the kept rows)
#pragma unroll
          for (int c = 0; c < DS; ++c) ko[k][c] = xs[c];
#pragma unroll
          for (int c = 0; c < DU; ++c) ka[k][c] = ua[c];
        }
    }
    if (!F.do_fit) return false;
  }

#pragma unroll
  for (int r = 0; r < MAXM; ++r) bv[r] = 0.0;
  // ---- build A, b from buffer rows 0 .. m (the oldest rows, controllers.py:1231-1234) ----------
#pragma unroll
  for (int r = 0; r <= MAXM; ++r) {
    if (r <= m) {
      double y[DS], u[DU], chi[NCHI], phi[DC];
      if (r < KEEP) {
#pragma unroll
        for (int c = 0; c < DS; ++c) y[c] = (double)ko[r][c];
#pragma unroll
        for (int c = 0; c < DU; ++c) u[c] = (double)ka[r][c];
      } else {  // m > 3: beyond the kept rows (written above by this lane: same-address order, served by L2)
        const int pr = phys(r);
#pragma unroll
        for (int c = 0; c < DS; ++c) y[c] = (double)F.obs_buf[((long)pr * DS + c) * B + b];
#pragma unroll
        for (int c = 0; c < DU; ++c) u[c] = (double)F.act_buf[((long)pr * DU + c) * B + b];
      }
      if (P.has_target)
        make_chi<DS, DU, true, double>(P, y, u, chi);
      else
        make_chi<DS, DU, false, double>(P, y, u, chi);
      critic_phi<CS, DS, DU>(chi, y, u, phi);
      if (r > 0) {  // gamma * w_prev . phi(row r) belongs to TD row r - 1
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < DC; ++i) q = fma_r((double)wpr[i], phi[i], q);
        bv[r - 1 < MAXM ? r - 1 : 0] += P.gamma * q;
      }
      if (r < m && r < MAXM) {
        row(r, phi);
        bv[r] += stage_any<NCHI, double>(P, chi);
      }
    }
  }

  return true;
}

// the whole stack A [m][dc] (rows >= m zero), b, and the box: what the one-lane walks work on
template <typename Sys, typename real, int CS, int MAXM>
__device__ __forceinline__ bool critic_prologue(const FitArgs<real>& F, const KParams<double>& P, const KParams<real>& Pr,
                                                const long b, const bool store,
                                                double (&A)[MAXM][CriticDim<CS, Sys::DS, Sys::DU>::value], double (&bv)[MAXM],
                                                double (&w0)[CriticDim<CS, Sys::DS, Sys::DU>::value],
                                                double (&lo)[CriticDim<CS, Sys::DS, Sys::DU>::value],
                                                double (&hi)[CriticDim<CS, Sys::DS, Sys::DU>::value]) {
  constexpr int DC = CriticDim<CS, Sys::DS, Sys::DU>::value;
#pragma unroll
  for (int r = 0; r < MAXM; ++r)
#pragma unroll
    for (int i = 0; i < DC; ++i) A[r][i] = 0.0;
  const bool fit = critic_prologue_rows<Sys, real, CS, MAXM>(F, P, Pr, b, store, bv, [&](int r, const double (&phi)[DC]) {
#pragma unroll
    for (int i = 0; i < DC; ++i) A[r][i] = phi[i];
  });
  if (!fit) return false;
#pragma unroll
  for (int i = 0; i < DC; ++i) {
    w0[i] = F.wcfg[i];
    lo[i] = F.wcfg[40 + i];
    hi[i] = F.wcfg[80 + i];
  }
  return true;
}

// After `pushes` ring-mode pushes physical row p of env b holds logical row (p - pushes) mod buffer_size: rotate both buffers left by
// pushes mod buffer_size, in place (cycle by cycle; all components of a row travel together), by the lane that stored them.
template <typename Sys, typename real>
__device__ __forceinline__ void critic_ring_unrotate(const FitArgs<real>& F, const KParams<real>& Pr, const long b, const int pushes) {
  constexpr int DS = Sys::DS, DU = Sys::DU;
  const long B = Pr.B;
  const int bs = Pr.buffer_size, sh = pushes % bs;
  if (sh == 0) return;
  int g = bs, r = sh;  // gcd(bs, sh) = the number of cycles
  while (r) {
    const int q = g % r;
    g = r;
    r = q;
  }
  for (int s = 0; s < g; ++s) {
    real to[DS], ta[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) to[c] = F.obs_buf[((long)s * DS + c) * B + b];
#pragma unroll
    for (int c = 0; c < DU; ++c) ta[c] = F.act_buf[((long)s * DU + c) * B + b];
    int j = s;
    for (;;) {
      int nx = j + sh;
      if (nx >= bs) nx -= bs;
      if (nx == s) break;
      real vo[DS], va[DU];
#pragma unroll
      for (int c = 0; c < DS; ++c) vo[c] = F.obs_buf[((long)nx * DS + c) * B + b];
#pragma unroll
      for (int c = 0; c < DU; ++c) va[c] = F.act_buf[((long)nx * DU + c) * B + b];
#pragma unroll
      for (int c = 0; c < DS; ++c) F.obs_buf[((long)j * DS + c) * B + b] = vo[c];
#pragma unroll
      for (int c = 0; c < DU; ++c) F.act_buf[((long)j * DU + c) * B + b] = va[c];
      j = nx;
    }
#pragma unroll
    for (int c = 0; c < DS; ++c) F.obs_buf[((long)j * DS + c) * B + b] = to[c];
#pragma unroll
    for (int c = 0; c < DU; ++c) F.act_buf[((long)j * DU + c) * B + b] = ta[c];
  }
}

// Everything of one env, lane-private: [env step] -> [push] -> [fit], in and out through the handle's tensors.  The body of
// k_critic_fit (lane == env) and of the critic phase of k_ticks_mem (rcg_ticks.hpp: the lanes that stand for the wave's envs).
template <typename Sys, typename real, int CS, int MAXM>
__device__ __forceinline__ void critic_update_env(const FitArgs<real>& F, const KParams<double>& P, const KParams<real>& Pr,
                                                  const long b) {
  constexpr int DC = CriticDim<CS, Sys::DS, Sys::DU>::value;
  const long B = P.B;
  const int m = P.n_critic - 1;  // rows of the TD stack, 1 <= m <= MAXM (checked on the host)
  double A[MAXM][DC], bv[MAXM], w0[DC], lo[DC], hi[DC];
  if (!critic_prologue<Sys, real, CS, MAXM>(F, P, Pr, b, true, A, bv, w0, lo, hi)) return;

  double tr = 0.0;
#pragma unroll
  for (int r = 0; r < MAXM; ++r)
#pragma unroll
    for (int i = 0; i < DC; ++i) tr = fma_r(A[r][i], A[r][i], tr);  // rows >= m are zero
  double mu = FIT_MU_REL * (tr / (double)m);
  if (!(mu > 1e-30)) mu = 1e-30;

  double w[DC], z[DC];
  unsigned long long fm = 0ull, at_hi = 0ull, blocked = 0ull;  // free / fixed-at-upper / not-to-release masks
#pragma unroll
  for (int i = 0; i < DC; ++i) {
    w[i] = w0[i] < lo[i] ? lo[i] : (w0[i] > hi[i] ? hi[i] : w0[i]);
    z[i] = w[i];
    if (w[i] > lo[i] && w[i] < hi[i])
      fm |= 1ull << i;
    else if (w[i] >= hi[i])
      at_hi |= 1ull << i;
  }
  int last_freed = -1;

  // Cost model, measured at B = 131072 (2tank, quadratic critic, steady state of an RQL loop, tools/critic_fit_probe.py
  // with the walk capped at c iterations): 14 us + 3.6 us x c up to c = 8, 66 us uncapped (the longest walk of the batch):
  // with B / 64 = 2 waves per SIMD the kernel lasts as long as its slowest wave, and a wave runs as many iterations as
  // its slowest lane.  Per-variable work is straight-line code on selects and the ratio test divides once; one
  // iteration is ~680 VALU instructions of which 250 are float64 arithmetic and 160 are the halves of 64-bit selects.
  // Re-packing unfinished walks into dense waves between levels of iterations (LDS hand-over, blocks of 256 envs) was
  // built and measured in round 2: 34 % fewer instructions per wave, bit-identical weights, the same duration - dropped.
  for (int it = 0; it < fit_max_iters(DC); ++it) {
    // rhs = b - A_B w_B - A_F w0_F,  M = A_F A_F^T + mu I  (rows >= m: M = mu I, rhs = 0 -> lam = 0)
    double L[MAXM][MAXM], lam[MAXM];
    // A restricted to the free columns (bound columns zero: adding an exact zero leaves the sums' bits unchanged) and
    // the point the bound part of the right-hand side is taken at, once per iteration instead of a select per term
    double Af[MAXM][DC], wb[DC];
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      const bool fr = (fm >> i) & 1ull;
      wb[i] = fr ? w0[i] : w[i];
#pragma unroll
      for (int r = 0; r < MAXM; ++r) Af[r][i] = fr ? A[r][i] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < MAXM; ++r) {
      double s = bv[r];
#pragma unroll
      for (int i = 0; i < DC; ++i) s = fma_r(-A[r][i], wb[i], s);
      lam[r] = s;
#pragma unroll
      for (int q = 0; q <= r; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < DC; ++i) acc = fma_r(Af[r][i], Af[q][i], acc);
        L[r][q] = acc + (r == q ? mu : 0.0);
      }
    }
    // root-free Cholesky M = L D L^T (unit lower L): one reciprocal per pivot, no sqrt, no other division - the kernel is
    // bound by f64 instruction issue and a divide / square root is 10-20 instructions (round 1: L L^T, 3 m of each)
    const double floor_piv = mu * 1e-6;
    double dg[MAXM], rc[MAXM];
#pragma unroll
    for (int j = 0; j < MAXM; ++j) {
      double dj = L[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) dj -= (L[j][k] * L[j][k]) * dg[k];
      if (!(dj > floor_piv)) dj = floor_piv;
      dg[j] = dj;
      rc[j] = 1.0 / dj;
#pragma unroll
      for (int i = j + 1; i < MAXM; ++i) {
        double s = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= (L[i][k] * L[j][k]) * dg[k];
        L[i][j] = s * rc[j];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {  // forward: L y = rhs
      double s = lam[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= L[i][k] * lam[k];
      lam[i] = s;
    }
#pragma unroll
    for (int i = 0; i < MAXM; ++i) lam[i] = lam[i] * rc[i];  // D z = y
#pragma unroll
    for (int i = MAXM - 1; i >= 0; --i) {  // backward: L^T lam = z
      double s = lam[i];
#pragma unroll
      for (int k = i + 1; k < MAXM; ++k) s -= L[k][i] * lam[k];
      lam[i] = s;
    }
    // z_F = w0_F + A_F^T lam and the ratio test towards it: the step to the bound z_i crosses is a_i = n_i / d_i with
    // n_i = |bound_i - w_i| <= d_i = |z_i - w_i|; the smallest a_i (first index on ties) is found by comparing
    // n_i d_best with n_best d_i - no division - and alpha = n_best / d_best is the one division of the iteration
    double nb = 2.0, db = 1.0;
    int jmin = -1;
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      const bool fr = (fm >> i) & 1ull;
      double c = 0.0;
#pragma unroll
      for (int r = 0; r < MAXM; ++r) c = fma_r(A[r][i], lam[r], c);
      const double zi = w0[i] + c;
      z[i] = fr ? zi : z[i];
      const bool vlo = zi < lo[i], vhi = zi > hi[i];
      const double ni = fabs((vlo ? lo[i] : hi[i]) - w[i]), di = fabs(zi - w[i]);
      if (fr && (vlo || vhi) && ni * db < nb * di) {
        nb = ni;
        db = di;
        jmin = i;
      }
    }
    if (jmin >= 0) {  // move towards z until the first bound, fix that variable
      double alpha = nb / db;
      if (!(alpha > 0.0)) alpha = 0.0;
#pragma unroll
      for (int i = 0; i < DC; ++i) {
        const bool fr = (fm >> i) & 1ull;
        double v = w[i] + alpha * (z[i] - w[i]);
        v = v < lo[i] ? lo[i] : (v > hi[i] ? hi[i] : v);
        const bool up = z[i] > hi[i];
        if (i == jmin) {
          v = up ? hi[i] : lo[i];
          at_hi = up ? (at_hi | (1ull << i)) : (at_hi & ~(1ull << i));
        }
        w[i] = fr ? v : w[i];
      }
      fm &= ~(1ull << jmin);
      if (alpha > 0.0)
        blocked = 0ull;
      else if (jmin == last_freed)
        blocked |= 1ull << jmin;
      last_freed = -1;
      continue;
    }
    double res[MAXM];
#pragma unroll
    for (int r = 0; r < MAXM; ++r) res[r] = -bv[r];
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      w[i] = ((fm >> i) & 1ull) ? z[i] : w[i];
#pragma unroll
      for (int r = 0; r < MAXM; ++r) res[r] = fma_r(A[r][i], w[i], res[r]);
    }
    int best = -1;
    double best_score = 0.0;
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      double g = mu * (w[i] - w0[i]);
      double scale = fabs(g);
#pragma unroll
      for (int r = 0; r < MAXM; ++r) {
        const double t = A[r][i] * res[r];
        g += t;
        scale += fabs(t);
      }
      const double score = ((at_hi >> i) & 1ull) ? g : -g;
      if (!(((fm | blocked) >> i) & 1ull) && score > FIT_KKT_TOL * scale && score > best_score) {
        best = i;
        best_score = score;
      }
    }
    if (best < 0) break;
    fm |= 1ull << best;
    last_freed = best;
  }

  // safeguard (non-finite buffers): keep the start point unless Jc(w) <= Jc(w_init)
  double Pw = 0.0, P0 = 0.0;
#pragma unroll
  for (int r = 0; r < MAXM; ++r) {
    double s = -bv[r], s0 = -bv[r];
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      const double wi = w0[i] < lo[i] ? lo[i] : (w0[i] > hi[i] ? hi[i] : w0[i]);
      s = fma_r(A[r][i], w[i], s);
      s0 = fma_r(A[r][i], wi, s0);
    }
    Pw = fma_r(s, s, Pw);
    P0 = fma_r(s0, s0, P0);
  }
  const bool keep = Pw <= P0;
#pragma unroll
  for (int i = 0; i < DC; ++i) {
    const double wi = w0[i] < lo[i] ? lo[i] : (w0[i] > hi[i] ? hi[i] : w0[i]);
    const double v = keep ? w[i] : wi;
    F.w_critic[(long)i * B + b] = (real)v;
    F.w_prev[(long)i * B + b] = (real)v;  // w_critic_prev = w_critic (controllers.py:1471)
  }
}

template <typename Sys, typename real, int CS, int MAXM>
__global__ __launch_bounds__(64) void k_critic_fit(const FitArgs<real> F, const KParams<double> P, const KParams<real> Pr) {
  const long b = F.env_lo + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (F.env_hi > 0 ? (long)F.env_hi : P.B)) return;
  critic_update_env<Sys, real, CS, MAXM>(F, P, Pr, b);
}

}  // namespace rcg
