// rcg_critic_fit_ml.hpp - k_critic_fit_ml: the critic update with FOUR LANES PER ENV
// (replacement of CtrlOptPred._critic_optimizer, rcognita/controllers.py:1248-1271; same build-defined fit as
// rcg_critic_fit.hpp / oracle/rcg_oracle.py::critic_fit_single: the unique minimiser of the w_init-regularised bounded
// least squares, by the primal active-set walk).
//
// Why: k_critic_fit (lane == env) lasts as long as the longest walk of the batch, one iteration being ~680 mostly dependent
// VALU instructions of a lone wave.  Here the VARIABLES of an env are dealt over the four lanes of a quad
// (variable i lives in lane i mod 4, slot i div 4), so the per-variable work of an iteration - restricted columns, z, the
// ratio test, the step, the multipliers - is a quarter as long per lane, the m x m system is formed by quad sums
// ((p0 + p1) + (p2 + p3) through two DPP quad permutes: every lane of the quad ends with the same bits) and solved
// redundantly, and the decisions (first bound hit, most wrong-signed multiplier) are quad reductions with the walk's
// tie rules (smallest ratio / largest score, then lowest index).  A wave holds 16 envs instead of 64: four times the
// waves, each a quarter as long and with the maximum walk of 16 envs instead of 64.
// The TD stack (A, b) is built by critic_prologue exactly as for k_critic_fit - all four lanes run it, lane 0 of the quad
// stores - so the problem instance is the same bits; what differs is the ASSOCIATION of the sums over variables: the weights
// agree with k_critic_fit and with the float64 oracle to the tolerances of the parity tests on the same inputs (the whole
// critic test set passes with this form forced for every structure: 324 tests, --rcg-lib + RCG_FIT_LANES=4), not bit for
// bit; on rank-deficient stacks a last-bit difference can end the walk on another vertex of equal cost, which 40 free-running
// ticks amplify (profiles/r04_fit_four_lanes.txt).
// Where it runs: the structures with >= 20 weights (rcg_sysops.hpp::launch_fit3) - there the one-lane walk, up to 3 dc + 10
// iterations of ~2000 instructions for the slowest lane of a wave, takes 2.3 ms for 32 768 envs and this form 0.41 ms; with
// few weights the redundant solve and the quad reductions make it the slower form (configs[2], 6 weights: 65 -> 88 us).
// Rows: m <= 3 (the exact-m instance; every preset).
#pragma once
#include "rcg_critic_fit.hpp"

namespace rcg {

constexpr int FIT_L = 4;  // lanes per env

__device__ __forceinline__ double quad_xchg(double x, int ctrl) {  // ctrl 0xB1: lanes 0<->1, 2<->3; 0x4E: 0<->2, 1<->3
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  unsigned lo, hi;
  if (ctrl == 0xB1) {
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0xB1, 0xF, 0xF, false);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0xB1, 0xF, 0xF, false);
  } else {
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x4E, 0xF, 0xF, false);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x4E, 0xF, 0xF, false);
  }
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int quad_xchg(int x, int ctrl) {
  return ctrl == 0xB1 ? __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false)
                      : __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
}
// (p0 + p1) + (p2 + p3), the same bits in the four lanes (addition commutes exactly)
__device__ __forceinline__ double quad_sum(double x) {
  x += quad_xchg(x, 0xB1);
  return x + quad_xchg(x, 0x4E);
}

template <typename Sys, typename real, int CS, int MAXM>
__device__ __forceinline__ void critic_update_env_ml(const FitArgs<real>& F, const KParams<double>& P,
                                                     const KParams<real>& Pr, const long b, const int s) {
  constexpr int DC = CriticDim<CS, Sys::DS, Sys::DU>::value;
  constexpr int NS = (DC + FIT_L - 1) / FIT_L;  // slots per lane
  const long B = P.B;
  const int m = P.n_critic - 1;
  // my variables: i = k * FIT_L + s (a slot beyond dc: a variable fixed at 0 with a zero column - it never moves).  Every lane
  // runs the prologue (the env step, the push, b: the same bits as k_critic_fit's) but keeps only ITS columns of the regressor
  // rows - round 6: with the whole 3 x 35 stack and three 35-entry box arrays per lane the kernel needed 256 VGPRs + 134 AGPRs
  // (one wave per SIMD); the arithmetic is unchanged
  double A[MAXM][NS], bv[MAXM], w0[NS], lo[NS], hi[NS];
  unsigned own = 0u;  // slots that hold a variable
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    if (k * FIT_L + s < DC) own |= 1u << k;
#pragma unroll
    for (int r = 0; r < MAXM; ++r) A[r][k] = 0.0;
  }
  const bool fit = critic_prologue_rows<Sys, real, CS, MAXM>(F, P, Pr, b, s == 0, bv, [&](int r, const double (&phi)[DC]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double v = 0.0;
#pragma unroll
      for (int t = 0; t < FIT_L; ++t)
        if (k * FIT_L + t < DC) v = s == t ? phi[k * FIT_L + t] : v;
      A[r][k] = v;
    }
  });
  if (!fit) return;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const bool mine = (own >> k) & 1u;
    const int i = mine ? k * FIT_L + s : 0;
    const double v0 = F.wcfg[i], vl = F.wcfg[40 + i], vh = F.wcfg[80 + i];
    w0[k] = mine ? v0 : 0.0;
    lo[k] = mine ? vl : 0.0;
    hi[k] = mine ? vh : 0.0;
  }

  double trp = 0.0;
#pragma unroll
  for (int r = 0; r < MAXM; ++r)
#pragma unroll
    for (int k = 0; k < NS; ++k) trp = fma_r(A[r][k], A[r][k], trp);  // rows >= m and empty slots are zero
  double mu = FIT_MU_REL * (quad_sum(trp) / (double)m);
  if (!(mu > 1e-30)) mu = 1e-30;

  double w[NS], z[NS];
  unsigned fm = 0u, at_hi = 0u, blocked = 0u;  // free / fixed-at-upper / not-to-release masks over my slots
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    w[k] = w0[k] < lo[k] ? lo[k] : (w0[k] > hi[k] ? hi[k] : w0[k]);
    z[k] = w[k];
    if ((own >> k) & 1u) {
      if (w[k] > lo[k] && w[k] < hi[k])
        fm |= 1u << k;
      else if (w[k] >= hi[k])
        at_hi |= 1u << k;
    }
  }
  int last_freed = -1;  // global variable index (the same in the four lanes)

  for (int it = 0; it < fit_max_iters(DC); ++it) {
    double L[MAXM][MAXM], lam[MAXM];
#pragma unroll
    for (int r = 0; r < MAXM; ++r) {
      double sp = 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k) sp = fma_r(-A[r][k], ((fm >> k) & 1u) ? w0[k] : w[k], sp);
      lam[r] = bv[r] + quad_sum(sp);
#pragma unroll
      for (int q = 0; q <= r; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {  // A_F = the free columns (a fixed one counts as a zero column: fma(0, 0, acc) = acc)
          const bool fr = (fm >> k) & 1u;
          acc = fma_r(fr ? A[r][k] : 0.0, fr ? A[q][k] : 0.0, acc);
        }
        L[r][q] = quad_sum(acc) + (r == q ? mu : 0.0);
      }
    }
    // root-free Cholesky M = L D L^T, as rcg_critic_fit.hpp (the four lanes hold the same M: the same solve four times)
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
        double sv = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sv -= (L[i][k] * L[j][k]) * dg[k];
        L[i][j] = sv * rc[j];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {
      double sv = lam[i];
#pragma unroll
      for (int k = 0; k < i; ++k) sv -= L[i][k] * lam[k];
      lam[i] = sv;
    }
#pragma unroll
    for (int i = 0; i < MAXM; ++i) lam[i] = lam[i] * rc[i];
#pragma unroll
    for (int i = MAXM - 1; i >= 0; --i) {
      double sv = lam[i];
#pragma unroll
      for (int k = i + 1; k < MAXM; ++k) sv -= L[k][i] * lam[k];
      lam[i] = sv;
    }
    // z_F = w0_F + A_F^T lam; ratio test over MY slots in index order, then over the quad (smallest ratio by cross
    // multiplication, ties -> lowest variable index: what the sequential scan of the one-lane walk picks)
    double nb = 2.0, db = 1.0;
    int jmin = -1;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const bool fr = (fm >> k) & 1u;
      double c = 0.0;
#pragma unroll
      for (int r = 0; r < MAXM; ++r) c = fma_r(A[r][k], lam[r], c);
      const double zi = w0[k] + c;
      z[k] = fr ? zi : z[k];
      const bool vlo = zi < lo[k], vhi = zi > hi[k];
      const double ni = fabs((vlo ? lo[k] : hi[k]) - w[k]), di = fabs(zi - w[k]);
      if (fr && (vlo || vhi) && ni * db < nb * di) {
        nb = ni;
        db = di;
        jmin = k * FIT_L + s;
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int ctrl = st == 0 ? 0xB1 : 0x4E;
      const double on = quad_xchg(nb, ctrl), od = quad_xchg(db, ctrl);
      const int oj = quad_xchg(jmin, ctrl);
      const double l = on * db, r = nb * od;  // other < mine  <=>  on / od < nb / db
      // total order: whenever neither ratio is strictly smaller - equal, or a NaN product from non-finite
      // buffers (0 * inf) - the lower index wins in BOTH partners, so the quad always agrees on (nb, db, jmin)
      const bool take = oj >= 0 && (jmin < 0 || l < r || (!(r < l) && oj < jmin));
      nb = take ? on : nb;
      db = take ? od : db;
      jmin = take ? oj : jmin;
    }
    if (jmin >= 0) {  // move towards z until the first bound, fix that variable (its owner does)
      double alpha = nb / db;
      if (!(alpha > 0.0)) alpha = 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const bool fr = (fm >> k) & 1u;
        double v = w[k] + alpha * (z[k] - w[k]);
        v = v < lo[k] ? lo[k] : (v > hi[k] ? hi[k] : v);
        const bool up = z[k] > hi[k];
        if (k * FIT_L + s == jmin) {
          v = up ? hi[k] : lo[k];
          at_hi = up ? (at_hi | (1u << k)) : (at_hi & ~(1u << k));
        }
        w[k] = fr ? v : w[k];
        if (k * FIT_L + s == jmin) fm &= ~(1u << k);
      }
      if (alpha > 0.0)
        blocked = 0u;
      else if (jmin == last_freed && (jmin % FIT_L) == s)
        blocked |= 1u << (jmin / FIT_L);
      last_freed = -1;
      continue;
    }
    double res[MAXM];
#pragma unroll
    for (int k = 0; k < NS; ++k) w[k] = ((fm >> k) & 1u) ? z[k] : w[k];
#pragma unroll
    for (int r = 0; r < MAXM; ++r) {
      double sp = 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k) sp = fma_r(A[r][k], w[k], sp);
      res[r] = quad_sum(sp) - bv[r];
    }
    int best = -1;
    double best_score = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double g = mu * (w[k] - w0[k]);
      double scale = fabs(g);
#pragma unroll
      for (int r = 0; r < MAXM; ++r) {
        const double t = A[r][k] * res[r];
        g += t;
        scale += fabs(t);
      }
      const double score = ((at_hi >> k) & 1u) ? g : -g;
      if (((own >> k) & 1u) && !(((fm | blocked) >> k) & 1u) && score > FIT_KKT_TOL * scale && score > best_score) {
        best = k * FIT_L + s;
        best_score = score;
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // largest score, ties -> lowest index (the sequential scan keeps the first maximum)
      const int ctrl = st == 0 ? 0xB1 : 0x4E;
      const double os = quad_xchg(best_score, ctrl);
      const int ob = quad_xchg(best, ctrl);
      const bool take = ob >= 0 && (best < 0 || os > best_score || (!(best_score > os) && ob < best));  // total, as above
      best_score = take ? os : best_score;
      best = take ? ob : best;
    }
    if (best < 0) break;
    if ((best % FIT_L) == s) fm |= 1u << (best / FIT_L);
    last_freed = best;
  }

  // safeguard (non-finite buffers): keep the start point unless Jc(w) <= Jc(w_init)
  double Pw = 0.0, P0 = 0.0;
#pragma unroll
  for (int r = 0; r < MAXM; ++r) {
    double sp = 0.0, sp0 = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double wi = w0[k] < lo[k] ? lo[k] : (w0[k] > hi[k] ? hi[k] : w0[k]);
      sp = fma_r(A[r][k], w[k], sp);
      sp0 = fma_r(A[r][k], wi, sp0);
    }
    const double sr = quad_sum(sp) - bv[r], sr0 = quad_sum(sp0) - bv[r];
    Pw = fma_r(sr, sr, Pw);
    P0 = fma_r(sr0, sr0, P0);
  }
  const bool keep = Pw <= P0;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    if ((own >> k) & 1u) {
      const int i = k * FIT_L + s;
      const double wi = w0[k] < lo[k] ? lo[k] : (w0[k] > hi[k] ? hi[k] : w0[k]);
      const double v = keep ? w[k] : wi;
      F.w_critic[(long)i * B + b] = (real)v;
      F.w_prev[(long)i * B + b] = (real)v;  // w_critic_prev = w_critic (controllers.py:1471)
    }
  }
}

// one quad per env: blocks of 64 threads = 16 envs
template <typename Sys, typename real, int CS, int MAXM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_critic_fit_ml(const FitArgs<real> F, const KParams<double> P, const KParams<real> Pr) {
  const long b = F.env_lo + (long)blockIdx.x * (blockDim.x / FIT_L) + (threadIdx.x / FIT_L);
  if (b >= (F.env_hi > 0 ? (long)F.env_hi : P.B)) return;  // (whole quads leave together: the DPP exchanges below stay inside a quad)
  critic_update_env_ml<Sys, real, CS, MAXM>(F, P, Pr, b, (int)(threadIdx.x & (FIT_L - 1)));
}

}  // namespace rcg
