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
// computed by a semismooth Newton method on the m-dimensional dual with Armijo backtracking, and
// returns the feasible iterate with the smallest Jc (never worse than w_init).  This mirrors
// oracle/rcg_oracle.py::critic_fit_single statement by statement; all arithmetic is float64 whatever
// the handle's dtype (the systems are tiny - m <= 8, dc <= 35 - and badly scaled).
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

constexpr double FIT_MU_REL = 1e-8;
constexpr int FIT_ITERS = 40;
constexpr int FIT_LS = 20;
constexpr double FIT_GTOL = 1e-12;

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
  const real* obs_buf;    // [buffer_size][dy][B]
  const real* act_buf;    // [buffer_size][du][B]
  const double* wcfg;     // [3][40]: w_init, w_min, w_max
};

template <typename Sys, typename real, int CS, int MAXM>
__global__ __launch_bounds__(64) void k_critic_fit(const FitArgs<real> F, const KParams<double> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, DC = CriticDim<CS, DS, DU>::value;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long B = P.B;
  if (b >= B) return;
  const int m = P.n_critic - 1;  // rows of the TD stack, 1 <= m <= MAXM (checked on the host)

  double A[MAXM][DC], bv[MAXM], wp[DC], w0[DC], lo[DC], hi[DC];
#pragma unroll
  for (int i = 0; i < DC; ++i) {
    wp[i] = (double)F.w_prev[(long)i * B + b];
    w0[i] = F.wcfg[i];
    lo[i] = F.wcfg[40 + i];
    hi[i] = F.wcfg[80 + i];
  }
#pragma unroll
  for (int r = 0; r < MAXM; ++r) {
    bv[r] = 0.0;
#pragma unroll
    for (int i = 0; i < DC; ++i) A[r][i] = 0.0;
  }
  // ---- build A, b from buffer rows 0 .. m (the oldest rows, controllers.py:1231-1234) ----------
#pragma unroll
  for (int r = 0; r <= MAXM; ++r) {
    if (r <= m) {
      double y[DS], u[DU], chi[NCHI], phi[DC];
#pragma unroll
      for (int c = 0; c < DS; ++c) y[c] = (double)F.obs_buf[((long)r * DS + c) * B + b];
#pragma unroll
      for (int c = 0; c < DU; ++c) u[c] = (double)F.act_buf[((long)r * DU + c) * B + b];
      if (P.has_target)
        make_chi<DS, DU, true, double>(P, y, u, chi);
      else
        make_chi<DS, DU, false, double>(P, y, u, chi);
      critic_phi<CS, DS, DU>(chi, y, u, phi);
      if (r > 0) {  // gamma * w_prev . phi(row r) belongs to TD row r - 1
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < DC; ++i) q = fma_r(wp[i], phi[i], q);
        bv[r - 1 < MAXM ? r - 1 : 0] += P.gamma * q;
      }
      if (r < m && r < MAXM) {
#pragma unroll
        for (int i = 0; i < DC; ++i) A[r][i] = phi[i];
        bv[r] += stage_any<NCHI, double>(P, chi);
      }
    }
  }

  double tr = 0.0;
#pragma unroll
  for (int r = 0; r < MAXM; ++r)
#pragma unroll
    for (int i = 0; i < DC; ++i) tr = fma_r(A[r][i], A[r][i], tr);  // rows >= m are zero
  const double mu = FIT_MU_REL * (tr / (double)m) + 1e-300;
  const double inv_mu = 1.0 / mu;
  double bnorm = 0.0;
#pragma unroll
  for (int r = 0; r < MAXM; ++r) bnorm = fma_r(bv[r], bv[r], bnorm);
  bnorm = sqrt(bnorm);

  // w(y) = clip(w0 - A^T y / mu); returns -dual(y); optionally the primal cost, w and the free mask
  auto eval = [&](const double* yv, double* w_out, unsigned long long* free_out, double* primal_out) -> double {
    double acc = 0.0, res[MAXM];
    unsigned long long fm = 0ull;
#pragma unroll
    for (int r = 0; r < MAXM; ++r) res[r] = -bv[r];
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      double c = 0.0;
#pragma unroll
      for (int r = 0; r < MAXM; ++r) c = fma_r(A[r][i], yv[r], c);
      const double z = w0[i] - c * inv_mu;
      const double w = z < lo[i] ? lo[i] : (z > hi[i] ? hi[i] : z);
      if (z > lo[i] && z < hi[i]) fm |= (1ull << i);
      const double dw = w - w0[i];
      acc += 0.5 * mu * dw * dw + c * w;
      if (w_out) w_out[i] = w;
#pragma unroll
      for (int r = 0; r < MAXM; ++r) res[r] = fma_r(A[r][i], w, res[r]);
    }
    double yy = 0.0, by = 0.0, pr = 0.0;
#pragma unroll
    for (int r = 0; r < MAXM; ++r) {
      yy = fma_r(yv[r], yv[r], yy);
      by = fma_r(bv[r], yv[r], by);
      pr = fma_r(res[r], res[r], pr);
    }
    if (free_out) *free_out = fm;
    if (primal_out) *primal_out = 0.5 * pr;
    return 0.5 * yy + by - acc;
  };

  double yv[MAXM], w[DC], best_w[DC];
#pragma unroll
  for (int r = 0; r < MAXM; ++r) yv[r] = 0.0;
  unsigned long long fm;
  double Pw;
  double f = eval(yv, w, &fm, &Pw);  // y = 0  ->  w = w0 (w_init lies inside the box)
  double best_P = Pw;
#pragma unroll
  for (int i = 0; i < DC; ++i) best_w[i] = w[i];

  for (int it = 0; it < FIT_ITERS; ++it) {
    if (it > 0) {
      eval(yv, w, &fm, &Pw);
      if (Pw < best_P) {
        best_P = Pw;
#pragma unroll
        for (int i = 0; i < DC; ++i) best_w[i] = w[i];
      }
    }
    // g = -(A w - b - y)
    double g[MAXM], gn = 0.0;
#pragma unroll
    for (int r = 0; r < MAXM; ++r) {
      double s = -bv[r] - yv[r];
#pragma unroll
      for (int i = 0; i < DC; ++i) s = fma_r(A[r][i], w[i], s);
      g[r] = -s;
      gn = fma_r(s, s, gn);
    }
    if (sqrt(gn) <= FIT_GTOL * (bnorm + 1.0)) break;
    // H = I + (A_F A_F^T) / mu  (rows >= m: identity), Cholesky H = L L^T, d = -H^{-1} g
    double L[MAXM][MAXM];
#pragma unroll
    for (int r = 0; r < MAXM; ++r)
#pragma unroll
      for (int s = 0; s <= r; ++s) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < DC; ++i)
          if ((fm >> i) & 1ull) q = fma_r(A[r][i], A[s][i], q);
        L[r][s] = q * inv_mu + (r == s ? 1.0 : 0.0);
      }
#pragma unroll
    for (int j = 0; j < MAXM; ++j) {
      double dj = L[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k];
      dj = sqrt(dj);
      L[j][j] = dj;
#pragma unroll
      for (int i = j + 1; i < MAXM; ++i) {
        double s = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
        L[i][j] = s / dj;
      }
    }
    double d[MAXM];
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {  // forward: L z = -g
      double s = -g[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= L[i][k] * d[k];
      d[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = MAXM - 1; i >= 0; --i) {  // backward: L^T d = z
      double s = d[i];
#pragma unroll
      for (int k = i + 1; k < MAXM; ++k) s -= L[k][i] * d[k];
      d[i] = s / L[i][i];
    }
    double slope = 0.0;
#pragma unroll
    for (int r = 0; r < MAXM; ++r) slope = fma_r(g[r], d[r], slope);
    double t = 1.0, fn = f;
    double yn[MAXM];
    for (int ls = 0; ls < FIT_LS; ++ls) {
#pragma unroll
      for (int r = 0; r < MAXM; ++r) yn[r] = yv[r] + t * d[r];
      fn = eval(yn, nullptr, nullptr, nullptr);
      if (fn <= f + 1e-4 * t * slope) break;
      t *= 0.5;
    }
#pragma unroll
    for (int r = 0; r < MAXM; ++r) yv[r] = yv[r] + t * d[r];
    f = fn;
  }

#pragma unroll
  for (int i = 0; i < DC; ++i) {
    F.w_critic[(long)i * B + b] = (real)best_w[i];
    F.w_prev[(long)i * B + b] = (real)best_w[i];  // w_critic_prev = w_critic (controllers.py:1471)
  }
}

}  // namespace rcg
