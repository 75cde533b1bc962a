// rcg_critic_fit_gen.hpp - k_critic_fit_gen: the critic fit for ANY number of TD rows (Ncritic - 1 > 8).
//
// The reference only clips Ncritic to buffer_size - 1 (rcognita/controllers.py:1015) and its class default is buffer_size = 20
// (:828), so `--Ncritic 12 --buffer_size 20` is a legal run: a TD stack of 11 rows, up to 19 with the default buffer.  The
// exact-m kernels (rcg_critic_fit.hpp: m <= 3 and m <= 8, everything in registers) cannot hold such a stack; this one runs the
// SAME active-set walk - statement for statement what critic_update_env runs and oracle/rcg_oracle.py::critic_fit_single
// states, sums in the same order - with the m x dc stack, the m x m factor and the five m-vectors of an env in a per-handle
// scratch tensor in HBM (allocated on first use: (m dc + m^2 + 5 m) doubles per env, env index innermost, so every access of a
// wave is one contiguous 512-byte segment served by L2), and only the per-variable state (w, z, two accumulators) in
// registers.  lane == env.  Speed is secondary here (the walk rebuilds the m x m Gram matrix of the free columns every
// iteration: m (m + 1) / 2 x dc fused multiply-adds on operands that come from L2); every preset and every BASELINE config has
// Ncritic - 1 <= 3 and stays on the register kernels.
#pragma once
#include "rcg_critic_fit.hpp"

namespace rcg {

// doubles of scratch per env
__host__ __device__ constexpr long fit_gen_scratch_doubles(int m, int dc) { return (long)m * dc + (long)m * m + 5L * m; }

template <typename Sys, typename real, int CS>
__global__ __launch_bounds__(64) void k_critic_fit_gen(const FitArgs<real> F, const KParams<double> P, const KParams<real> Pr,
                                                       double* const S) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, DC = CriticDim<CS, DS, DU>::value;
  const long b = F.env_lo + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (F.env_hi > 0 ? (long)F.env_hi : P.B)) return;
  const long B = P.B;
  const int m = P.n_critic - 1;  // rows of the TD stack (>= 1, checked on the host)

  // [env step] -> [push]: the exact-m kernels' prologue with the fit switched off (same code, same bits)
  if (F.do_sim || F.do_push) {
    FitArgs<real> F0 = F;
    F0.do_fit = 0;
    double A1[1][DC], b1[1], t0[DC], t1[DC], t2[DC];
    critic_prologue<Sys, real, CS, 1>(F0, P, Pr, b, true, A1, b1, t0, t1, t2);
  }
  if (!F.do_fit) return;

  // scratch of this env: element k at S[k * B + b]
  const long oA = 0, oL = (long)m * DC, oB = oL + (long)m * m, oLam = oB + m, oD = oLam + m, oR = oD + m, oRes = oR + m;
#define SX(k) S[(long)(k) * B + b]
#define AX(r, i) SX(oA + (long)(r) * DC + (i))
#define LX(r, q) SX(oL + (long)(r) * m + (q))
  // wave-uniform: the start point and the box (scalar loads)
  auto w0 = [&](int i) -> double { return F.wcfg[i]; };
  auto lo = [&](int i) -> double { return F.wcfg[40 + i]; };
  auto hi = [&](int i) -> double { return F.wcfg[80 + i]; };

  // ---- TD stack from buffer rows 0 .. m (the oldest rows, controllers.py:1231-1234; the push above was this lane's own store) ----
  double tr = 0.0;
  {
    double wp[DC];
#pragma unroll
    for (int i = 0; i < DC; ++i) wp[i] = (double)F.w_prev[(long)i * B + b];
    for (int r = 0; r <= m; ++r) {
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
        SX(oB + r - 1) = SX(oB + r - 1) + P.gamma * q;
      }
      if (r < m) {
#pragma unroll
        for (int i = 0; i < DC; ++i) {
          AX(r, i) = phi[i];
          tr = fma_r(phi[i], phi[i], tr);
        }
        SX(oB + r) = 0.0 + stage_any<NCHI, double>(P, chi);
      }
    }
  }
  double mu = FIT_MU_REL * (tr / (double)m);
  if (!(mu > 1e-30)) mu = 1e-30;

  double w[DC], z[DC];
  unsigned long long fm = 0ull, at_hi = 0ull, blocked = 0ull;  // free / fixed-at-upper / not-to-release masks
#pragma unroll
  for (int i = 0; i < DC; ++i) {
    w[i] = w0(i) < lo(i) ? lo(i) : (w0(i) > hi(i) ? hi(i) : w0(i));
    z[i] = w[i];
    if (w[i] > lo(i) && w[i] < hi(i))
      fm |= 1ull << i;
    else if (w[i] >= hi(i))
      at_hi |= 1ull << i;
  }
  int last_freed = -1;

  for (int it = 0; it < fit_max_iters(DC); ++it) {
    // rhs = b - A_B w_B - A_F w0_F,  M = A_F A_F^T + mu I
    for (int r = 0; r < m; ++r) {
      double s = SX(oB + r);
      double ar[DC];
#pragma unroll
      for (int i = 0; i < DC; ++i) {
        const bool fr = (fm >> i) & 1ull;
        const double a = AX(r, i);
        s = fma_r(-a, fr ? w0(i) : w[i], s);
        ar[i] = fr ? a : 0.0;
      }
      SX(oLam + r) = s;
      for (int q = 0; q <= r; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < DC; ++i) {
          const bool fr = (fm >> i) & 1ull;
          const double aq = q == r ? ar[i] : (fr ? AX(q, i) : 0.0);
          acc = fma_r(ar[i], aq, acc);
        }
        LX(r, q) = acc + (r == q ? mu : 0.0);
      }
    }
    // root-free Cholesky M = L D L^T (unit lower L), the operation order of critic_update_env / _chol_solve
    const double floor_piv = mu * 1e-6;
    for (int j = 0; j < m; ++j) {
      double dj = LX(j, j);
      for (int k = 0; k < j; ++k) {
        const double l = LX(j, k);
        dj -= (l * l) * SX(oD + k);
      }
      if (!(dj > floor_piv)) dj = floor_piv;
      SX(oD + j) = dj;
      const double rcj = 1.0 / dj;
      SX(oR + j) = rcj;
      for (int i = j + 1; i < m; ++i) {
        double s = LX(i, j);
        for (int k = 0; k < j; ++k) s -= (LX(i, k) * LX(j, k)) * SX(oD + k);
        LX(i, j) = s * rcj;
      }
    }
    for (int i = 0; i < m; ++i) {  // forward: L y = rhs
      double s = SX(oLam + i);
      for (int k = 0; k < i; ++k) s -= LX(i, k) * SX(oLam + k);
      SX(oLam + i) = s;
    }
    for (int i = 0; i < m; ++i) SX(oLam + i) = SX(oLam + i) * SX(oR + i);  // D z = y
    for (int i = m - 1; i >= 0; --i) {  // backward: L^T lam = z
      double s = SX(oLam + i);
      for (int k = i + 1; k < m; ++k) s -= LX(k, i) * SX(oLam + k);
      SX(oLam + i) = s;
    }
    // z_F = w0_F + A_F^T lam and the ratio test towards it (cross-multiplied, one division per iteration)
    double cacc[DC];
#pragma unroll
    for (int i = 0; i < DC; ++i) cacc[i] = 0.0;
    for (int r = 0; r < m; ++r) {
      const double lr = SX(oLam + r);
#pragma unroll
      for (int i = 0; i < DC; ++i) cacc[i] = fma_r(AX(r, i), lr, cacc[i]);
    }
    double nb = 2.0, db = 1.0;
    int jmin = -1;
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      const bool fr = (fm >> i) & 1ull;
      const double zi = w0(i) + cacc[i];
      z[i] = fr ? zi : z[i];
      const bool vlo = zi < lo(i), vhi = zi > hi(i);
      const double ni = fabs((vlo ? lo(i) : hi(i)) - w[i]), di = fabs(zi - w[i]);
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
        v = v < lo(i) ? lo(i) : (v > hi(i) ? hi(i) : v);
        const bool up = z[i] > hi(i);
        if (i == jmin) {
          v = up ? hi(i) : lo(i);
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
#pragma unroll
    for (int i = 0; i < DC; ++i) w[i] = ((fm >> i) & 1ull) ? z[i] : w[i];
    // residual, then per variable the multiplier g_i = mu (w_i - w0_i) + sum_r A[r][i] res[r] and its scale
    double g[DC], sc[DC];
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      g[i] = mu * (w[i] - w0(i));
      sc[i] = fabs(g[i]);
    }
    for (int r = 0; r < m; ++r) {
      double rr = -SX(oB + r);
#pragma unroll
      for (int i = 0; i < DC; ++i) rr = fma_r(AX(r, i), w[i], rr);
      SX(oRes + r) = rr;
    }
    for (int r = 0; r < m; ++r) {
      const double rr = SX(oRes + r);
#pragma unroll
      for (int i = 0; i < DC; ++i) {
        const double t = AX(r, i) * rr;
        g[i] += t;
        sc[i] += fabs(t);
      }
    }
    int best = -1;
    double best_score = 0.0;
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      const double score = ((at_hi >> i) & 1ull) ? g[i] : -g[i];
      if (!(((fm | blocked) >> i) & 1ull) && score > FIT_KKT_TOL * sc[i] && score > best_score) {
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
  for (int r = 0; r < m; ++r) {
    double s = -SX(oB + r), s0 = s;
#pragma unroll
    for (int i = 0; i < DC; ++i) {
      const double wi = w0(i) < lo(i) ? lo(i) : (w0(i) > hi(i) ? hi(i) : w0(i));
      const double a = AX(r, i);
      s = fma_r(a, w[i], s);
      s0 = fma_r(a, wi, s0);
    }
    Pw = fma_r(s, s, Pw);
    P0 = fma_r(s0, s0, P0);
  }
  const bool keep = Pw <= P0;
#pragma unroll
  for (int i = 0; i < DC; ++i) {
    const double wi = w0(i) < lo(i) ? lo(i) : (w0(i) > hi(i) ? hi(i) : w0(i));
    const double v = keep ? w[i] : wi;
    F.w_critic[(long)i * B + b] = (real)v;
    F.w_prev[(long)i * B + b] = (real)v;  // w_critic_prev = w_critic (controllers.py:1471)
  }
#undef SX
#undef AX
#undef LX
}

}  // namespace rcg
