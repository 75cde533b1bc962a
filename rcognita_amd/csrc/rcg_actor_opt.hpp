// rcg_actor_opt.hpp - k_actor_opt: on-device replacement of the SLSQP call in CtrlOptPred._actor_optimizer
// (rcognita/controllers.py:1330-1427), MPC with a diagonal R1.
//
// One wave owns one env.  Per iteration:
//   1. gradient of _actor_cost w.r.t. the whole action sequence u [N][du] by a forward Euler rollout and a reverse
//      (adjoint) sweep - computed once per wave (wave-uniform data, every lane executes the same instruction
//      stream, lane 0 publishes to LDS);
//   2. direction d = g * (hi - lo)^2 (box-width metric); 64 step lengths alpha_l = 2^(2 - l/2) / max|d/(hi-lo)|,
//      ONE PER LANE; lane l evaluates _actor_cost of clip(u - alpha_l d) - this is the same rollout as k_actor,
//      reading u and d from LDS (broadcast reads);
//   3. wave argmin over (J, l) (lower J, then lower l; NaN = +inf); if it improves the incumbent, lanes i < N*du
//      update u[i] in LDS, otherwise the search stops.
// No HBM traffic inside the loop; ~1e3 wave instructions per iteration.  Mirrors oracle/rcg_oracle.py::
// actor_optimize_single statement by statement; on the reference's own test states it reaches SLSQP's cost within
// 0.2 % after 10 iterations (tests/test_oracle_optimizer.py, tests/test_hip_optimizer.py).
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

template <typename real>
struct OptArgs {
  const real* obs;        // [dy][B]
  const real* state_sys;  // [ds][B]
  const real* pars_env;   // [np][B] or nullptr
  const real* u_init;     // [B][N][du] or nullptr (-> u0 tiled over the horizon)
  real* u_opt;            // [B][N][du] or nullptr
  real* action_out;       // [du][B] or nullptr
  real* best_J;           // [B] or nullptr
  int32_t* n_iter;        // [B] or nullptr
  real* accum;            // tick epilogue (or nullptr)
  int32_t* step_idx;      // tick epilogue (or nullptr)
  real u0[RCG_MAX_DU];    // action_sqn_init entry (controllers.py:973-978)
  int iters;
  int shift;              // warm start: u_init is last tick's optimum, shift it by one step (last entry repeated)
};

__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <typename Sys, typename real, bool TGT>
__global__ __launch_bounds__(256) void k_actor_opt(const OptArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, NP = Sys::NP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long b = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  if (b >= B) return;
  const int N = P.n_actor, R = N * DU;
  // per-wave LDS: u [R] | d [R] | X [N][DS] | gk [N]
  real* const su = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * (2 * R + N * DS + N);
  real* const sd = su + R;
  real* const sX = sd + R;
  real* const sg = sX + N * DS;

  real y0[DS], xs[DS], pv[NP > 0 ? NP : 1], w[DU], w2[DU];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    y0[c] = A.obs[(long)c * B + b];
    xs[c] = A.state_sys[(long)c * B + b];
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) pv[i] = A.pars_env ? A.pars_env[(long)i * B + b] : P.pars[i];
  const auto pre = Sys::template prepare<real>(pv);
#pragma unroll
  for (int c = 0; c < DU; ++c) {
    w[c] = P.hi[c] - P.lo[c];
    w2[c] = w[c] * w[c];
  }
  const real h = P.h_pred;

  // initial sequence -> LDS (lane i owns element i)
  if (lane < R) {
    real v;
    if (A.u_init) {
      int i = lane;
      if (A.shift) i = (lane + DU < R) ? lane + DU : lane;  // u_k <- u_{k+1}, the last step repeated
      v = A.u_init[b * R + i];
    } else {
      v = A.u0[lane % DU];
    }
    su[lane] = v;
  }
  if (lane == 0) {
    real gk = 1;
    for (int k = 0; k < N; ++k) {
      sg[k] = gk;
      gk *= P.gamma;
    }
  }
  wave_lds_sync();

  // _actor_cost of the sequence clip(u - alpha d) (alpha = 0: of u itself); controllers.py:1284-1306
  auto cost_of = [&](real alpha) -> real {
    real x[DS], y[DS], up[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      x[c] = xs[c];
      y[c] = y0[c];
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) up[c] = 0;
    real J = 0;
    for (int k = 0; k < N; ++k) {
      real u[DU];
#pragma unroll
      for (int c = 0; c < DU; ++c) u[c] = clamp_r<real>(fma_r(-alpha, sd[k * DU + c], su[k * DU + c]), P.lo[c], P.hi[c]);
      if (k > 0) {
        real d[DS];
        Sys::template rhs<real>(pre, x, up, d);
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          x[c] = fma_r(h, d[c], x[c]);
          y[c] = x[c];
        }
      }
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, y, u, chi);
      J = fma_r(sg[k], stage_diag<NCHI, real>(P, chi), J);
#pragma unroll
      for (int c = 0; c < DU; ++c) up[c] = u[c];
    }
    return J;
  };

  // direction: zero until the first gradient
  if (lane < R) sd[lane] = 0;
  wave_lds_sync();
  real Jinc = cost_of((real)0);
  int used = 0;

  for (int it = 0; it < A.iters; ++it) {
    // ---- 1. forward rollout (states to LDS), reverse adjoint sweep (direction to LDS) -------------------
    {
      real x[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) x[c] = xs[c];
      for (int k = 1; k < N; ++k) {
        real u[DU], d[DS];
#pragma unroll
        for (int c = 0; c < DU; ++c) u[c] = su[(k - 1) * DU + c];
        Sys::template rhs<real>(pre, x, u, d);
#pragma unroll
        for (int c = 0; c < DS; ++c) x[c] = fma_r(h, d[c], x[c]);
        if (lane == 0) {
#pragma unroll
          for (int c = 0; c < DS; ++c) sX[k * DS + c] = x[c];
        }
      }
    }
    wave_lds_sync();
    real gn = 0;
    {
      real lam[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) lam[c] = 0;
      for (int k = N - 1; k >= 0; --k) {
        const real gk = sg[k];
        real u[DU], xk[DS], g[DU];
#pragma unroll
        for (int c = 0; c < DU; ++c) {
          u[c] = su[k * DU + c];
          g[c] = gk * (real)2 * P.R1d[DS + c] * u[c];
        }
#pragma unroll
        for (int c = 0; c < DS; ++c) xk[c] = (k >= 1) ? sX[k * DS + c] : xs[c];
        real lamk[DS];
        if (k < N - 1) {
          real ax[DS], bu[DU];
          Sys::template jac_T<real>(pre, xk, u, lam, ax, bu);
#pragma unroll
          for (int c = 0; c < DU; ++c) g[c] = fma_r(h, bu[c], g[c]);
#pragma unroll
          for (int c = 0; c < DS; ++c) lamk[c] = fma_r(h, ax[c], lam[c]);
        } else {
#pragma unroll
          for (int c = 0; c < DS; ++c) lamk[c] = 0;
        }
        if (k >= 1) {  // y_0 is the observation, not a function of the actions
#pragma unroll
          for (int c = 0; c < DS; ++c)
            lamk[c] += gk * (real)2 * P.R1d[c] * (TGT ? xk[c] - P.target[c] : xk[c]);
        }
#pragma unroll
        for (int c = 0; c < DS; ++c) lam[c] = lamk[c];
#pragma unroll
        for (int c = 0; c < DU; ++c) {
          const real dc = g[c] * w2[c];
          if (lane == 0) sd[k * DU + c] = dc;
          const real m = (dc < 0 ? -dc : dc) / w[c];
          gn = m > gn ? m : gn;
        }
      }
    }
    wave_lds_sync();
    if (!(gn > (real)0) || !finite_r<real>(gn)) break;  // wave-uniform

    // ---- 2. 64-way line search ------------------------------------------------------------------------
    const real alpha = ((real)1 / gn) * (real)exp2((double)2 - 0.5 * (double)lane);
    const real J = cost_of(alpha);
    real bj = (J != J) ? inf_r<real>() : J;
    int bi = lane;
    for (int m = 1; m < 64; m <<= 1) {
      const real oJ = __shfl_xor(bj, m, 64);
      const int oI = __shfl_xor(bi, m, 64);
      if ((oJ < bj) || (oJ == bj && oI < bi)) {
        bj = oJ;
        bi = oI;
      }
    }
    if (!(bj < Jinc)) break;  // wave-uniform

    // ---- 3. accept: u <- clip(u - alpha_best d) -------------------------------------------------------
    const real abest = ((real)1 / gn) * (real)exp2((double)2 - 0.5 * (double)bi);
    if (lane < R) {
      const int c = lane % DU;
      su[lane] = clamp_r<real>(fma_r(-abest, sd[lane], su[lane]), P.lo[c], P.hi[c]);
    }
    wave_lds_sync();
    Jinc = bj;
    ++used;
  }

  if (A.u_opt && lane < R) A.u_opt[b * R + lane] = su[lane];
  if (lane == 0) {
    real a[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) {
      a[c] = su[c];
      if (A.action_out) A.action_out[(long)c * B + b] = a[c];
    }
    if (A.best_J) A.best_J[b] = Jinc;
    if (A.n_iter) A.n_iter[b] = used;
    if (A.accum) {
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, y0, a, chi);
      A.accum[b] += stage_diag<NCHI, real>(P, chi) * P.sampling_time;
    }
    if (A.step_idx) A.step_idx[b] += 1;
  }
}

}  // namespace rcg
