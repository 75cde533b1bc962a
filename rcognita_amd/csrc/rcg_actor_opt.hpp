// rcg_actor_opt.hpp - k_actor_opt: on-device replacement of the SLSQP call in CtrlOptPred._actor_optimizer
// (rcognita/controllers.py:1330-1427), MPC with a diagonal R1.
//
// One wave owns OPT_G = 16 envs.  Per iteration:
//   1. lane e < 16 = env e: gradient of _actor_cost w.r.t. the whole action sequence u [N][du] by a forward Euler
//      rollout (states to LDS) and a reverse (adjoint) sweep; direction d = g * (hi - lo)^2 (box-width metric) to LDS,
//      gn = max |d / (hi - lo)| stays in the lane;
//   2. four envs at a time, one per row of 16 lanes: OPT_NA = 16 step lengths alpha_l = 4^(1 - l) / gn, ONE PER LANE of
//      the row; the lane evaluates _actor_cost of clip(u_e - alpha_l d_e) - the same rollout as k_actor, u and d from
//      LDS (broadcast reads within the row); row argmin over (J, l) (lower J, then lower l; NaN = +inf; f32: four DPP
//      stages, a DPP row IS 16 lanes); if it improves the incumbent, the row's lanes update u_e in LDS, otherwise env e
//      stops.
// History (profiles/r02_*_valu_pmc.json has the SQ counters): v1 gave every env a whole wave and computed the gradient
// redundantly on all 64 lanes (55 % of its instruction stream); v2 shared a wave between 16 envs and searched 64 step
// lengths (ratio sqrt 2) per env with the whole wave - 56 k VALU instructions per wave for 5 iterations, 90 % of them the
// line search (16 passes of a full rollout per iteration), issue slots 93 % busy: VALU-bound on trial rollouts.  On the
// reference's own F8 states a 16-step ladder of ratio 4 over the same range (4 .. 2^-28 box widths) reaches the same
// cost to five digits (oracle/experiments/ladder_experiment.py), so v3 runs four envs per pass: 4 x fewer trial rollouts.
// No HBM traffic inside the loop.  Mirrors oracle/rcg_oracle.py::actor_optimize_single statement by statement; on the
// reference's own test states it reaches SLSQP's cost within 0.2 % after 10 iterations
// (tests/test_oracle_optimizer.py, tests/test_hip_optimizer.py).
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

constexpr int OPT_G = 16;   // envs per wave
constexpr int OPT_NA = 16;  // step lengths tried per env and iteration = lanes of one DPP row
constexpr int OPT_EP = 64 / OPT_NA;  // envs per line-search pass

// argmin over a row of 16 lanes of (cost, index): lower cost wins, ties -> lower index; every lane of the row ends with
// the row's winner
__device__ __forceinline__ void row16_argmin(float& bj, int& bi) {
  unsigned long long k = ((unsigned long long)float_order_key(bj) << 32) | (unsigned)bi;
#define RCG_DPP_MIN(CTRL)                                                                                          \
  {                                                                                                                \
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xF, 0xF, false);         \
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, false); \
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                              \
    k = o < k ? o : k;                                                                                             \
  }
  RCG_DPP_MIN(0xB1)   // quad_perm [1,0,3,2]
  RCG_DPP_MIN(0x4E)   // quad_perm [2,3,0,1]
  RCG_DPP_MIN(0x141)  // row_half_mirror
  RCG_DPP_MIN(0x140)  // row_mirror
#undef RCG_DPP_MIN
  bj = float_from_order_key((unsigned)(k >> 32));
  bi = (int)(unsigned)k;
}
__device__ __forceinline__ void row16_argmin(double& bj, int& bi) {
  for (int m = 1; m < 16; m <<= 1) {
    const double oJ = __shfl_xor(bj, m, 64);
    const int oI = __shfl_xor(bi, m, 64);
    if ((oJ < bj) || (oJ == bj && oI < bi)) {
      bj = oJ;
      bi = oI;
    }
  }
}

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

// reals of LDS one wave needs (host and device agree through this one function)
__host__ __device__ constexpr int opt_lds_reals(int N, int DS, int DU, int NP) {
  return OPT_G * (2 * N * DU + N * DS + 2 * DS + (NP > 0 ? NP : 1)) + N;
}

template <typename Sys, typename real, bool TGT>
__global__ __launch_bounds__(256) void k_actor_opt(const OptArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, NP = Sys::NP, NPS = NP > 0 ? NP : 1, G = OPT_G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const long b0 = wave * G;
  if (b0 >= B) return;
  const int ng = (int)((B - b0) < G ? (B - b0) : G);  // envs of this wave (wave-uniform)
  const int N = P.n_actor, R = N * DU;
  // per-wave LDS: u [G][R] | d [G][R] | X [N*DS][G] | y0 [DS][G] | xs [DS][G] | pars [NPS][G]
  real* const su = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * opt_lds_reals(N, DS, DU, NP);
  real* const sd = su + G * R;
  real* const sX = sd + G * R;
  real* const sY = sX + N * DS * G;
  real* const sS = sY + DS * G;
  real* const sP = sS + DS * G;
  real* const sg = sP + NPS * G;  // gamma^k, k < N, formed as the forward sum forms it (gk = 1; gk *= gamma)

  const bool mine = lane < ng;       // lane == env view
  const long be = b0 + (mine ? lane : 0);
  real y0e[DS], xse[DS], pve[NPS], w[DU], w2[DU];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    y0e[c] = A.obs[(long)c * B + be];
    xse[c] = A.state_sys[(long)c * B + be];
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) pve[i] = A.pars_env ? A.pars_env[(long)i * B + be] : P.pars[i];
  const auto pre_e = Sys::template prepare<real>(pve);
#pragma unroll
  for (int c = 0; c < DU; ++c) {
    w[c] = P.hi[c] - P.lo[c];
    w2[c] = w[c] * w[c];
  }
  const real h = P.h_pred;
  if (lane < G) {  // the env data once more in LDS: phase 2 needs env e's values wave-uniformly
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      sY[c * G + lane] = y0e[c];
      sS[c * G + lane] = xse[c];
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) sP[i * G + lane] = pve[i];
  }

  // initial sequences -> LDS, direction zero until the first gradient
  for (int idx = lane; idx < ng * R; idx += 64) {
    const int e = idx / R, i = idx - e * R;
    real v;
    if (A.u_init) {
      int j = i;
      if (A.shift) j = (i + DU < R) ? i + DU : i;  // u_k <- u_{k+1}, the last step repeated
      v = A.u_init[(b0 + e) * R + j];
    } else {
      v = A.u0[i % DU];
    }
    su[idx] = v;
    sd[idx] = 0;
  }
  if (lane == 0) {
    real gk = 1;
    for (int k = 0; k < N; ++k) {
      sg[k] = gk;
      gk *= P.gamma;
    }
  }
  wave_lds_sync();

  const bool g1 = P.gamma == (real)1;
  // _actor_cost of clip(u_e - alpha d_e) from the state (y0, xs, pre): controllers.py:1284-1306
  auto cost_of = [&](const real* ue, const real* de, const real* y0, const real* xs,
                     const typename Sys::template Pre<real>& pre, real alpha) -> real {
    real x[DS], y[DS], up[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      x[c] = xs[c];
      y[c] = y0[c];
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) up[c] = 0;
    real J = 0;
    real S[NCHI];  // gamma == 1: per-component sums of squares, weighted once at the end (as the rollout kernels)
#pragma unroll
    for (int i = 0; i < NCHI; ++i) S[i] = 0;
    for (int k = 0; k < N; ++k) {
      real u[DU];
#pragma unroll
      for (int c = 0; c < DU; ++c) u[c] = clamp_r<real>(fma_r(-alpha, de[k * DU + c], ue[k * DU + c]), P.lo[c], P.hi[c]);
      if (k > 0) {
        real d[DS];
        // f32: hardware v_sin/v_cos behind the exact reduction, as in every f32 rollout of the build (the 64 trial
        // rollouts of the line search are where this kernel spends its instructions)
        Sys::template rhs<real, true>(pre, x, up, d);
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          x[c] = fma_r(h, d[c], x[c]);
          y[c] = x[c];
        }
      }
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, y, u, chi);
      if (g1) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < NCHI; ++i) S[i] = fma_r(chi[i], chi[i], S[i]);
      } else {
        J = fma_r(sg[k], stage_diag<NCHI, real>(P, chi), J);
      }
#pragma unroll
      for (int c = 0; c < DU; ++c) up[c] = u[c];
    }
    if (g1) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) J = fma_r(P.R1d[i], S[i], J);
    }
    return J;
  };

  // lane == env registers
  real Jinc = mine ? cost_of(su + lane * R, sd + lane * R, y0e, xse, pre_e, (real)0) : (real)0;
  int used = 0;
  bool active = mine;
  real gn = 0;
  const int row = lane >> 4, tl = lane & (OPT_NA - 1);                    // row of 16 lanes = one env of the pass, trial
  const real ladder = (real)exp2((double)2 - 2.0 * (double)tl);          // alpha_l * gn = 4^(1 - l)

  for (int it = 0; it < A.iters; ++it) {
    if (__builtin_amdgcn_readfirstlane((int)__builtin_popcountll(__ballot(active))) == 0) break;
    // ---- 1. lane == env: forward rollout (states to LDS), reverse adjoint sweep (direction to LDS) --------
    if (active) {
      const real* ue = su + lane * R;
      real x[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) x[c] = xse[c];
      for (int k = 1; k < N; ++k) {
        real u[DU], d[DS];
#pragma unroll
        for (int c = 0; c < DU; ++c) u[c] = ue[(k - 1) * DU + c];
        // f32: the hardware-trig rollout every cost evaluation of this kernel uses, so that the gradient is the gradient
        // of the function the line search evaluates (and a third of phase 1's instructions go away)
        Sys::template rhs<real, true>(pre_e, x, u, d);
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          x[c] = fma_r(h, d[c], x[c]);
          sX[(k * DS + c) * G + lane] = x[c];
        }
      }
      gn = 0;
      real lam[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) lam[c] = 0;
      for (int k = N - 1; k >= 0; --k) {
        const real gk = sg[k];
        real u[DU], xk[DS], g[DU];
#pragma unroll
        for (int c = 0; c < DU; ++c) {
          u[c] = ue[k * DU + c];
          g[c] = gk * (real)2 * P.R1d[DS + c] * u[c];
        }
#pragma unroll
        for (int c = 0; c < DS; ++c) xk[c] = (k >= 1) ? sX[(k * DS + c) * G + lane] : xse[c];
        real lamk[DS];
        if (k < N - 1) {
          real ax[DS], bu[DU];
          Sys::template jac_T<real, true>(pre_e, xk, u, lam, ax, bu);
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
          sd[lane * R + k * DU + c] = dc;
          const real m = (dc < 0 ? -dc : dc) / w[c];
          gn = m > gn ? m : gn;
        }
      }
      if (!(gn > (real)0) || !finite_r<real>(gn)) active = false;
    }
    wave_lds_sync();

    // ---- 2. four envs per pass, one per row of 16 lanes: 16-way line search, accept or stop ----------------
    for (int e0 = 0; e0 < ng; e0 += OPT_EP) {
      const unsigned long long am = __ballot(active);
      if (!((am >> e0) & ((1ull << OPT_EP) - 1))) continue;  // wave-uniform: none of the four is still running
      const int e = e0 + row;                                 // this row's env (row-uniform)
      const bool on = e < ng && ((am >> e) & 1ull);
      const int es = on ? e : e0;                             // idle rows read a valid slot, their result is dropped
      const real gn_e = __shfl(gn, es, 64);
      const real Jinc_e = __shfl(Jinc, es, 64);
      real y0[DS], xs[DS], pv[NPS];
#pragma unroll
      for (int c = 0; c < DS; ++c) {
        y0[c] = sY[c * G + es];
        xs[c] = sS[c * G + es];
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) pv[i] = sP[i * G + es];
      const auto pre = Sys::template prepare<real>(pv);
      real* const ue = su + es * R;
      const real* const de = sd + es * R;
      const real alpha = ((real)1 / gn_e) * ladder;
      real bj = inf_r<real>();
      if (on) {
        const real J = cost_of(ue, de, y0, xs, pre, alpha);
        bj = (J != J) ? inf_r<real>() : J;
      }
      int bi = tl;
      row16_argmin(bj, bi);
      const bool better = on && (bj < Jinc_e);  // row-uniform
      // accept: u_e <- clip(u_e - alpha_best d_e); otherwise env e is done
      const real abest = ((real)1 / gn_e) * (real)exp2((double)2 - 2.0 * (double)bi);
      wave_lds_sync();  // every lane has finished reading its u_e
      if (better) {
        for (int i = tl; i < R; i += OPT_NA) {
          const int c = i % DU;
          ue[i] = clamp_r<real>(fma_r(-abest, de[i], ue[i]), P.lo[c], P.hi[c]);
        }
      }
      // hand the rows' results to the env-view lanes (env e0 + r lives in lane e0 + r; its row is lanes 16 r ...)
      const int r_of_me = lane - e0;  // which row carries the env this lane stands for
      const bool mine_now = r_of_me >= 0 && r_of_me < OPT_EP;
      const int src = mine_now ? r_of_me * OPT_NA : 0;
      const real bj_me = __shfl(bj, src, 64);
      const int better_me = __shfl((int)better, src, 64);
      if (mine_now && active) {
        if (better_me) {
          Jinc = bj_me;
          ++used;
        } else {
          active = false;
        }
      }
    }
    wave_lds_sync();
  }

  for (int idx = lane; idx < ng * R; idx += 64)
    if (A.u_opt) A.u_opt[b0 * R + idx] = su[idx];
  if (mine) {
    real a[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) {
      a[c] = su[lane * R + c];
      if (A.action_out) A.action_out[(long)c * B + be] = a[c];
    }
    if (A.best_J) A.best_J[be] = Jinc;
    if (A.n_iter) A.n_iter[be] = used;
    if (A.accum) {
      real chi[NCHI];
      make_chi<DS, DU, TGT, real>(P, y0e, a, chi);
      A.accum[be] += stage_diag<NCHI, real>(P, chi) * P.sampling_time;
    }
    if (A.step_idx) A.step_idx[be] += 1;
  }
}

}  // namespace rcg
