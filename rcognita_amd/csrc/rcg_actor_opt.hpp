// rcg_actor_opt.hpp - k_actor_opt: on-device replacement of the SLSQP call in CtrlOptPred._actor_optimizer
// (rcognita/controllers.py:1330-1427) for every mode (MPC / RQL / SQL, controllers.py:1304-1326),
// stage-cost structure (diagonal or full R1, biquadratic: controllers.py:1076-1082) and critic structure
// (controllers.py:1204-1212).
//
// Algorithm (oracle twin: oracle/rcg_oracle.py::actor_optimize_single, same statements in the same order): projected
// limited-memory quasi-Newton descent on the box [lo, hi]^N.  One wave owns OPT_G = 16 envs.  Per iteration:
//   1. lane e < 16 = env e: gradient of _actor_cost w.r.t. the whole action sequence u [N][du] by a forward Euler
//      rollout (states to LDS) and a reverse (adjoint) sweep - the stage terms contribute gamma^k d rho / d chi, the
//      critic terms of RQL (last step) and SQL (every step) d (w . phi) / d chi, closed form for all four structures;
//      the (s, y) pair of the last accepted step is completed; the free set (a coordinate on a bound whose descent
//      direction leaves the box is held) goes to a 64-bit mask; direction d = H g on the free set by the L-BFGS
//      two-loop recursion over the <= `memory` pairs kept in LDS, initial metric scale * diag((hi - lo)^2); without
//      pairs, or when d is not a descent direction, d = (hi - lo)^2 g (box-scaled steepest descent);
//   2. four envs at a time, one per row of 16 lanes: OPT_NA = 16 step lengths, ONE PER LANE of the row - quasi-Newton
//      alpha_l = 2^(2 - l), steepest descent alpha_l = 4^(1 - l) / max |d / (hi - lo)|; the lane evaluates _actor_cost of
//      clip(u_e - alpha_l d_e), u and d from LDS (broadcast reads within the row); row argmin over (J, l) (lower J, then
//      lower l; NaN = +inf; f32: four DPP stages, a DPP row IS 16 lanes); if it improves the incumbent the row's lanes
//      update u_e and open the next pair, otherwise a quasi-Newton env drops its memory (and retries with steepest
//      descent), a steepest-descent env stops.
// No HBM traffic inside the loop.
// History (profiles/r02_*_valu_pmc.json has the SQ counters): v1 gave every env a whole wave and computed the gradient
// redundantly on all 64 lanes; v2 shared a wave between 16 envs and searched 64 step lengths (ratio sqrt 2) with the
// whole wave - 90 % of its instructions were the line search; v3 (round 2-3): 16-step ladder of ratio 4, four envs per
// pass, steepest descent only, MPC with a diagonal R1 only.  v4 (round 4): every mode and structure, and the curvature
// pairs - steepest descent stalls 3-14 % above SLSQP's cost on the reference's critic-mode ticks (fixtures F8c: the
// terminal action of RQL carries 1e4 times the curvature of the others), four pairs reach it within 0.5 % in 20
// iterations (tests/test_oracle_optimizer.py, tests/test_hip_optimizer.py).
#pragma once
#include "rcg_kernels.hpp"
#include "rcg_loop.hpp"

namespace rcg {

constexpr int OPT_G = 16;   // envs per wave
constexpr int OPT_NA = 16;  // step lengths tried per env and iteration = lanes of one DPP row
constexpr int OPT_EP = 64 / OPT_NA;  // envs per line-search pass
constexpr int OPT_MAXM = 8;  // most curvature pairs an env may keep (rcg_set_optimizer)

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

// quad helpers of phase 1b: (p0 + p1) + (p2 + p3) and the maximum over the four lanes of a quad, the same bits in all four
__device__ __forceinline__ float opt_quad_xchg(float x, bool swap1) {
  const int b = __float_as_int(x);
  return __int_as_float(swap1 ? __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, false)
                              : __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, false));
}
__device__ __forceinline__ double opt_quad_xchg(double x, bool swap1) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  unsigned lo, hi;
  if (swap1) {
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0xB1, 0xF, 0xF, false);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0xB1, 0xF, 0xF, false);
  } else {
    lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x4E, 0xF, 0xF, false);
    hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x4E, 0xF, 0xF, false);
  }
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename real>
__device__ __forceinline__ real opt_quad_sum(real x) {
  x += opt_quad_xchg(x, true);   // lanes 0 <-> 1, 2 <-> 3
  return x + opt_quad_xchg(x, false);  // 0 <-> 2, 1 <-> 3
}
template <typename real>
__device__ __forceinline__ real opt_quad_max(real x) {
  real o = opt_quad_xchg(x, true);
  x = o > x ? o : x;
  o = opt_quad_xchg(x, false);
  return o > x ? o : x;
}

template <typename real>
struct OptArgs {
  const real* obs;        // [dy][B]
  const real* state_sys;  // [ds][B]
  const real* pars_env;   // [np][B] or nullptr
  const real* w;          // [dc][B] critic weights (RQL / SQL) or nullptr
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
  int memory;             // curvature pairs kept per env, 0 .. OPT_MAXM (0: projected steepest descent)
  int dcw;                // critic weights staged in LDS per env (dc for RQL / SQL, else 0)
  real ftol;              // an env is done after an accepted step that lowered J by <= ftol (rcg_set_optimizer_tol; 0: never)
  LoopArgs<real> loop;    // LOOP instances only (rcg_loop_step): the loop iteration's head and tail around the decision
};

__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// reals of LDS one wave needs (host and device agree through this one function):
// u [G][R] | d [G][R] (= q, the two-loop work vector) | X [N*DS][G] | y0 [DS][G] | xs [DS][G] | pars [NPS][G] | w [DC][G] | g [R][G] |
// S [M][R][G] | Y [M][R][G] | gamma^k [N]
__host__ __device__ constexpr int opt_lds_reals(int N, int DS, int DU, int NP, int DC, int M) {
  return OPT_G * (2 * N * DU + N * DS + 2 * DS + (NP > 0 ? NP : 1) + DC + N * DU + 2 * M * N * DU) + N;
}

// gk * d rho / d chi of stage_obj (controllers.py:1076-1082): quadratic chi R1 chi -> (R1 + R1^T) chi;
// biquadratic adds chi^2 R2 chi^2 -> 2 chi * ((R2 + R2^T) chi^2)
template <int NCHI, typename real>
__device__ __forceinline__ void stage_grad_with(const KParams<real>& P, const real* chi, const int sk, real gk, real* g) {
  if (!(sk & STAGE_FULL)) {
#pragma unroll
    for (int i = 0; i < NCHI; ++i) g[i] = ((real)2 * P.R1d[i]) * chi[i];
    if (sk & STAGE_BIQUAD) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) g[i] = fma_r((real)4 * P.R2d[i] * (chi[i] * chi[i]), chi[i], g[i]);
    }
  } else {
#pragma unroll
    for (int p = 0; p < NCHI; ++p) {
      real v = 0;
#pragma unroll
      for (int j = 0; j < NCHI; ++j) v = fma_r(P.Rfull[p * NCHI + j] + P.Rfull[j * NCHI + p], chi[j], v);
      g[p] = v;
    }
    if (sk & STAGE_BIQUAD) {
      real c2[NCHI];
#pragma unroll
      for (int i = 0; i < NCHI; ++i) c2[i] = chi[i] * chi[i];
#pragma unroll
      for (int p = 0; p < NCHI; ++p) {
        real v = 0;
#pragma unroll
        for (int j = 0; j < NCHI; ++j) v = fma_r(P.Rfull[49 + p * NCHI + j] + P.Rfull[49 + j * NCHI + p], c2[j], v);
        g[p] = fma_r((real)2 * chi[p], v, g[p]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NCHI; ++i) g[i] *= gk;
}

// (d Q / d y, d Q / d u) of _critic = w . phi (controllers.py:1192-1214); chi = [y - target, u] so d chi / d y = I;
// quad-mix works on the raw observation (controllers.py:1212).  Feature order = critic_with's.
template <int DS, int DU, typename real, typename WGet>
__device__ __forceinline__ void critic_grad_with(const real* chi, const real* y, const real* u, WGet w, const int cs,
                                                 real* gy, real* gu) {
  constexpr int NCHI = DS + DU;
  if (cs == RCG_CRITIC_QUAD_MIX) {
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      real v = ((real)2 * w(i)) * y[i];
#pragma unroll
      for (int c = 0; c < DU; ++c) v = fma_r(w(DS + i * DU + c), u[c], v);
      gy[i] = v;
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) {
      real v = ((real)2 * w(DS + DS * DU + c)) * u[c];
#pragma unroll
      for (int i = 0; i < DS; ++i) v = fma_r(w(DS + i * DU + c), y[i], v);
      gu[c] = v;
    }
    return;
  }
  real g[NCHI];
  if (cs == RCG_CRITIC_QUAD_NOMIX) {
#pragma unroll
    for (int i = 0; i < NCHI; ++i) g[i] = ((real)2 * w(i)) * chi[i];
  } else {
#pragma unroll
    for (int i = 0; i < NCHI; ++i) g[i] = 0;
    int idx = 0;  // uptria2vec order; the diagonal entry is added twice: 2 w_pp chi_p
#pragma unroll
    for (int i = 0; i < NCHI; ++i)
#pragma unroll
      for (int j = i; j < NCHI; ++j) {
        const real wij = w(idx++);
        g[i] = fma_r(wij, chi[j], g[i]);
        g[j] = fma_r(wij, chi[i], g[j]);
      }
    if (cs == RCG_CRITIC_QUAD_LIN) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) g[i] += w(idx++);
    }
  }
#pragma unroll
  for (int i = 0; i < DS; ++i) gy[i] = g[i];
#pragma unroll
  for (int c = 0; c < DU; ++c) gu[c] = g[DS + c];
}

// GENERIC = false: MPC with a diagonal quadratic stage cost (every preset in its default mode); true: the rest, mode and
// structures read from KParams (wave-uniform branches)
// PAIRS = false: the instance without curvature pairs (memory 0: box-scaled steepest descent, the default of MPC with a diagonal
// stage cost) - the quad phase 1b and its registers are compiled out (121 VGPRs, 4 waves per SIMD; 153 with it)
// LOOP (rcg_loop_step's sample with the plain MPC decision, one launch instead of three): lane == env first does k_loop's head -
// System.receive_action + Simulator.sim_step, the fields written as k_loop writes them - and decides from (obs = the new STATE,
// state_sys = STATE_PREV) held in its registers; after the decision it does k_loop's tail (stage_obj of the new state and the
// decided action, the row and the sequence number into the pinned host buffer).  Same device functions as the three launches.
template <typename Sys, typename real, bool TGT, bool GENERIC, bool PAIRS, bool LOOP = false>
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
  const int M = PAIRS ? A.memory : 0, DCW = A.dcw;
  real* const su = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * opt_lds_reals(N, DS, DU, NP, DCW, M);
  real* const sd = su + G * R;
  real* const sX = sd + G * R;
  real* const sY = sX + N * DS * G;
  real* const sS = sY + DS * G;
  real* const sP = sS + DS * G;
  real* const sW = sP + NPS * G;    // critic weights [DCW][G]
  real* const sGc = sW + DCW * G;   // current gradient [R][G]
  real* const sQ = sd;              // two-loop work vector [G][R]: IN the direction's storage (d of the last iteration is dead by then
                                    // and the recursion's result is the new d: round 6, 1.25 KB per wave that decide 7 or 8 waves per CU)
  real* const sLS = sGc + R * G;    // pairs: s [M][R][G]
  real* const sLY = sLS + M * R * G;  //        y [M][R][G]
  real* const sg = sLY + M * R * G;  // gamma^k, k < N, formed as the forward sum forms it (gk = 1; gk *= gamma)

  const bool mine = lane < ng;       // lane == env view
  const long be = b0 + (mine ? lane : 0);
  real y0e[DS], xse[DS], pve[NPS], w[DU], w2[DU];
  if constexpr (LOOP) {
    real ul[DU];
    if (mine) {
      loop_head<Sys, real>(A.loop, P, be, ul, y0e, xse);
    } else {
#pragma unroll
      for (int c = 0; c < DS; ++c) y0e[c] = xse[c] = (real)0;  // an idle lane stands for no env (its values are never used)
    }
  } else {
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      y0e[c] = A.obs[(long)c * B + be];
      xse[c] = A.state_sys[(long)c * B + be];
    }
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
    if (GENERIC)
      for (int i = 0; i < DCW; ++i) sW[i * G + lane] = A.w[(long)i * B + be];
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
  const int mode = GENERIC ? P.mode : (int)RCG_MODE_MPC, sk = GENERIC ? P.stage_kind : 0, cs = P.critic_struct;
  // _actor_cost of clip(u_e - alpha d_e) from the state (y0, xs, pre): controllers.py:1284-1326; `wg(i)`: weight i of
  // the env the caller stands for
  auto cost_of = [&](const real* ue, const real* de, const real* y0, const real* xs,
                     const typename Sys::template Pre<real>& pre, real alpha, auto wg) -> real {
    real x[DS], y[DS], up[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      x[c] = xs[c];
      y[c] = y0[c];
    }
#pragma unroll
    for (int c = 0; c < DU; ++c) up[c] = 0;
    real J = 0;
    real S[NCHI];  // MPC, diagonal, gamma == 1: per-component sums of squares, weighted once at the end
#pragma unroll
    for (int i = 0; i < NCHI; ++i) S[i] = 0;
    for (int k = 0; k < N; ++k) {
      real u[DU];
#pragma unroll
      for (int c = 0; c < DU; ++c)  // (finite: a direction that is not finite has ended the env's search in phase 1)
        u[c] = clamp_fin(fma_r(-alpha, de[k * DU + c], ue[k * DU + c]), P.lo[c], P.hi[c]);
      if (k > 0) {
        real d[DS];
        // f32: hardware v_sin/v_cos behind the exact reduction, as in every f32 rollout of the build (the trial
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
      if (!GENERIC) {
        if (g1) {  // wave-uniform
#pragma unroll
          for (int i = 0; i < NCHI; ++i) S[i] = fma_r(chi[i], chi[i], S[i]);
        } else {
          J = fma_r(sg[k], stage_diag<NCHI, real>(P, chi), J);
        }
      } else if (mode == RCG_MODE_SQL || (mode == RCG_MODE_RQL && k == N - 1)) {  // wave-uniform
        J += critic_with<DS, DU, real>(chi, y, u, wg, cs);
      } else {
        J = fma_r(sg[k], stage_with<NCHI, real>(P, chi, sk), J);
      }
#pragma unroll
      for (int c = 0; c < DU; ++c) up[c] = u[c];
    }
    if (!GENERIC && g1) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) J = fma_r(P.R1d[i], S[i], J);
    }
    return J;
  };
  auto w_mine = [&](int i) -> real { return sW[i * G + lane]; };

  // lane == env registers
  real Jinc = mine ? cost_of(su + lane * R, sd + lane * R, y0e, xse, pre_e, (real)0, w_mine) : (real)0;
  int used = 0;
  bool active = mine;
  real gn = 0;
  bool quasi = false, pending = false;  // this iteration's direction is quasi-Newton; a pair waits for its y
  int head = 0, n_pairs = 0;            // ring of curvature pairs: next slot, pairs held
  const int row = lane >> 4, tl = lane & (OPT_NA - 1);                    // row of 16 lanes = one env of the pass, trial
  const real ladder_sd = (real)exp2((double)2 - 2.0 * (double)tl);       // alpha_l * gn = 4^(1 - l)
  const real ladder_qn = (real)exp2((double)2 - (double)tl);             // alpha_l = 2^(2 - l)

  for (int it = 0; it < A.iters; ++it) {
    if (__builtin_amdgcn_readfirstlane((int)__builtin_popcountll(__ballot(active))) == 0) break;
    // ---- 1. lane == env: forward rollout (states to LDS), reverse adjoint sweep, direction to LDS ---------
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
        // of the function the line search evaluates
        Sys::template rhs<real, true>(pre_e, x, u, d);
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          x[c] = fma_r(h, d[c], x[c]);
          sX[(k * DS + c) * G + lane] = x[c];
        }
      }
      real lam[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) lam[c] = 0;
      for (int k = N - 1; k >= 0; --k) {
        const real gk = sg[k];
        real u[DU], xk[DS], g[DU], gy[DS];
#pragma unroll
        for (int c = 0; c < DU; ++c) u[c] = ue[k * DU + c];
#pragma unroll
        for (int c = 0; c < DS; ++c) xk[c] = (k >= 1) ? sX[(k * DS + c) * G + lane] : xse[c];
        if (!GENERIC) {
#pragma unroll
          for (int c = 0; c < DU; ++c) g[c] = gk * (real)2 * P.R1d[DS + c] * u[c];
#pragma unroll
          for (int c = 0; c < DS; ++c) gy[c] = gk * (real)2 * P.R1d[c] * (TGT ? xk[c] - P.target[c] : xk[c]);
        } else {
          real yk[DS], chi[NCHI];
#pragma unroll
          for (int c = 0; c < DS; ++c) yk[c] = (k >= 1) ? xk[c] : y0e[c];  // y_0 is the observation
          make_chi<DS, DU, TGT, real>(P, yk, u, chi);
          if (mode == RCG_MODE_SQL || (mode == RCG_MODE_RQL && k == N - 1)) {  // wave-uniform
            critic_grad_with<DS, DU, real>(chi, yk, u, w_mine, cs, gy, g);
          } else {
            real gc[NCHI];
            stage_grad_with<NCHI, real>(P, chi, sk, gk, gc);
#pragma unroll
            for (int c = 0; c < DS; ++c) gy[c] = gc[c];
#pragma unroll
            for (int c = 0; c < DU; ++c) g[c] = gc[DS + c];
          }
        }
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
          for (int c = 0; c < DS; ++c) lamk[c] += gy[c];
        }
#pragma unroll
        for (int c = 0; c < DS; ++c) lam[c] = lamk[c];
#pragma unroll
        for (int c = 0; c < DU; ++c) sGc[(k * DU + c) * G + lane] = g[c];
      }
    }
    wave_lds_sync();
    // ---- 1b. FOUR LANES PER ENV (round 5): env e = lanes 4 e .. 4 e + 3, coordinate i of the sequence in lane i mod 4.  The
    // completion of the pending (s, y) pair, the free set, the two-loop recursion and the direction are sums and updates over
    // the R coordinates; on lane == env they ran on 16 of the 64 lanes (the critic modes' default, 4 pairs: 0.48 of the issue
    // slots).  Dot products: each lane sums ITS coordinates in index order, the quad adds (p0 + p1) + (p2 + p3) - the same
    // bits in its four lanes; oracle/rcg_oracle.py::actor_optimize_single associates its sums the same way.
    if constexpr (!PAIRS) {  // no pairs (MPC with a diagonal stage cost): free set and box-scaled steepest descent stay on lane == env
      if (active) {  // (the quad form's hand-over - six shuffles and a barrier - costs more than these two short loops: 0.144 ->
        const real* const ue = su + lane * R;  // 0.160 ms per C2 tick when it served this case too)
        gn = 0;
        for (int i = 0; i < R; ++i) {
          const int c = i % DU;
          const real ui = ue[i], gi = sGc[i * G + lane];
          const bool held = (ui <= P.lo[c] && gi > (real)0) || (ui >= P.hi[c] && gi < (real)0);
          const real dc = held ? (real)0 : gi * w2[c];
          sd[lane * R + i] = dc;
          const real m = (dc < 0 ? -dc : dc) / w[c];
          gn = m > gn ? m : gn;
        }
        quasi = false;
        if (!(gn > (real)0) || !finite_r<real>(gn)) active = false;
      }
    } else {
      const int qe = lane >> 2, qq = lane & 3;  // env and part of this lane in the quad view (qe < 16 always)
      const bool act_q = __shfl((int)active, qe, 64) != 0;
      int head_q = __shfl(head, qe, 64), np_q = __shfl(n_pairs, qe, 64);
      const bool pend_q = __shfl((int)pending, qe, 64) != 0;
      bool quasi_q = false;
      real gn_q = 0;
      if (act_q) {
        const real* const ue = su + qe * R;
        // the pair of the last accepted step: its slot holds the gradient at the step's start point
        if (pend_q) {
          for (int i = qq; i < R; i += 4) {
            real* const yp = sLY + ((size_t)head_q * R + i) * G + qe;
            *yp = sGc[i * G + qe] - *yp;
          }
          head_q = head_q + 1 == M ? 0 : head_q + 1;
          np_q = np_q + 1 < M ? np_q + 1 : M;
        }
        // free set: a coordinate on a bound whose descent direction leaves the box is held (this lane's coordinates)
        unsigned long long fm = 0ull;
        for (int i = qq; i < R; i += 4) {
          const int c = i % DU;
          const real ui = ue[i], gi = sGc[i * G + qe];
          const bool held = (ui <= P.lo[c] && gi > (real)0) || (ui >= P.hi[c] && gi < (real)0);
          if (!held) fm |= 1ull << i;
        }
        quasi_q = np_q > 0;
        if (quasi_q) {  // L-BFGS two-loop recursion over the pairs restricted to the free set
          for (int i = qq; i < R; i += 4) sQ[qe * R + i] = ((fm >> i) & 1ull) ? sGc[i * G + qe] : (real)0;
          real a_t[OPT_MAXM], sy_t[OPT_MAXM];
          unsigned okm = 0u;
          real scale = 1;
#pragma unroll
          for (int t = 0; t < OPT_MAXM; ++t) {  // newest -> oldest
            a_t[t] = 0;
            sy_t[t] = 1;
            if (t < np_q) {
              int j = head_q - 1 - t;
              if (j < 0) j += M;
              const real* const Sj = sLS + (size_t)j * R * G + qe;
              const real* const Yj = sLY + (size_t)j * R * G + qe;
              real sy = 0, ss = 0, yy = 0, sq = 0, yhy = 0;
              for (int i = qq; i < R; i += 4)
                if ((fm >> i) & 1ull) {
                  const real s_ = Sj[i * G], y_ = Yj[i * G];
                  sy = fma_r(s_, y_, sy);
                  ss = fma_r(s_, s_, ss);
                  yy = fma_r(y_, y_, yy);
                  sq = fma_r(s_, sQ[qe * R + i], sq);
                  yhy = fma_r(y_ * w2[i % DU], y_, yhy);
                }
              sy = opt_quad_sum(sy);
              ss = opt_quad_sum(ss);
              yy = opt_quad_sum(yy);
              sq = opt_quad_sum(sq);
              yhy = opt_quad_sum(yhy);
              const bool ok = sy > (real)0 && sy * sy > (real)1e-24 * (ss * yy);
              if (t == 0 && ok && yhy > (real)0) scale = sy / yhy;
              sy_t[t] = sy;
              if (ok) {
                okm |= 1u << t;
                const real a = sq / sy;
                a_t[t] = a;
                for (int i = qq; i < R; i += 4)
                  if ((fm >> i) & 1ull) sQ[qe * R + i] = fma_r(-a, Yj[i * G], sQ[qe * R + i]);
              }
            }
          }
          for (int i = qq; i < R; i += 4) sQ[qe * R + i] = (scale * w2[i % DU]) * sQ[qe * R + i];
#pragma unroll
          for (int t = OPT_MAXM - 1; t >= 0; --t) {  // oldest -> newest
            if (t < np_q && ((okm >> t) & 1u)) {
              int j = head_q - 1 - t;
              if (j < 0) j += M;
              const real* const Sj = sLS + (size_t)j * R * G + qe;
              const real* const Yj = sLY + (size_t)j * R * G + qe;
              real yr = 0;
              for (int i = qq; i < R; i += 4)
                if ((fm >> i) & 1ull) yr = fma_r(Yj[i * G], sQ[qe * R + i], yr);
              yr = opt_quad_sum(yr);
              const real cf = a_t[t] - yr / sy_t[t];
              for (int i = qq; i < R; i += 4)
                if ((fm >> i) & 1ull) sQ[qe * R + i] = fma_r(Sj[i * G], cf, sQ[qe * R + i]);
            }
          }
          real dg = 0;
          for (int i = qq; i < R; i += 4) dg = fma_r(sQ[qe * R + i], sGc[i * G + qe], dg);
          dg = opt_quad_sum(dg);
          if (!(dg > (real)0) || !finite_r<real>(dg)) {  // not a descent direction: drop the memory
            quasi_q = false;
            np_q = 0;
          }
        }
        for (int i = qq; i < R; i += 4) {
          const int c = i % DU;
          const real dc = quasi_q ? sQ[qe * R + i] : (((fm >> i) & 1ull) ? sGc[i * G + qe] * w2[c] : (real)0);
          sd[qe * R + i] = dc;
          const real m = (dc < 0 ? -dc : dc) / w[c];
          gn_q = m > gn_q ? m : gn_q;
        }
        gn_q = opt_quad_max(gn_q);
      }
      // back to lane == env: env e's results sit in lane 4 e
      const int src = (lane < G ? lane : 0) * 4;
      const int head_b = __shfl(head_q, src, 64), np_b = __shfl(np_q, src, 64), quasi_b = __shfl((int)quasi_q, src, 64);
      const real gn_b = __shfl(gn_q, src, 64);
      if (active) {
        head = head_b;
        n_pairs = np_b;
        pending = false;
        quasi = quasi_b != 0;
        gn = gn_b;
        if (!(gn > (real)0) || !finite_r<real>(gn)) active = false;
      }
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
      const int quasi_e = __shfl((int)quasi, es, 64);
      const int head_e = __shfl(head, es, 64);
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
      const real alpha = quasi_e ? ladder_qn : ((real)1 / gn_e) * ladder_sd;
      real bj = inf_r<real>();
      if (on) {
        const real J = cost_of(ue, de, y0, xs, pre, alpha, [&](int i) -> real { return sW[i * G + es]; });
        bj = (J != J) ? inf_r<real>() : J;
      }
      int bi = tl;
      row16_argmin(bj, bi);
      const bool better = on && (bj < Jinc_e);  // row-uniform
      // accept: u_e <- clip(u_e - alpha_best d_e); otherwise env e drops its memory or is done
      const real abest = quasi_e ? (real)exp2((double)2 - (double)bi) : ((real)1 / gn_e) * (real)exp2((double)2 - 2.0 * (double)bi);
      wave_lds_sync();  // every lane has finished reading its u_e
      if (better) {
        for (int i = tl; i < R; i += OPT_NA) {
          const int c = i % DU;
          const real uo = ue[i];
          const real un = clamp_r<real>(fma_r(-abest, de[i], uo), P.lo[c], P.hi[c]);
          ue[i] = un;
          if (M > 0) {  // open the pair of this step: s now, y = (next gradient) - (this gradient) in phase 1
            sLS[((size_t)head_e * R + i) * G + es] = un - uo;
            sLY[((size_t)head_e * R + i) * G + es] = sGc[i * G + es];
          }
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
          if (Jinc - bj_me <= A.ftol) active = false;  // the step is kept and it was the last one (ftol = 0: never - a kept step has a positive gain)
          Jinc = bj_me;
          ++used;
          pending = M > 0;
        } else if (quasi) {
          n_pairs = 0;  // retry from the same point with steepest descent
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
    if (A.accum) A.accum[be] = accum_update<Sys, TGT, real>(P, y0e, a, A.accum[be]);
    if (A.step_idx) A.step_idx[be] += 1;
    if constexpr (LOOP) loop_tail<Sys, real>(A.loop, P, be, a, y0e, (double)Jinc);
  }
}

}  // namespace rcg
