// rcg_actor_dma_packed.hpp - k_actor_dma_packed: k_actor_dma's data path for FEW candidates per env
// (CtrlOptPred._actor_cost for K candidates per env + argmin + tick epilogue; controllers.py:1273-1427).
//
// k_actor_dma gives every env tiles of its own: at K <= 32 a tile would be at least half empty (K = 16: a quarter of the lanes
// work, the wave moves 1.3 KB per round trip to HBM).  Here a tile of 64 rows holds G = 64 / K consecutive envs - their
// rows are contiguous in the [B][K][N][du] tensor - so the DMA, the row read and the rollout are exactly k_actor_dma's,
// and what changes is per-lane bookkeeping:
//   * lane l rolls out row (l mod K) of env (l div K) of the tile: the env's state is a per-lane load (lanes of one env
//     read the same address), requested one tile ahead like k_actor_dma's;
//   * the argmin is segmented (ties -> lower candidate index, NaN = +inf, as everywhere): for K = 4, 8, 16, 32 a butterfly
//     inside aligned groups of K lanes (DPP quad / row permutes; one ds_bpermute across two rows), otherwise one masked
//     wave argmin per env of the tile; the winner's first action and its env's observation are gathered from the
//     winner's lane with ds_bpermute;
//   * a wave owns gpw = G * 2^n <= 64 consecutive envs and parks env (env0 + i)'s results in lane i; one coalesced store
//     per field when the wave is done.
//   * RQL / SQL (round 4): the critic weights of the lane's env are per-lane loads as well ([dc][B]: the lanes of one env
//     read one address), requested when the tile's rows have been read - BEFORE the next tile's DMA, so that waiting for
//     them does not drain the tile - and first used at the last step of the rollout (RQL: Q_w of the last step; SQL: the
//     regressor summed over the horizon, dotted with the weights once), so their latency hides behind the rollout.
//     Instances exist for <= 36 dwords of weights (every structure in f32; f64: up to 18 weights - the robots'
//     quad-lin / quadratic structures in f64 stay on k_actor_dma's ragged tile / k_actor).
// Shapes: streamed candidates, 4 <= K <= 32 with K * R * esz % 16 == 0 (whole 16-byte pieces per env), MPC (gamma == 1 per-component
// instance, discounted instance), RQL and SQL x 4 critic structures, diagonal quadratic stage cost, the preset's observation
// target, rows of <= 40 reals, f32 and f64.
// Until round 3 these shapes ran on k_actor (tile HBM -> VGPR -> LDS, row walked from LDS with a runtime horizon):
// 3.4-3.7 TB/s at K = 16 / 32 (B = 65536, Nactor = 10).
#pragma once
#include "rcg_actor_dma.hpp"

namespace rcg {

// argmin of (cost, index) inside aligned groups of K lanes, K a power of two <= 32: lower cost wins, ties -> lower index;
// every lane ends with its group's winner.  Steps of 1, 2 (quad permutes), 4 (row_half_mirror), 8 (row_mirror) are DPP;
// the step across two rows of 16 is a ds_bpermute.
__device__ __forceinline__ void seg_argmin_pow2(float& J, int& I, int K) {
  unsigned long long k = ((unsigned long long)float_order_key(J) << 32) | (unsigned)I;
#define RCG_DPP_MIN(CTRL)                                                                                        \
  {                                                                                                              \
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xF, 0xF, false);         \
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, false); \
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                            \
    k = o < k ? o : k;                                                                                           \
  }
  RCG_DPP_MIN(0xB1)                 // quad_perm [1,0,3,2]
  RCG_DPP_MIN(0x4E)                 // quad_perm [2,3,0,1]
  if (K >= 8) RCG_DPP_MIN(0x141)    // row_half_mirror
  if (K >= 16) RCG_DPP_MIN(0x140)   // row_mirror
#undef RCG_DPP_MIN
  if (K >= 32) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)k, 16, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(k >> 32), 16, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    k = o < k ? o : k;
  }
  J = float_from_order_key((unsigned)(k >> 32));
  I = (int)(unsigned)k;
}
__device__ __forceinline__ void seg_argmin_pow2(double& J, int& I, int K) {
  for (int m = 1; m < K; m <<= 1) {
    const double oJ = __shfl_xor(J, m, 64);
    const int oI = __shfl_xor(I, m, 64);
    if ((oJ < J) || (oJ == J && oI < I)) {
      J = oJ;
      I = oI;
    }
  }
}

// critic variants with per-lane weights exist for at most 36 dwords of them (a second copy of the SQL regressor sums sits
// next to them in registers)
__host__ __device__ constexpr bool packed_critic_ok(int dc, int esz) { return dc * esz <= 144; }

template <typename Sys, typename real, int R, bool TGT, int V>
__global__ __launch_bounds__(256) void k_actor_dma_packed(const ActorArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, NP = Sys::NP;
  constexpr bool G1 = V == DMA_MPC_G1, SQL = V >= DMA_SQL_0, RQL = V >= DMA_RQL_0 && !SQL, CRIT = RQL || SQL;
  constexpr int CS = SQL ? V - DMA_SQL_0 : (RQL ? V - DMA_RQL_0 : 0);  // compile-time critic structure
  constexpr int DC = CRIT ? dma_dc(CS, DS, DU) : 1;
  constexpr int ESZ = (int)sizeof(real);
  static_assert(!CRIT || packed_critic_ok(DC, ESZ), "per-lane critic weights: at most 36 dwords");
  static_assert(R % DU == 0 && R >= DU && R <= 40, "row = N*du reals, at most 40");
  constexpr int N = R / DU;
  constexpr int TILE = 64 * R * ESZ;                               // bytes of a full tile
  constexpr int NFULL = TILE / 1024, NREM = (TILE % 1024) / 256;  // 1-KiB and 256-B direct-to-LDS loads per tile
  static_assert(NFULL * 1024 + NREM * 256 == TILE, "a tile is a whole number of 256-B segments");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const int K = A.K;
  const int G = A.G;  // envs per tile = 64 / K (>= 2)
  const long env0 = wave * A.gpw;
  if (env0 >= B) return;
  const long env1 = env0 + A.gpw < B ? env0 + A.gpw : B;
  const int n_env = (int)(env1 - env0);
  const int n_tiles = (n_env + G - 1) / G;
  const int le = lane / K, lk = lane - le * K;  // my env within the tile, my candidate within the env

  unsigned char* const tile = smem_raw + (size_t)wave_in_wg * TILE;
  const size_t row_bytes = (size_t)R * ESZ;
  const unsigned char* const slab = reinterpret_cast<const unsigned char*>(A.cand) + (size_t)env0 * K * row_bytes;
  // operator mode: the costs of all envs of the wave are staged in LDS (gpw * K <= 64 * 32 reals) and written once
  real* const jstage = reinterpret_cast<real*>(smem_raw + (size_t)4 * TILE) + (size_t)wave_in_wg * (A.gpw * K);

  auto envs_in = [&](int j) -> int { return (j + 1) * G <= n_env ? G : n_env - j * G; };  // wave-uniform
  // valid bytes of tile j (a multiple of 16: the launcher's condition); lanes beyond them load nothing, the LDS keeps stale rows
  auto issue_tile = [&](int j) {
    const unsigned char* const g = slab + (size_t)j * G * K * row_bytes;
    const int vb = envs_in(j) * K * (int)row_bytes;
#pragma unroll
    for (int i = 0; i < NFULL; ++i)
      if (i * 1024 + lane * 16 < vb)
        __builtin_amdgcn_global_load_lds((glb_void*)(g + i * 1024 + lane * 16), (lds_void*)(tile + i * 1024), 16, 0,
                                         RCG_DMA_AUX);
#pragma unroll
    for (int i = 0; i < NREM; ++i)
      if (NFULL * 1024 + i * 256 + lane * 4 < vb)
        __builtin_amdgcn_global_load_lds((glb_void*)(g + NFULL * 1024 + i * 256 + lane * 4),
                                         (lds_void*)(tile + NFULL * 1024 + i * 256), 4, 0, RCG_DMA_AUX);
  };

  // (MPC tick) the env step of the tick, fused: lane == env for the wave's n_env <= 64 envs - k_sim's code on k_sim's data,
  // ONE RK4 per wave whatever the number of tiles - and the new states (and the states before the last substep, for the
  // reference's loop order) parked in LDS, from where the lanes of every tile fetch their env's.  Saves the k_sim launch and
  // the dispatch gap behind it: 3 us of a 14-20 us tick at K = 8 .. 16.
  const bool fused = A.sim_n_sub > 0;  // wave-uniform
  real* const sst = reinterpret_cast<real*>(smem_raw + (size_t)4 * TILE) + (size_t)wave_in_wg * (2 * DS * 64);
  real y0[DS], yn[DS], x0[DS], xn[DS], pn[NP > 0 ? NP : 1];
  const bool lag = A.state_sys != A.obs;  // wave-uniform
#pragma unroll
  for (int c = 0; c < DS; ++c) yn[c] = xn[c] = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) pn[i] = P.pars[i];
  auto fetch_env = [&](int j) {  // the state of MY env of tile j (lanes without a row request nothing)
    if (le < envs_in(j)) {
      const long b = env0 + (long)j * G + le;
      if (fused) {
        const int el = j * G + le;
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          yn[c] = sst[c * 64 + el];
          xn[c] = sst[(DS + c) * 64 + el];
        }
      } else {
#pragma unroll
        for (int c = 0; c < DS; ++c) yn[c] = A.obs[(long)c * B + b];
        if (lag) {
#pragma unroll
          for (int c = 0; c < DS; ++c) xn[c] = A.state_sys[(long)c * B + b];
        }
      }
      if (A.pars_env) {
#pragma unroll
        for (int i = 0; i < NP; ++i) pn[i] = A.pars_env[(long)i * B + b];
      }
    }
  };

  if (fused) {
    // Round 6: the env step's loads are REQUESTED BEFORE the first tile's DMA.  vmcnt retires in issue order: with the tile issued
    // first (round 5) the RK4 could not start before the whole tile had landed; now the wait in front of it covers only the few
    // state loads and the tile travels while the wave steps its envs.
    const bool mine = lane < n_env;
    const long b = env0 + (mine ? lane : 0);
    uint32_t st = A.sim_status[b];
    real x[DS], xp[DS], u[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) xp[c] = x[c] = A.sim_state[(long)c * B + b];
#pragma unroll
    for (int c = 0; c < DU; ++c) u[c] = A.sim_action[(long)c * B + b];
    const auto pre = load_pre<Sys, real>(P, A.pars_env, b);
    issue_tile(0);  // the first tile does not depend on the env step
    if (mine) {
      if (!(st & 1u)) {
        real accum_unused = 0;  // (RCG_FLAG_ACCUM_EVERY_SUBSTEP handles are not fused)
        if (env_substeps<Sys, real, TGT>(P, pre, A.sim_n_sub, x, xp, u, st, accum_unused)) {
#pragma unroll
          for (int c = 0; c < DS; ++c) {
            A.sim_state[(long)c * B + b] = x[c];
            A.sim_state_prev[(long)c * B + b] = xp[c];
          }
        } else {
          A.sim_status[b] = st;  // became non-finite: frozen at its last finite state, nothing else is written
        }
      }
      if (st & 1u) {  // a frozen env keeps STATE and STATE_PREV as they are: the decision sees what k_sim would have left
#pragma unroll
        for (int c = 0; c < DS; ++c) xp[c] = A.sim_state_prev[(long)c * B + b];
      }
#pragma unroll
      for (int c = 0; c < DS; ++c) {
        sst[c * 64 + lane] = x[c];
        sst[(DS + c) * 64 + lane] = xp[c];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    fetch_env(0);  // (from LDS)
  } else {
    fetch_env(0);
    issue_tile(0);  // after the env request: retiring the env state must not drain the tile (vmcnt retires in order)
  }

  const real h = P.h_pred;
  auto pre_env = Sys::template prepare<real>(pn);  // homogeneous parameters: once
  real resJ = 0, resAcc = 0, resU[DU];
  int resI = 0;
#pragma unroll
  for (int c = 0; c < DU; ++c) resU[c] = 0;

  for (int j = 0; j < n_tiles; ++j) {
    const int ne = envs_in(j);
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      y0[c] = yn[c];
      x0[c] = lag ? xn[c] : yn[c];
    }
    if (A.pars_env) pre_env = Sys::template prepare<real>(pn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile j has landed
    real cur[R];
    {
      const real* const myrow = reinterpret_cast<const real*>(tile) + lane * R;
#pragma unroll
      for (int i = 0; i < R; ++i) cur[i] = myrow[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    real wc[DC];  // (RQL / SQL) the critic weights of MY env: requested here, first read at the end of the rollout
    if constexpr (CRIT) {
      const long bw = env0 + (long)j * G + (le < ne ? le : 0);
#pragma unroll
      for (int i = 0; i < DC; ++i) wc[i] = A.w[(long)i * B + bw];
    }
    if (j + 1 < n_tiles) {  // the LDS tile is free: next env states first, then the next tile
      fetch_env(j + 1);
      issue_tile(j + 1);
    }

    // _actor_cost of my row (controllers.py:1284-1326): k_actor_dma's rollout, registers only
    real x[DS], y[DS];
#pragma unroll
    for (int c = 0; c < DS; ++c) {
      x[c] = x0[c];
      y[c] = y0[c];
    }
    real J = 0, gk = 1;
    real S[NCHI];
#pragma unroll
    for (int i = 0; i < NCHI; ++i) S[i] = 0;
    real Phi[SQL ? DC : 1];  // SQL: the regressor summed over the horizon (J = sum_k w . phi_k = w . sum_k phi_k)
#pragma unroll
    for (int i = 0; i < (SQL ? DC : 1); ++i) Phi[i] = 0;
    auto wget = [&](int i) -> real { return wc[CRIT ? i : 0]; };
#pragma unroll
    for (int kk = 0; kk < N; ++kk) {
      if (kk > 0) {
        real d[DS];
        Sys::template rhs<real, true>(pre_env, x, &cur[(kk - 1) * DU], d);
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          x[c] = fma_r(h, d[c], x[c]);
          y[c] = x[c];
        }
      }
      real chi[NCHI];
#pragma unroll
      for (int c = 0; c < DS; ++c) chi[c] = TGT ? y[c] - P.target[c] : y[c];
#pragma unroll
      for (int c = 0; c < DU; ++c) chi[DS + c] = cur[kk * DU + c];
      if (G1) {
#pragma unroll
        for (int i = 0; i < NCHI; ++i) S[i] = fma_r(chi[i], chi[i], S[i]);
      } else if (SQL) {
        critic_phi_accum<DS, DU, real>(chi, y, &cur[kk * DU], Phi, CS);
      } else if (RQL && kk == N - 1) {
        J += critic_with<DS, DU, real>(chi, y, &cur[kk * DU], wget, CS);
      } else {
        J = fma_r(gk, stage_diag<NCHI, real>(P, chi), J);
        gk *= P.gamma;
      }
    }
    if (G1) {
#pragma unroll
      for (int i = 0; i < NCHI; ++i) J = fma_r(P.R1d[i], S[i], J);
    }
    if (SQL) {  // J = w . sum_k phi(chi_k)
#pragma unroll
      for (int i = 0; i < DC; ++i) J = fma_r(wget(i), Phi[i], J);
    }

    const bool has_row = le < ne;
    if (A.J && has_row) jstage[j * G * K + lane] = J;  // rows of a tile are consecutive in J too
    const real Jc = (J != J) ? inf_r<real>() : J;    // NaN counts as +inf

    // segmented argmin: every lane ends with the (cost, index) of ITS env's winner
    real segJ = (le < ne) ? Jc : inf_r<real>();
    int segI = (le < ne) ? lk : 0x7fffffff;
    if ((K & (K - 1)) == 0) {  // K = 4, 8, 16, 32: a butterfly inside aligned groups of K lanes
      seg_argmin_pow2(segJ, segI, K);
    } else {  // one masked wave argmin per env of the tile
      real mJ = segJ;
      int mI = segI;
      for (int g = 0; g < ne; ++g) {
        real kJ = (le == g) ? mJ : inf_r<real>();
        int kI = (le == g) ? mI : 0x7fffffff;
        wave_argmin(kJ, kI);  // all +inf: the lowest index of env g wins (lk = 0), numpy.argmin's answer
        if (le == g) {
          segJ = kJ;
          segI = kI;
        }
      }
    }
    // park env (j*G + g)'s result in lane (j*G + g): gather it from the env's lanes (per-lane source: ds_bpermute)
    const int gi = lane - j * G;                   // the env of the tile this lane parks
    const bool parks = gi >= 0 && gi < ne;
    const int src0 = parks ? gi * K : lane;        // any lane of that env
    const real pJ = __shfl(segJ, src0, 64);
    const int pI = __shfl(segI, src0, 64);
    const int wl = parks ? gi * K + pI : lane;     // the winner's lane
    real bu[DU], yw[DS];
#pragma unroll
    for (int c = 0; c < DU; ++c) bu[c] = __shfl(cur[c], wl, 64);  // the sequence's first action
    real acc_inc = 0;
    if (A.accum) {  // upd_accum_obj (controllers.py:1086-1093) at the env's observation (every lane of the env holds it)
#pragma unroll
      for (int c = 0; c < DS; ++c) yw[c] = __shfl(y0[c], src0, 64);
      real chi[NCHI];
#pragma unroll
      for (int c = 0; c < DS; ++c) chi[c] = TGT ? yw[c] - P.target[c] : yw[c];
#pragma unroll
      for (int c = 0; c < DU; ++c) chi[DS + c] = bu[c];
      acc_inc = stage_diag<NCHI, real>(P, chi) * P.sampling_time;
    }
    if (parks) {
      resJ = pJ;
      resI = pI;
      resAcc = acc_inc;
#pragma unroll
      for (int c = 0; c < DU; ++c) resU[c] = bu[c];
    }
  }

  if (A.J) {  // the wave's costs in one piece
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    real* const Jout = A.J + env0 * K;
    const int n = n_env * K;
    for (int i = lane; i < n; i += 64) Jout[i] = jstage[i];
  }
  if (lane < n_env) {
    const long bb = env0 + lane;
#pragma unroll
    for (int c = 0; c < DU; ++c)
      if (A.action_out) A.action_out[(long)c * B + bb] = resU[c];
    if (A.best_J) A.best_J[bb] = resJ;
    if (A.best_idx) A.best_idx[bb] = resI;
    if (A.accum) atomicAdd(&A.accum[bb], resAcc);
    if (A.step_idx) atomicAdd(&A.step_idx[bb], 1);
  }
}

// instances: rcg_dma_inst.hip, groups 3 (MPC), 4 (SQL), 5 (RQL) - one object per system x element type x group
template <typename Sys, typename real, int GROUP>
bool launch_dma_packed(int r, int variant, dim3 grid, dim3 block, size_t lds, hipStream_t s, const ActorArgs<real>& A,
                       const KParams<real>& P, hipEvent_t ev_a, hipEvent_t ev_b);

}  // namespace rcg
