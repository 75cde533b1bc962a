// rcg_actor_dma.hpp - k_actor_dma: the production kernel of the streamed rollout
// (CtrlOptPred._actor_cost for K candidates per env + argmin + tick epilogue; controllers.py:1273-1427).
//
// Shape it serves: f32, MPC with a diagonal R1 (every reference preset), K a multiple of 64, candidate rows
// of R = N*du <= 32 floats, rollout started from the observation (state_sys == obs: the control tick without
// ref_lag).  Everything else goes to k_actor (rcg_kernels.hpp).
//
//   per tile of 64 candidate rows (256*R bytes, contiguous in HBM):
//     1. direct-to-LDS loads: R/4 x global_load_lds_dwordx4 (64 lanes x 16 B = 1 KiB each) plus R%4 x
//        global_load_lds_dword (256 B each), `nt` (the tensor is read once per tick) - fully coalesced, written
//        by the memory pipeline straight into this wave's LDS tile, no VGPR staging;
//     2. the lane pulls ITS row LDS -> registers (R floats at lane*R*4: ds_read_b128 when R % 4 == 0, b64 when
//        R is even, b32 otherwise; R = 20 is conflict-free, any residual conflict is noise next to the rollout);
//     3. as soon as the row is in registers the SAME LDS tile is free again: the next tile's loads are issued
//        here, before the rollout, so they are in flight during all of step 4;
//     4. the rollout runs on registers only, the horizon fully unrolled (N = R/du is a template constant): no
//        wait of any kind.  Trig: hardware v_sin/v_cos behind an exact reduction (rcg_math.hpp::sincos_hw).
// The only vmcnt wait is the one in front of step 2 of the NEXT tile, which is exactly the data it needs.
// vmcnt retires in issue order, so whatever else the next iteration needs from memory (the next env's state)
// is requested BEFORE the tile loads and never drains them.
//     5. per env: wave argmin on a packed (cost, index) key (DPP + v_readlane, rcg_math.hpp); the winner is parked in
//        lane (env - env0).  When the wave's envs are done, lanes < n_envs store action / best_J / best_idx coalesced
//        and issue ACCUM / STEP_IDX as no-return atomics (one adder per address: still deterministic).  Per-env
//        4-byte writes from lane 0 cost 7 % (scattered partial-line writes interleaved with the read stream).
// No s_barrier anywhere: a wave only reads LDS it filled itself.
// Launch geometry (rcg_sysops.hpp::launch_actor): a wave owns a power-of-two number of consecutive envs, 2 blocks per
// CU resident (4 for rows shorter than 20 floats), grid of several rounds.
// Measured on C2 (B = 65536, K = 256, N = 10): 203 us per launch = 6.6 TB/s (83 % of the 8 TB/s peak); the bare data
// path of this kernel (steps 1-3, no arithmetic) holds 7.0-7.2 TB/s (tools/bw_probe.hip residency).  In a development
// build (`make dev`, -DRCG_DEV -> librcg_dev.so) the A.dbg bits (env RCG_DBG) switch pieces off for such measurements:
// 1 rollout, 2 argmin + writes, 4 env-state loads.  The production library compiles them out (RCG_DBG(A, bit) == 0).
#pragma once
#include "rcg_kernels.hpp"

namespace rcg {

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the immediate must be a constant): n <= 16 here (a tile is at most
// 8 loads, an env-state request at most 7); anything else waits for everything.
__device__ __forceinline__ void wait_vmcnt(int n) {
#define RCG_VMCNT_CASE(k) \
  case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    RCG_VMCNT_CASE(1) RCG_VMCNT_CASE(2) RCG_VMCNT_CASE(3) RCG_VMCNT_CASE(4) RCG_VMCNT_CASE(5) RCG_VMCNT_CASE(6)
    RCG_VMCNT_CASE(7) RCG_VMCNT_CASE(8) RCG_VMCNT_CASE(9) RCG_VMCNT_CASE(10) RCG_VMCNT_CASE(11) RCG_VMCNT_CASE(12)
    RCG_VMCNT_CASE(13) RCG_VMCNT_CASE(14) RCG_VMCNT_CASE(15) RCG_VMCNT_CASE(16)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef RCG_VMCNT_CASE
}

// G1: gamma == 1 (the reference's default and every preset, main_3wrobot.py:147): the discounted sum of weighted
// squares is accumulated per component, S_i += chi_i^2 (one fma per term and step instead of mul + fma + the
// discount bookkeeping) and weighted once at the end, J = sum_i R1_ii S_i.  About 30 VALU ops per step before, 14 of
// them the stage cost; measured +1.2 % on C2 (the kernel is HBM-bound, the VALU work only has to stay out of the way).
// CRIT: mode RQL (instantiated with G1 = false only): the last stage cost is replaced by Q_w(y_{N-1}, u_{N-1})
// (controllers.py:1307-1310) - the horizon is unrolled, so that is a compile-time position.  The env's critic weights
// travel with its state (requested one tile ahead, held in registers).  Measured on configs[2] with streamed
// candidates: 0.474 ms against 0.506 ms on k_actor.  SQL (Q_w at every step, :1311-1326) was tried here as well and
// is 18 % SLOWER than on k_actor, whose rollout is specialised on the critic structure at compile time: it stays there.
template <typename Sys, int R, bool TGT, bool G1, bool CRIT>
__global__ __launch_bounds__(256) void k_actor_dma(const ActorArgs<float> A, const KParams<float> P) {
  typedef float real;
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, NP = Sys::NP;
  constexpr int DCMAX = CRIT ? NCHI * (NCHI + 1) / 2 + NCHI : 1;  // quad-lin, the largest critic structure
  static_assert(!(CRIT && G1), "critic modes use the discounted accumulation");
  static_assert(R % DU == 0 && R >= DU && R <= 32, "row = N*du floats, at most 32");
  constexpr int N = R / DU;
  constexpr int NFULL = R / 4, NREM = R % 4;  // 1-KiB and 256-B direct-to-LDS loads per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int lane = threadIdx.x & 63;
  // readfirstlane: provably wave-uniform, so tile bases live in SGPRs and the control flow is scalar
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const int K = A.K;
  const int T = K >> 6;  // tiles per env
  const long env0 = wave * A.gpw;
  if (env0 >= B) return;
  const long env1 = env0 + A.gpw < B ? env0 + A.gpw : B;
  // 32-bit on purpose: a 64-bit loop compare has no scalar form, and its VGPR temporary once landed in the registers
  // of the in-flight env-state prefetch (a WAW hazard the compiler resolves with s_waitcnt vmcnt(0))
  const int n_tiles = (int)(env1 - env0) * T;

  // this wave's LDS: one tile, or two (A.depth == 2: tile g lives in buffer g & 1 and tile g + 2 is requested into it
  // as soon as tile g's rows are in registers, so two tiles are in flight while the rollout runs)
  const int depth = A.depth == 2 ? 2 : 1;
  unsigned char* const tile0 = smem_raw + (size_t)wave_in_wg * (256 * R) * depth;
  const unsigned char* gb = reinterpret_cast<const unsigned char*>(A.cand) + (size_t)env0 * K * (4 * R);
  // operator mode (rcg_actor_cost): the J of one env is staged in LDS ([K] floats per wave, behind the tiles of all
  // four waves) and written out at the env's end in 1-KiB bursts; a 256-B store after every tile, interleaved with
  // the read stream, made the operator 35 % slower than the tick for 5 % more bytes
  // (A.jwave: the staging area holds all envs of the wave and is written once, when the wave is done)
  const int jspan = A.jwave ? A.gpw * K : K;  // floats of staging per wave
  real* const jstage = reinterpret_cast<real*>(smem_raw + (size_t)4 * (256 * R) * depth) + (size_t)wave_in_wg * jspan;

  auto issue_tile = [&](const unsigned char* g, unsigned char* tile) {
#pragma unroll
    for (int j = 0; j < NFULL; ++j)
      __builtin_amdgcn_global_load_lds((glb_void*)(g + j * 1024 + lane * 16), (lds_void*)(tile + j * 1024), 16, 0,
                                       2 /* nt */);
#pragma unroll
    for (int j = 0; j < NREM; ++j)
      __builtin_amdgcn_global_load_lds((glb_void*)(g + NFULL * 1024 + j * 256 + lane * 4),
                                       (lds_void*)(tile + NFULL * 1024 + j * 256), 4, 0, 2 /* nt */);
  };

  // env state: `n`-suffixed = requested one tile ahead for the next env.  Loads only, no
  // "pointer ? load : default" selects (a default written into a register with a load in flight would force a
  // vmcnt(0) on the spot).
  real y0[DS], yn[DS], pv[NP > 0 ? NP : 1], pn[NP > 0 ? NP : 1], wc[DCMAX], wn[DCMAX];
#pragma unroll
  for (int i = 0; i < NP; ++i) pn[i] = P.pars[i];
#pragma unroll
  for (int i = 0; i < DCMAX; ++i) wn[i] = wc[i] = 0;  // entries >= dc are never loaded and never read
  auto fetch_env = [&](long b) {
    if (RCG_DBG(A, 4)) {  // development: no env-state loads
#pragma unroll
      for (int c = 0; c < DS; ++c) yn[c] = (real)0.5;
      return;
    }
#pragma unroll
    for (int c = 0; c < DS; ++c) yn[c] = A.obs[(long)c * B + b];
    if (A.pars_env) {
#pragma unroll
      for (int i = 0; i < NP; ++i) pn[i] = A.pars_env[(long)i * B + b];
    }
    if (CRIT) {
#pragma unroll
      for (int i = 0; i < DCMAX; ++i)
        if (i < P.dc) wn[i] = A.w[(long)i * B + b];  // wave-uniform branch
    }
  };
  // loads one env-state request issues (vmcnt bookkeeping of the depth-2 pipeline)
  const int n_env_loads =
      (RCG_DBG(A, 4) || A.sim_state != nullptr) ? 0 : DS + (A.pars_env ? NP : 0) + (CRIT ? P.dc : 0);

  // Fused env step (the tick's Simulator.sim_step in this launch, A.sim_state != nullptr): lane e < ne integrates env
  // env0 + e - exactly k_sim's arithmetic (rk4_step, clip, freeze on a non-finite state) - while the first tile is in
  // flight, writes STATE / STATE_PREV / STATUS, and keeps the new state; each env's rollout then starts from
  // v_readlane of that lane instead of a global load.  Saves the k_sim launch and the state's HBM round trip.
  const bool fused = A.sim_state != nullptr;
  real xs[DS], ps[NP > 0 ? NP : 1];
#pragma unroll
  for (int c = 0; c < DS; ++c) xs[c] = 0;
#pragma unroll
  for (int i = 0; i < (NP > 0 ? NP : 1); ++i) ps[i] = NP > 0 ? P.pars[i] : (real)0;
  if (fused) {
    const int ne = (int)(env1 - env0);
    const bool mine = lane < ne;
    const long be = env0 + (mine ? lane : 0);
    real x[DS], xp[DS], u[DU];
#pragma unroll
    for (int c = 0; c < DS; ++c) x[c] = A.sim_state[(long)c * B + be];
#pragma unroll
    for (int c = 0; c < DU; ++c) u[c] = A.sim_action[(long)c * B + be];
    const uint32_t st = A.sim_status[be];
    if (A.pars_env) {
#pragma unroll
      for (int i = 0; i < NP; ++i) ps[i] = A.pars_env[(long)i * B + be];
    }
    issue_tile(gb, tile0);  // after the requests above: using them must not wait for the tile
#pragma unroll
    for (int c = 0; c < DU; ++c) u[c] = P.clip ? clamp_r<real>(u[c], P.lo[c], P.hi[c]) : u[c];  // systems.py:241-243
#pragma unroll
    for (int c = 0; c < DS; ++c) xs[c] = xp[c] = x[c];
    if (mine && !(st & 1u)) {  // frozen envs keep their state
      const auto pre = Sys::template prepare<real>(ps);
      for (int s = 0; s < A.sim_nsub; ++s) {
#pragma unroll
        for (int c = 0; c < DS; ++c) xp[c] = x[c];
        rk4_step<Sys, real>(pre, x, u, P.dt_sim);
      }
      bool ok = true;
#pragma unroll
      for (int c = 0; c < DS; ++c) ok = ok && finite_r<real>(x[c]);
      if (!ok) {  // freeze the env at its last finite state and flag it (as k_sim)
        A.sim_status[be] = st | 1u;
      } else {
#pragma unroll
        for (int c = 0; c < DS; ++c) {
          A.sim_state[(long)c * B + be] = x[c];
          A.sim_state_prev[(long)c * B + be] = xp[c];
          xs[c] = x[c];
        }
      }
    }
  } else {
    fetch_env(env0);
    issue_tile(gb, tile0);  // after the env request: retiring the env state must not drain the first tile
  }
  if (depth == 2 && n_tiles > 1) issue_tile(gb + 256 * R, tile0 + 256 * R);  // T >= 2: same env

  const real h = P.h_pred;
  long b = env0;
  int t = 0;
  auto pre_env = Sys::template prepare<real>(pn);
  real bestJ = inf_r<real>();
  int bestI = 0x7fffffff;
  real bu[DU];
#pragma unroll
  for (int c = 0; c < DU; ++c) bu[c] = 0;
  // results of env (env0 + j) wait in lane j and are written once, coalesced, when the wave is done: per-env 4-byte
  // stores from lane 0 were 6 scattered partial-line writes per env, interleaved with the read stream (measured: 7 %)
  real resJ = 0, resAcc = 0, resU[DU];
  int resI = 0;
#pragma unroll
  for (int c = 0; c < DU; ++c) resU[c] = 0;

  for (int g = 0; g < n_tiles; ++g) {
    if (t == 0) {  // first tile of env b: adopt the state requested one tile ago / integrated in the prologue
      if (fused) {
        const int e = __builtin_amdgcn_readfirstlane((int)(b - env0));
#pragma unroll
        for (int c = 0; c < DS; ++c)
          y0[c] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(xs[c]), e));
#pragma unroll
        for (int i = 0; i < NP; ++i)
          pv[i] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(ps[i]), e));
      } else {
#pragma unroll
        for (int c = 0; c < DS; ++c) y0[c] = yn[c];
#pragma unroll
        for (int i = 0; i < NP; ++i) pv[i] = pn[i];
      }
      pre_env = Sys::template prepare<real>(pv);
      if (CRIT) {
#pragma unroll
        for (int i = 0; i < DCMAX; ++i) wc[i] = wn[i];
      }
      bestJ = inf_r<real>();
      bestI = 0x7fffffff;
    }
    // 2. tile g has landed -> my row into registers.  vmcnt retires in issue order: depth 1 waits for everything;
    //    depth 2 lets what was issued AFTER tile g stay in flight: tile g + 1, preceded by an env-state request iff
    //    tile g + 1 opens the next env (t == T - 1)
    if (depth == 2 && g + 1 < n_tiles)
      wait_vmcnt(NFULL + NREM + (t == T - 1 ? n_env_loads : 0));
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned char* const tile = tile0 + ((depth == 2 && (g & 1)) ? 256 * R : 0);
    const real* const myrow = reinterpret_cast<const real*>(tile) + lane * R;
    real cur[R];
#pragma unroll
    for (int i = 0; i < R; ++i) cur[i] = myrow[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane's row is out of LDS (in-order per wave)
    __builtin_amdgcn_wave_barrier();
    // 3. this LDS buffer is free: request tile g + depth into it, an env-state request first if it opens an env
    //    (one code path for both depths: two sites would be merged by the compiler with register copies of the
    //    prefetched state, i.e. with a wait for it)
    gb += 256 * R;
    if (g + depth < n_tiles) {
      if (t == T - depth && !fused) fetch_env(b + 1);  // tile g + depth opens env b + 1 (depth 2: T >= 2)
      issue_tile(gb + (depth - 1) * (256 * R), tile);
    }

    // 4. _actor_cost of this lane's row (controllers.py:1284-1306), registers only
    real x[DS], y[DS];
#pragma unroll
    for (int c = 0; c < DS; ++c) x[c] = y[c] = y0[c];  // state_sys == observation (see the launcher)
    real J = 0, gk = 1;
    real S[NCHI];
#pragma unroll
    for (int i = 0; i < NCHI; ++i) S[i] = 0;
    if (RCG_DBG(A, 1)) {  // timing-only variant (RCG_DBG=1): consume the row, skip the rollout
#pragma unroll
      for (int i = 0; i < R; ++i) J += cur[i];
    } else {
#pragma unroll
      for (int kk = 0; kk < N; ++kk) {
        if (kk > 0) {
          real d[DS];
          Sys::template rhs<real, true>(pre_env, x, &cur[(kk - 1) * DU], d);  // unclipped: sys_rhs([], state, u[k-1])
#pragma unroll
          for (int c = 0; c < DS; ++c) {
            x[c] = fma_r(h, d[c], x[c]);
            y[c] = x[c];  // sys_out is the identity
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
        } else if (CRIT && kk == N - 1) {
          J += critic_with<DS, DU, real>(chi, y, &cur[kk * DU], [&](int i) -> real { return wc[i]; }, P.critic_struct);
        } else {
          J = fma_r(gk, stage_diag<NCHI, real>(P, chi), J);
          gk *= P.gamma;
        }
      }
      if (G1) {
#pragma unroll
        for (int i = 0; i < NCHI; ++i) J = fma_r(P.R1d[i], S[i], J);
      }
    }

    const int k = t * 64 + lane;
    if (A.J) jstage[(A.jwave ? (int)(b - env0) * K : 0) + k] = J;
    const real Jc = (J != J) ? inf_r<real>() : J;  // NaN counts as +inf
    if (Jc < bestJ || bestI == 0x7fffffff) {
      bestJ = Jc;
      bestI = k;
#pragma unroll
      for (int c = 0; c < DU; ++c) bu[c] = cur[c];  // the sequence's first action
    }

    // env b complete (per-env staging) or wave complete (A.jwave): the staged costs go out in one piece
    if (A.J && (A.jwave ? g == n_tiles - 1 : t == T - 1)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      real* const Jout = A.J + (A.jwave ? env0 : b) * K;
      const int n = A.jwave ? (int)(env1 - env0) * K : K;  // a multiple of 64
      if ((n & 255) == 0) {
        for (int i = lane * 4; i < n; i += 256) {
          const v4f v = *reinterpret_cast<const v4f*>(jstage + i);
          *reinterpret_cast<v4f*>(Jout + i) = v;
        }
      } else {
        for (int i = lane; i < n; i += 64) Jout[i] = jstage[i];
      }
    }
    if (++t == T) {  // env b complete: wave argmin (lower J, then lower index) + tick epilogue
      if (RCG_DBG(A, 2)) {  // development: no argmin / stores (one store keeps the work alive)
        if (bestJ == (real)-12345.678f) A.best_J[b] = bestJ;
        if (lane == 0 && A.step_idx) atomicAdd(&A.step_idx[b], 1);
        t = 0;
        ++b;
        continue;
      }
      // packed (cost, index) key, DPP + readlane reduction (rcg_math.hpp); the winner's first action is read from the
      // winner's lane: every lane kept the action of its own best row, and bestI = tile * 64 + lane
      const unsigned long long wkey =
          wave_min_u64(((unsigned long long)float_order_key(bestJ) << 32) | (unsigned)bestI);
      bestJ = float_from_order_key((unsigned)(wkey >> 32));
      bestI = (int)(unsigned)wkey;
      const int wl = __builtin_amdgcn_readfirstlane(bestI & 63);
#pragma unroll
      for (int c = 0; c < DU; ++c) bu[c] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(bu[c]), wl));
      real acc_inc = 0;
      if (A.accum) {  // upd_accum_obj (controllers.py:1086-1093), wave-uniform
        real chi[NCHI];
#pragma unroll
        for (int c = 0; c < DS; ++c) chi[c] = TGT ? y0[c] - P.target[c] : y0[c];
#pragma unroll
        for (int c = 0; c < DU; ++c) chi[DS + c] = bu[c];
        acc_inc = stage_diag<NCHI, real>(P, chi) * P.sampling_time;
      }
      if (lane == (int)(b - env0)) {
        resJ = bestJ;
        resI = bestI;
        resAcc = acc_inc;
#pragma unroll
        for (int c = 0; c < DU; ++c) resU[c] = bu[c];
      }
      t = 0;
      ++b;
    }
  }

  // one coalesced write per field for the envs of this wave; stores and no-return atomics only
  if (lane < (int)(env1 - env0) && !RCG_DBG(A, 2)) {
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

}  // namespace rcg
