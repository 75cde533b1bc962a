// rcg_actor_dma.hpp - k_actor_dma: the production kernel of the streamed rollout
// (CtrlOptPred._actor_cost for K candidates per env + argmin + tick epilogue; controllers.py:1273-1427).
//
// Shape it serves: diagonal quadratic stage cost (every reference preset), K >= 33 with K * R * esz a multiple of 16 (an env's last
// tile may be ragged: its direct-to-LDS loads are masked per lane, so not a byte beyond the env's rows is read, and the lanes
// without a row sit out the argmin), the observation target as
// the system's preset has it; the rollout starts from `state_sys` with `obs` as y_0 (controllers.py:1286-1296) - the same
// vector in the plain tick, the state before the last substep with RCG_FLAG_REF_LAG (the reference's loop order);
// candidate rows of R = N*du <= 40 reals; modes MPC, RQL and SQL in f32 and in f64 (the reference's own arithmetic width).
// Everything else goes to k_actor (rcg_kernels.hpp).
//
//   per tile of 64 candidate rows (64*R*sizeof(real) bytes, contiguous in HBM):
//     1. direct-to-LDS loads: global_load_lds_dwordx4 (64 lanes x 16 B = 1 KiB each) plus global_load_lds_dword (256 B
//        each) for the remainder, `nt` (the tensor is read once per tick) - fully coalesced, written by the memory
//        pipeline straight into this wave's LDS tile, no VGPR staging;
//     2. the lane pulls ITS row LDS -> registers (R reals at lane*R*sizeof(real): ds_read_b128 / b64 / b32 as the
//        alignment allows; R = 20 floats is conflict-free, any residual conflict is noise next to the rollout);
//     3. as soon as the row is in registers the SAME LDS tile is free again: the next tile's loads are issued
//        here, before the rollout, so they are in flight during all of step 4;
//     4. the rollout runs on registers only, the horizon fully unrolled (N = R/du is a template constant): no
//        wait of any kind.  Trig: f32 hardware v_sin/v_cos behind an exact reduction (rcg_math.hpp::sincos_hw); f64 a
//        Cody-Waite reduction + minimax polynomials (sincos_fast, ~40 VALU ops against libm's several hundred).
// The only vmcnt wait is the one in front of step 2 of the NEXT tile, which is exactly the data it needs.
// vmcnt retires in issue order, so whatever else the next iteration needs from memory (the next env's state)
// is requested BEFORE the tile loads and never drains them.
//     5. per env: wave argmin (f32: packed (cost, index) key, DPP + v_readlane, rcg_math.hpp; f64: shuffle butterfly);
//        the winner is parked in lane (env - env0).  When the wave's envs are done, lanes < n_envs store action / best_J
//        / best_idx coalesced and issue ACCUM / STEP_IDX as no-return atomics (one adder per address: still
//        deterministic).  Per-env 4-byte writes from lane 0 cost 7 % (scattered partial-line writes interleaved with the
//        read stream).
// No s_barrier anywhere: a wave only reads LDS it filled itself.
// Launch geometry (rcg_sysops.hpp::launch_actor): a wave owns a power-of-two number of consecutive envs, 2 blocks per
// CU resident (4 for rows shorter than 80 bytes), grid of several rounds.
// Measured on C2 (B = 65536, K = 256, N = 10, f32): 202-203 us per launch = 6.6 TB/s (83 % of the 8 TB/s peak); the
// bare data path of this kernel (steps 1-3, no arithmetic) holds 7.0-7.2 TB/s (tools/bw_probe.hip residency).  In a
// development build (`make dev`, -DRCG_DEV -> librcg_dev.so) the A.dbg bits (env RCG_DBG) switch pieces off for such
// measurements: 1 rollout, 2 argmin + writes, 4 env-state loads.  The production library compiles them out.
// Tried and removed: two tiles in flight per wave (1.5-3 % slower), the tick's env step fused into the
// prologue (a wash: +3-4 % kernel time against one saved 7-us launch).
#pragma once
#include <hip/hip_ext.h>
#include "rcg_kernels.hpp"

// cache policy of the direct-to-LDS tile loads (the aux / cpol immediate of global_load_lds: 1 = sc0, 2 = nt, 16 = sc1):
// the candidate tensor is read once per tick -> nt.  Measured on C2 (tools/knob_sweep.py against libraries built with
// -DRCG_DMA_AUX=...): see DESIGN.md 4.
#ifndef RCG_DMA_AUX
#define RCG_DMA_AUX 2
#endif

namespace rcg {

// Variants of the cost accumulation, one kernel instance each (the horizon is unrolled, so each is straight-line code):
//   DMA_MPC_G1  MPC, gamma == 1 (the reference's default and every preset, main_3wrobot.py:147): the sum of weighted
//               squares is accumulated per component, S_i += chi_i^2 (one fma per term and step instead of mul + fma + the
//               discount bookkeeping) and weighted once at the end, J = sum_i R1_ii S_i (measured +1.2 % on C2)
//   DMA_MPC     MPC, discounted (controllers.py:1304-1306)
//   DMA_RQL_*   the last stage cost is replaced by Q_w(y_{N-1}, u_{N-1}) (controllers.py:1307-1310)
//   DMA_SQL_*   J = sum_k Q_w(y_k, u_k), undiscounted (controllers.py:1311-1326)
//               The critic structure is a compile-time constant of both (+ rcg_critic_struct).  Rounds 1-2 served RQL with
//               ONE instance and a wave-uniform runtime switch at the last step: compiled for the largest structure (35
//               weights on Sys3WRobot) with the code of all four inline, it needed 183-256 VGPRs (occupancy 2, 1 at rows of
//               40 floats) and lost 10-20 % of the stream whatever structure actually ran.
// The env's critic weights travel with its state (requested one tile ahead).  They are wave-uniform - a wave rolls out one
// env at a time.  Up to 9 of them (the tank; quad-nomix on the robots) are held in registers, 2 x dc (current env, next
// env).  Instances compiled for more (the robots' RQL instance serves every structure: 35 / 20 weights) park them in LDS:
// lane i requests weight i of the next env into ONE register, the wave writes the dc values into a small LDS slot of its
// own when it adopts the env, and the rollouts read them back as broadcast ds_reads (RQL: dc reads per rollout, at the last
// step; SQL: the regressor is summed over the horizon per feature - one fma per feature and step instead of a product and
// an fma - and dotted with the weights once, as k_actor's SQL rollout does).  Round 2 kept 2 x 35 weights in registers:
// the f32 RQL instance of Sys3WRobot needed 221 VGPRs + 149 spilled SGPRs (occupancy 2, 1 at rows of 40 floats) and
// streamed at 0.68-0.75 of the HBM peak, SQL with 28 / 35 weights at 0.43-0.47; the float64 critic modes of the robots
// were left on k_actor (0.69).
//   DMA_MPC_GEND / DMA_MPC_GENF (round 6; MPC with a cost structure no preset has - VERDICT r5 next 6: these ran on k_actor's
//               HBM -> VGPR -> LDS staging at 0.09-0.37 of the HBM peak): GEND = diagonal matrices - the biquadratic stage
//               cost (controllers.py:1079-1082), with or without an observation target; GENF = full
//               R1 (and R2): the symmetrised upper triangles (rcg_kernels.hpp::load_sym, 28 reals each for the robots) are
//               formed once per wave and held in registers across the tile loop.  Both are instantiated with TGT = true (a
//               handle without a target has one of zeros: y - 0 = y exactly); biquadratic or not is a wave-uniform branch.
enum : int {
  DMA_MPC_G1 = 0,
  DMA_MPC = 1,
  DMA_RQL_0 = 2 /* + rcg_critic_struct */,
  DMA_SQL_0 = 6 /* + rcg_critic_struct */,
  DMA_MPC_GEND = 10,
  DMA_MPC_GENF = 11,
  DMA_RQL_GEN_0 = 12 /* + rcg_critic_struct: RQL with a stage cost no preset has (stage_any per step), any target */,
  DMA_VARIANTS = 16
};
__host__ __device__ constexpr bool dma_is_sql(int v) { return v >= DMA_SQL_0 && v < DMA_MPC_GEND; }
__host__ __device__ constexpr bool dma_is_rql(int v) { return (v >= DMA_RQL_0 && v < DMA_SQL_0) || v >= DMA_RQL_GEN_0; }
// critic structure of a critic variant
__host__ __device__ constexpr int dma_cs(int v) {
  return v >= DMA_RQL_GEN_0 ? v - DMA_RQL_GEN_0 : (dma_is_sql(v) ? v - DMA_SQL_0 : (dma_is_rql(v) ? v - DMA_RQL_0 : 0));
}

__host__ __device__ constexpr int dma_dc(int cs, int ds, int du) {
  return cs == RCG_CRITIC_QUAD_LIN ? (ds + du) * (ds + du + 1) / 2 + (ds + du)
                                   : (cs == RCG_CRITIC_QUADRATIC ? (ds + du) * (ds + du + 1) / 2
                                                                 : (cs == RCG_CRITIC_QUAD_NOMIX ? ds + du : ds + ds * du + du));
}

// bytes of LDS per wave that hold the env's critic weights (0: they live in registers) - shared by the kernel (layout)
// and the launcher (dynamic-LDS request)
__host__ __device__ constexpr int dma_wslot(int esz, int variant, int ds, int du) {
  const int dcmax = (dma_is_sql(variant) || dma_is_rql(variant)) ? dma_dc(dma_cs(variant), ds, du) : 0;
  return dcmax > 9 ? ((dcmax * esz + 15) / 16) * 16 : 0;
}

// wave argmin of (cost, index): lower cost wins, ties -> lower index; every lane ends with the winner's pair
__device__ __forceinline__ void wave_argmin(float& bestJ, int& bestI) {
  const unsigned long long wkey = wave_min_u64(((unsigned long long)float_order_key(bestJ) << 32) | (unsigned)bestI);
  bestJ = float_from_order_key((unsigned)(wkey >> 32));
  bestI = (int)(unsigned)wkey;
}
__device__ __forceinline__ void wave_argmin(double& bestJ, int& bestI) {
  for (int m = 1; m < 64; m <<= 1) {
    const double oJ = __shfl_xor(bestJ, m, 64);
    const int oI = __shfl_xor(bestI, m, 64);
    if ((oJ < bestJ) || (oJ == bestJ && oI < bestI)) {
      bestJ = oJ;
      bestI = oI;
    }
  }
}

// Rows per lane and tile.  A tile is one round trip to HBM per wave whatever its size, so very short rows move few
// bytes per trip: rows of <= 24 bytes (the robots' Nactor <= 3 in f32) are taken four per lane (tiles of 256 rows), rows
// of <= 32 bytes two per lane.  One instance per row length: a K that is not a multiple of the tile is a ragged last tile
// like any other.  Measured (B = 65536, K = 256, several interleaved pairs against the one-row-per-lane build, every output
// bit-identical): f32 Nactor = 2 / 3 / 4: 4.75 -> 5.65 / 5.6 -> 5.8-5.95 / 5.75 -> 5.90 TB/s, f64 Nactor = 2: 5.16 -> 5.60;
// rows of 40 and 48 bytes (Nactor = 5, 6) gained nothing and stay at one row per lane.
__host__ __device__ constexpr int dma_rpl(int r, int esz) { return r * esz <= 24 ? 4 : (r * esz <= 32 ? 2 : 1); }

template <typename Sys, typename real, int R, bool TGT, int V>
__global__ __launch_bounds__(256) void k_actor_dma(const ActorArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU, NP = Sys::NP;
  constexpr bool G1 = V == DMA_MPC_G1, SQL = dma_is_sql(V), RQL = dma_is_rql(V), CRIT = RQL || SQL;
  constexpr bool GEND = V == DMA_MPC_GEND, GENF = V == DMA_MPC_GENF, GEN = GEND || GENF;
  constexpr bool GENR = V >= DMA_RQL_GEN_0;  // RQL whose stage cost is not the presets' diagonal quadratic one: stage_any per step
  static_assert(!(GEN || GENR) || TGT, "the generic-cost instances subtract a target (zeros for a handle without one)");
  constexpr int CS = dma_cs(V);  // compile-time critic structure
  constexpr int DCMAX = CRIT ? dma_dc(CS, DS, DU) : 1;
  constexpr int ESZ = (int)sizeof(real);
  constexpr int WSLOT = dma_wslot(ESZ, V, DS, DU);  // > 0: critic weights in LDS (instances with more than 9 of them)
  constexpr bool WLDS = WSLOT > 0;
  constexpr int WREG = WLDS ? 1 : DCMAX;             // register copies of the weights (current env, next env)
  static_assert(R % DU == 0 && R >= DU && R <= 40, "row = N*du reals, at most 40 (f32: 160 bytes, f64: 320)");
  constexpr int N = R / DU;
  constexpr int RPL = dma_rpl(R, ESZ), TROWS = 64 * RPL;           // rows per lane, rows per tile
  constexpr int TILE = TROWS * R * ESZ;                            // bytes of one tile
  constexpr int NFULL = TILE / 1024, NREM = (TILE % 1024) / 256;  // 1-KiB and 256-B direct-to-LDS loads per tile
  static_assert(NFULL * 1024 + NREM * 256 == TILE, "a tile is a whole number of 256-B segments");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int lane = threadIdx.x & 63;
  // readfirstlane: provably wave-uniform, so tile bases live in SGPRs and the control flow is scalar
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
  const long B = P.B;
  const int K = A.K;
  const int T = (K + TROWS - 1) / TROWS;  // tiles per env
  const int rem_rows = K % TROWS;         // rows of an env's LAST tile (0: it is full); K % 4 == 0: whole 16-byte pieces
  const long env_hi = A.env_hi > 0 ? A.env_hi : B;  // (a sub-range of the batch when the handle splits its tick)
  const long env0 = A.env_lo + wave * A.gpw;
  if (env0 >= env_hi) return;
  const long env1 = env0 + A.gpw < env_hi ? env0 + A.gpw : env_hi;
  // 32-bit on purpose: a 64-bit loop compare has no scalar form, and its VGPR temporary once landed in the registers
  // of the in-flight env-state prefetch (a WAW hazard the compiler resolves with s_waitcnt vmcnt(0))
  const int n_tiles = (int)(env1 - env0) * T;

  unsigned char* const tile = smem_raw + (size_t)wave_in_wg * TILE;  // this wave's LDS tile
  const size_t env_stride = (size_t)K * (R * ESZ);                    // bytes of one env's rows
  const unsigned char* envb = reinterpret_cast<const unsigned char*>(A.cand) + (size_t)env0 * env_stride;  // env b's rows
  // operator mode (rcg_actor_cost): the J of one env is staged in LDS ([K] reals per wave, behind the tiles of all
  // four waves) and written out at the env's end in 1-KiB bursts; a 256-B store after every tile, interleaved with
  // the read stream, made the operator 35 % slower than the tick for 5 % more bytes
  // (A.jwave: the staging area holds all envs of the wave and is written once, when the wave is done)
  const int jspan = A.jwave ? A.gpw * K : K;  // reals of staging per wave
  real* const jstage = reinterpret_cast<real*>(smem_raw + (size_t)4 * TILE + (size_t)4 * WSLOT) + (size_t)wave_in_wg * jspan;
  // (WLDS) this wave's critic weights: [dc] reals behind the four tiles
  real* const wl = reinterpret_cast<real*>(smem_raw + (size_t)4 * TILE + (size_t)wave_in_wg * WSLOT);

  // `rows` (wave-uniform): TROWS, or rem_rows for an env's ragged last tile - then every lane loads only pieces that lie
  // inside the env's rows (rows * R * ESZ bytes, a multiple of 16), the rest of the LDS tile keeps stale rows that no
  // valid candidate index points at
  auto issue_tile = [&](const unsigned char* g, int rows) {
    if (rows == TROWS) {
#pragma unroll
      for (int j = 0; j < NFULL; ++j)
        __builtin_amdgcn_global_load_lds((glb_void*)(g + j * 1024 + lane * 16), (lds_void*)(tile + j * 1024), 16, 0,
                                         RCG_DMA_AUX);
#pragma unroll
      for (int j = 0; j < NREM; ++j)
        __builtin_amdgcn_global_load_lds((glb_void*)(g + NFULL * 1024 + j * 256 + lane * 4),
                                         (lds_void*)(tile + NFULL * 1024 + j * 256), 4, 0, RCG_DMA_AUX);
    } else {
      const int vb = rows * (R * ESZ);  // valid bytes of the tile
#pragma unroll
      for (int j = 0; j < NFULL; ++j)
        if (j * 1024 + lane * 16 < vb)
          __builtin_amdgcn_global_load_lds((glb_void*)(g + j * 1024 + lane * 16), (lds_void*)(tile + j * 1024), 16, 0,
                                           RCG_DMA_AUX);
#pragma unroll
      for (int j = 0; j < NREM; ++j)
        if (NFULL * 1024 + j * 256 + lane * 4 < vb)
          __builtin_amdgcn_global_load_lds((glb_void*)(g + NFULL * 1024 + j * 256 + lane * 4),
                                           (lds_void*)(tile + NFULL * 1024 + j * 256), 4, 0, RCG_DMA_AUX);
    }
  };
  auto rows_of = [&](int tt) -> int { return (tt == T - 1 && rem_rows) ? rem_rows : TROWS; };

  // env state: `n`-suffixed = requested one tile ahead for the next env.  Loads only, no
  // "pointer ? load : default" selects (a default written into a register with a load in flight would force a
  // vmcnt(0) on the spot).
  real y0[DS], yn[DS], x0[DS], xn[DS], pv[NP > 0 ? NP : 1], pn[NP > 0 ? NP : 1], wc[WREG], wn[WREG];
  real wnl = 0;  // (WLDS) weight `lane` of the next env
  const bool lag = A.state_sys != A.obs;  // wave-uniform: a second state vector per env (20 B more per 20 KB of rows)
#pragma unroll
  for (int i = 0; i < NP; ++i) pn[i] = P.pars[i];
#pragma unroll
  for (int i = 0; i < WREG; ++i) wn[i] = wc[i] = 0;  // entries >= dc are never loaded and never read
  constexpr int dc_rt = CRIT ? DCMAX : 0;  // weights the env has
  auto fetch_env = [&](long b) {
    if (RCG_DBG(A, 4)) {  // development: no env-state loads
#pragma unroll
      for (int c = 0; c < DS; ++c) yn[c] = xn[c] = (real)0.5;
      return;
    }
#pragma unroll
    for (int c = 0; c < DS; ++c) yn[c] = A.obs[(long)c * B + b];
    if (lag) {
#pragma unroll
      for (int c = 0; c < DS; ++c) xn[c] = A.state_sys[(long)c * B + b];
    }
    if (A.pars_env) {
#pragma unroll
      for (int i = 0; i < NP; ++i) pn[i] = A.pars_env[(long)i * B + b];
    }
    if (CRIT && WLDS) {
      if (lane < dc_rt) wnl = A.w[(long)lane * B + b];  // one weight per lane; parked in LDS when the env is adopted
    } else if (CRIT) {
#pragma unroll
      for (int i = 0; i < WREG; ++i) wn[i] = A.w[(long)i * B + b];
    }
  };

  // (GENF, GENR) the symmetrised upper triangles of R1 / R2, once per wave, in registers for every rollout of the wave
  constexpr bool TRI = GENF || GENR;
  constexpr int NSYM = TRI ? sym_len<NCHI>() : 1;
  real S1[NSYM], S2[NSYM];
  const bool biq = (GEN || GENR) && (P.stage_kind & STAGE_BIQUAD);               // wave-uniform
  const bool full_rt = GENF || (GENR && (P.stage_kind & STAGE_FULL));           // wave-uniform (GENF: compile-time)
  if constexpr (TRI) {
    if (full_rt) {
      load_sym<NCHI, real>(P.Rfull, S1);
    } else {
#pragma unroll
      for (int i = 0; i < NSYM; ++i) S1[i] = 0;
    }
    if (full_rt && biq) {
      load_sym<NCHI, real>(P.Rfull + 49, S2);
    } else {
#pragma unroll
      for (int i = 0; i < NSYM; ++i) S2[i] = 0;
    }
  }
  // stage_with's arithmetic (rcg_kernels.hpp) on the register copies of the matrices: the same bits as stage_any
  auto gen_stage = [&](const real* chi) -> real {
    real q;
    if (TRI && full_rt)
      q = quad_sym<NCHI, real>(S1, chi);
    else
      q = stage_diag<NCHI, real>(P, chi);
    if (biq) {
      real c2[NCHI];
#pragma unroll
      for (int i = 0; i < NCHI; ++i) c2[i] = chi[i] * chi[i];
      real q4;
      if (TRI && full_rt) {
        q4 = quad_sym<NCHI, real>(S2, c2);
      } else {
        q4 = 0;
#pragma unroll
        for (int i = 0; i < NCHI; ++i) q4 = fma_r(P.R2d[i] * c2[i], c2[i], q4);
      }
      q = q4 + q;
    }
    return q;
  };

  fetch_env(env0);
  issue_tile(envb, rows_of(0));  // after the env request: retiring the env state must not drain the first tile

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
    if (t == 0) {  // first tile of env b: adopt the state requested one tile ago
#pragma unroll
      for (int c = 0; c < DS; ++c) {
        y0[c] = yn[c];
        x0[c] = lag ? xn[c] : yn[c];
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) pv[i] = pn[i];
      pre_env = Sys::template prepare<real>(pv);
      if (CRIT && WLDS) {  // the previous env's rollouts are done (program order): its weights may be overwritten
        if (lane < dc_rt) wl[lane] = wnl;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      } else if (CRIT) {
#pragma unroll
        for (int i = 0; i < WREG; ++i) wc[i] = wn[i];
      }
      bestJ = inf_r<real>();
      bestI = 0x7fffffff;
    }
    // 2. tile g has landed -> my row into registers (vmcnt retires in issue order: everything requested so far)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    real rows[RPL][R];  // the lane's rows: lane, lane + 64, ... of the tile
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      const real* const myrow = reinterpret_cast<const real*>(tile) + (j * 64 + lane) * R;
#pragma unroll
      for (int i = 0; i < R; ++i) rows[j][i] = myrow[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane's row is out of LDS (in-order per wave)
    __builtin_amdgcn_wave_barrier();
    // 3. the LDS tile is free: request tile g + 1 into it, an env-state request first if it opens an env
    if (g + 1 < n_tiles) {
      if (t == T - 1) {
        fetch_env(b + 1);
        envb += env_stride;
        issue_tile(envb, rows_of(0));
      } else {
        issue_tile(envb + (size_t)(t + 1) * TILE, rows_of(t + 1));
      }
    }

    // 4. _actor_cost of this lane's rows (controllers.py:1284-1326), registers only, in candidate order
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      const real* const cur = rows[j];
      real x[DS], y[DS];
#pragma unroll
      for (int c = 0; c < DS; ++c) {
        x[c] = x0[c];  // state_sys
        y[c] = y0[c];  // observation_sqn[0] = observation
      }
      real J = 0, gk = 1;
      real S[NCHI];
#pragma unroll
      for (int i = 0; i < NCHI; ++i) S[i] = 0;
      real Phi[SQL ? DCMAX : 1];  // SQL: the regressor summed over the horizon (J = sum_k w . phi_k = w . sum_k phi_k)
#pragma unroll
      for (int i = 0; i < (SQL ? DCMAX : 1); ++i) Phi[i] = 0;
      auto wget = [&](int i) -> real { return WLDS ? wl[i] : wc[WLDS ? 0 : i]; };
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
          } else if (SQL) {
            critic_phi_accum<DS, DU, real>(chi, y, &cur[kk * DU], Phi, CS);
          } else if (RQL && kk == N - 1) {
            J += critic_with<DS, DU, real>(chi, y, &cur[kk * DU], wget, CS);
          } else if (GEN || GENR) {
            J = fma_r(gk, gen_stage(chi), J);
            gk *= P.gamma;
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
          for (int i = 0; i < DCMAX; ++i) J = fma_r(wget(i), Phi[i], J);
        }
      }

      const int k = t * TROWS + j * 64 + lane;
      const bool has_row = k < K;  // false only in a ragged last tile
      if (A.J && has_row) jstage[(A.jwave ? (int)(b - env0) * K : 0) + k] = J;
      const real Jc = (J != J) ? inf_r<real>() : J;  // NaN counts as +inf
      if (has_row && (Jc < bestJ || bestI == 0x7fffffff)) {
        bestJ = Jc;
        bestI = k;
#pragma unroll
        for (int c = 0; c < DU; ++c) bu[c] = cur[c];  // the sequence's first action
      }
    }

    // env b complete (per-env staging) or wave complete (A.jwave): the staged costs go out in one piece
    if (A.J && (A.jwave ? g == n_tiles - 1 : t == T - 1)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      real* const Jout = A.J + (A.jwave ? env0 : b) * K;
      const int n = A.jwave ? (int)(env1 - env0) * K : K;  // a multiple of 64
      const int nbytes = n * ESZ;
      if ((nbytes & 1023) == 0) {
        const unsigned char* const src = reinterpret_cast<const unsigned char*>(jstage);
        unsigned char* const dst = reinterpret_cast<unsigned char*>(Jout);
        for (int i = lane * 16; i < nbytes; i += 1024)  // written once, never re-read by this launch: non-temporal
          __builtin_nontemporal_store(*reinterpret_cast<const v4f*>(src + i), reinterpret_cast<v4f*>(dst + i));
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
      // the winner's first action is read from the winner's lane: every lane kept the action of its own best row,
      // and bestI = tile * 64 + lane
      wave_argmin(bestJ, bestI);
      const int wl = __builtin_amdgcn_readfirstlane(bestI & 63);
#pragma unroll
      for (int c = 0; c < DU; ++c) bu[c] = readlane_r(bu[c], wl);
      real acc_inc = 0;
      if (A.accum) {  // upd_accum_obj (controllers.py:1086-1093), wave-uniform
        real chi[NCHI];
#pragma unroll
        for (int c = 0; c < DS; ++c) chi[c] = TGT ? y0[c] - P.target[c] : y0[c];
#pragma unroll
        for (int c = 0; c < DU; ++c) chi[DS + c] = bu[c];
        // (SQL has no stage cost inside the rollout, so its instances serve ANY stage structure: only upd_accum_obj sees it)
        acc_inc = ((GEN || GENR || SQL) ? stage_any<NCHI, real>(P, chi) : stage_diag<NCHI, real>(P, chi)) * P.sampling_time;
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

// ---- launchers: the instances live in their own translation units (rcg_dma_inst.hip, one object per system x
// element type x group, so that the library builds in parallel); rcg_sysops.hpp only sees this declaration ----------
// group 0: DMA_MPC_G1, DMA_MPC; group 1: DMA_SQL_0 .. + 3; group 2: DMA_RQL_0 .. + 3; (3-5: k_actor_dma_packed;) group 6: DMA_MPC_GEND,
// DMA_MPC_GENF; group 7: DMA_RQL_GEN_0 .. + 3.
// Returns false when there is no instance for (row length r, variant).  ev_a / ev_b (both or neither): the launch carries
// them as its start / stop events (rcg_profile, rcg_handle.hpp::ProfScope).
template <typename Sys, typename real, int GROUP>
bool launch_dma(int r, int variant, dim3 grid, dim3 block, size_t lds, hipStream_t s, const ActorArgs<real>& A,
                const KParams<real>& P, hipEvent_t ev_a, hipEvent_t ev_b);

// longest row with an instance, in reals: 40 = the robots' Nactor = 20 (f32: 160 bytes; f64: 320 bytes, a block's four
// tiles are then 80 KB of LDS - beyond the default dynamic limit, launch_dma raises it for those instances)
template <typename real>
constexpr int dma_max_row() {
  return 40;
}

}  // namespace rcg
