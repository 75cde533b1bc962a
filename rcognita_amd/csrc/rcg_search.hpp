// rcg_search.hpp - device-side candidate producer fused into the evaluation: k_actor_search (rcg_actor_search,
// rcg_control_tick_search) and the stand-alone producer k_cand_sample (rcg_candidates_sample).
//
// What it replaces: the candidate search that stands in for SLSQP in CtrlOptPred._actor_optimizer
// (rcognita/controllers.py:1330-1427) used to draw its candidates on the HOST with numpy and upload them for every
// refinement round (20 KB per env and round over PCIe); a closed loop whose candidates change every tick had no producer
// on the device at all.  Here the candidates never exist in HBM: a wave owns one env, every lane GENERATES its candidate
// row from a counter-based generator into the wave's LDS tile, rolls it out with the code k_actor runs (rollout_dispatch:
// every mode and cost structure), the wave takes the argmin, REGENERATES the winner's row as the centre of the next
// round, and after `rounds` rounds writes one sequence, one action and one cost per env.
//
// The sampling rule (build-defined; oracle twin oracle/rcg_oracle.py::candidates_sample, same statements):
//   centre   round 0: the caller's sequence (rcg_control_tick_search: action_sqn_init, or the previous optimum shifted by one
//            step); later rounds: the previous round's winner
//   k = 0    the centre itself (the search is monotone);  round 0, k = 1: action_sqn_init (the reference's start point)
//   else     clip(centre + sigma_r * xi, lo, hi),  sigma_r = 0.5 (hi - lo) 2^-round;  xi ~ N(0, 1):
//            k < K - K / 4 one draw per input held over the horizon, the last quarter one draw per input and step
//   xi       the env's key for the tick: words 0, 1 of Philox4x32-7(counter = (env id lo, env id hi, EPISODE_IDX, STEP_IDX),
//            key = (seed lo ^ 'CAND', seed hi)) - an env's stream does not depend on batch size, sharding or launch geometry.
//            A DRAW (round 5) = Philox4x32-7(counter = (a, b, round, kind), that key) -> four words -> EIGHT normals: word p
//            gives the Box-Muller pair (n_2p, n_2p+1) = sqrt(-2 ln u_r) (cos, sin)(2 pi u_a) with the 16-bit uniforms
//            u_r = ((w & 0xffff) + 0.5) 2^-16, u_a = ((w >> 16) + 0.5) 2^-16 (|xi| <= 4.8).
//              per-step candidate k (k >= K - K / 4): row element e comes from draw (a, b, kind) = (k, e / 8, 0), normal e % 8;
//              held candidate k (k < K - K / 4): with l = k mod 64, t = k / 64 and TPC = 8 / du, input c comes from draw
//              (l, t / TPC, 1), normal (t mod TPC) du + c - ONE draw serves the lane's held candidates of TPC tiles.
//            Round 4 drew four 24-bit uniforms from a 10-round call per four row elements: 12 calls per lane and round at
//            C2 (K = 256, R = 20) against 7 seven-round ones now.
//            The integer stream is bit-exact against the oracle; logarithm, square root, sine and cosine are the hardware's
//            float32 forms (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 on the revolution u_a itself), whatever the handle's
//            element type: a candidate agrees with the oracle's float64 evaluation to 1e-5 sigma (tests/test_hip_search.py).
// Rows: the instances with a compile-time horizon (NC = 3, 5, 10: MPC with a diagonal stage cost and the preset's target
// setting) build the lane's row in REGISTERS and roll it out fully unrolled (rollout_cost<..., NC>), skipping the preset's
// zero-weighted cost terms; every other shape writes the row to the wave's LDS tile and walks it with a runtime horizon.
#pragma once
#include "rcg_actor_dma.hpp"  // wave_argmin
#include "rcg_disturb.hpp"
#include "rcg_kernels.hpp"

namespace rcg {

constexpr int CAND_ROUNDS = 7;  // Philox rounds of a candidate draw

struct CandKey {
  uint32_t k0, k1;
};

__device__ __forceinline__ CandKey cand_subkey(uint64_t seed, int64_t env_id, int32_t episode, int32_t step) {
  const uint64_t e = (uint64_t)env_id;
  const PhiloxOut o = philox4x32<CAND_ROUNDS>((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)episode, (uint32_t)step,
                                              (uint32_t)seed ^ 0x43414E44u, (uint32_t)(seed >> 32));
  return CandKey{o.w[0], o.w[1]};
}

// the eight standard normals of draw (a, b, round, kind) under an env's key
__device__ __forceinline__ void cand_normals8(const CandKey& key, uint32_t a, uint32_t b, uint32_t round, uint32_t kind,
                                              float* n) {
  const PhiloxOut o = philox4x32<CAND_ROUNDS>(a, b, round, kind, key.k0, key.k1);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    // (m + 0.5) 2^-16: the product is exact, one rounding in the sum - and none at all for a 16-bit m (numpy float32 agrees)
    const float ur = __builtin_fmaf((float)(o.w[p] & 0xffffu), 1.52587890625e-05f, 7.62939453125e-06f);
    const float ua = __builtin_fmaf((float)(o.w[p] >> 16), 1.52587890625e-05f, 7.62939453125e-06f);
    const float r = __builtin_amdgcn_sqrtf(-2.0f * (__builtin_amdgcn_logf(ur) * 0.693147180559945309f));
    n[2 * p] = r * __builtin_amdgcn_cosf(ua);  // argument in revolutions
    n[2 * p + 1] = r * __builtin_amdgcn_sinf(ua);
  }
}

template <int DU>
struct CandGeom {
  static_assert(8 % DU == 0, "a draw of eight normals starts on a step boundary");
  static constexpr int TPC = 8 / DU;  // tiles of 64 held candidates one draw serves
};

// normal q (0 .. 7, per lane) of a draw held in registers: a select tree on the bits of q (written out - a loop over the
// array, even a fully unrolled one, left the array in scratch memory: k_cand_sample ran at 0.97 ms instead of 0.3)
__device__ __forceinline__ float pick8(const float* n, int q) {
  const float a0 = (q & 1) ? n[1] : n[0], a1 = (q & 1) ? n[3] : n[2], a2 = (q & 1) ? n[5] : n[4], a3 = (q & 1) ? n[7] : n[6];
  const float b0 = (q & 2) ? a1 : a0, b1 = (q & 2) ? a3 : a2;
  return (q & 4) ? b1 : b0;
}


// first per-step candidate: the last quarter of the K candidates draws one normal per input AND step, the others one per input
// (round 4: the last half - on the reference's own decisions, fixtures F8, the search ends within 0.40 % of SLSQP's cost either
// way (0.34 % with half), tests/test_search_oracle.py, and a per-step row costs three draws where a held one costs a third)
__host__ __device__ constexpr int cand_ps_first(int K) { return K - (K >> 2); }

// The held normals of candidate k = t * 64 + l from the lane's held draw H of tile group t / TPC.
template <int DU>
__device__ __forceinline__ void held_normals(const float* H, int t, float* xih) {
  const int q = (t % CandGeom<DU>::TPC) * DU;
#pragma unroll
  for (int c = 0; c < DU; ++c) xih[c] = pick8(H, q + c);
}

// Row of candidate k in round `round` -> put(i, value) for i = 0 .. R - 1 (see the rule above).  RC > 0: the row length is the
// compile-time constant RC and every index below is static.  `centre(i)`: the centre's element i; `xih`: the lane's held
// normals for this tile.  Candidates 0 (the centre itself) and, in round 0, 1 (action_sqn_init) are put in place by the caller
// (cand_row_fix: they live in the first tile only).
template <int DU, typename real, int RC, typename Centre, typename Put>
__device__ __forceinline__ void cand_row(const KParams<real>& P, const CandKey& key, int k, int K, int R_rt, int round,
                                         Centre centre, const real* sigma, const float* xih, Put put) {
  const int R = RC > 0 ? RC : R_rt;
  const int half = cand_ps_first(K), base = k & ~63;
  const bool ps = k >= half;
  const bool all_held = base + 63 < half, all_ps = base >= half;  // wave-uniform
  const int n_draws = (R + 7) / 8;
  auto draw = [&](const int j) {
    float xi[8];
    if (all_held) {
#pragma unroll
      for (int e = 0; e < 8; ++e) xi[e] = xih[e % DU];
    } else {
      float n8[8];
      cand_normals8(key, (uint32_t)k, (uint32_t)j, (uint32_t)round, 0u, n8);
      if (all_ps) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xi[e] = n8[e];
      } else {  // the one tile K / 2 falls into when it is not a multiple of 64
#pragma unroll
        for (int e = 0; e < 8; ++e) xi[e] = ps ? n8[e] : xih[e % DU];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = 8 * j + e, c = e % DU;
      if (i < R) {
        put(i, clamp_fin(fma_r(sigma[c], (real)xi[e], centre(i)), P.lo[c], P.hi[c]));
      }
    }
  };
  if constexpr (RC > 0) {
#pragma unroll
    for (int j = 0; j < (RC + 7) / 8; ++j) draw(j);
  } else {
    for (int j = 0; j < n_draws; ++j) draw(j);
  }
}

// Element i of candidate row k (the winner's row, regenerated lane by lane as the next round's centre; the producer
// k_cand_sample's rows), from the same counters.
template <int DU, typename real>
__device__ __forceinline__ real cand_element(const KParams<real>& P, const CandKey& key, int k, int K, int i, int round,
                                             real cv, const real* sigma, const real* u0) {
  const int c = i % DU;
  if (k == 0) return cv;
  if (k == 1 && round == 0) {
    real uu = u0[0];
#pragma unroll
    for (int q = 1; q < DU; ++q) uu = (c == q) ? u0[q] : uu;
    return uu;
  }
  float n8[8];
  float xi;
  if (k >= cand_ps_first(K)) {
    cand_normals8(key, (uint32_t)k, (uint32_t)(i >> 3), (uint32_t)round, 0u, n8);
    xi = pick8(n8, i & 7);
  } else {
    const int t = k >> 6;
    cand_normals8(key, (uint32_t)(k & 63), (uint32_t)(t / CandGeom<DU>::TPC), (uint32_t)round, 1u, n8);
    xi = pick8(n8, (t % CandGeom<DU>::TPC) * DU + c);
  }
  real sg = sigma[0], lo = P.lo[0], hi = P.hi[0];
#pragma unroll
  for (int q = 1; q < DU; ++q) {
    sg = (c == q) ? sigma[q] : sg;
    lo = (c == q) ? P.lo[q] : lo;
    hi = (c == q) ? P.hi[q] : hi;
  }
  return clamp_fin(fma_r(sg, (real)xi, cv), lo, hi);
}

template <typename real>
struct SearchArgs {
  const real* obs;        // [dy][B]
  const real* state_sys;  // [ds][B]
  const real* pars_env;   // [np][B] or nullptr
  const real* w;          // [dc][B] (RQL / SQL)
  const real* centre_in;  // [B][N][du] or nullptr (-> u0 tiled over the horizon)
  real* u_best;           // [B][N][du] or nullptr
  real* action_out;       // [du][B] or nullptr
  real* best_J;           // [B] or nullptr
  int32_t* best_idx;      // [B] or nullptr: the winner's index in the LAST round (0: the incumbent was kept)
  real* accum;            // tick epilogue (or nullptr)
  int32_t* step_rw;       // tick epilogue: STEP_IDX += 1 (or nullptr)
  const int32_t* episode_idx;  // [B] draw counters
  const int32_t* step_idx;     // [B]
  real u0[RCG_MAX_DU];    // action_sqn_init entry (controllers.py:973-978)
  int K, rounds, round0;  // candidates per round; rounds to run; number of the first one (sets sigma and the draw)
  int shift;              // centre_in is last tick's optimum: shift it by one step (last entry repeated)
  uint64_t seed;
  int64_t env_id_base;
};

// per-wave LDS: tile [64][R] | centre [R].  Runtime-horizon instances build and walk the lane's row in the tile; the
// register-row instances park there the best row each lane has met in the round, so that the winner's row is one LDS copy
// (regenerating it from its counters cost a draw, eight normals and a select tree per round and wave)
__host__ __device__ constexpr int search_lds_reals(int R, bool /*reg_rows*/) { return (65 * R + 3) & ~3; }  // (a whole number of 16-byte pieces per wave)

template <typename Sys, typename real, bool GENERIC, bool TGT, int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NC > 0 && sizeof(real) == 4 ? 4 : 1))) void k_actor_search(const SearchArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  constexpr int RC = NC * DU;  // 0: runtime horizon, rows in LDS
  constexpr int TPC = CandGeom<DU>::TPC;
  static_assert(NC == 0 || !GENERIC, "register rows: MPC with a diagonal stage cost");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long b = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;  // wave == env
  const long B = P.B;
  if (b >= B) return;  // wave-uniform; no workgroup barrier below
  const int K = A.K, N = NC > 0 ? NC : P.n_actor, R = N * DU;
  real* const wave_lds = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * search_lds_reals(R, NC > 0);
  real* const tile = wave_lds;
  // (16-byte aligned: the wave's region is a whole number of 16-byte pieces and 64 rows are 256 R bytes - the centre's reads
  // become ds_read_b128)
  real* const centre = static_cast<real*>(__builtin_assume_aligned(wave_lds + 64 * R, 16));
  real* const myrow = tile + (size_t)lane * R;

  real y0[DS], xs[DS];
#pragma unroll
  for (int c = 0; c < DS; ++c) {
    y0[c] = A.obs[(long)c * B + b];
    xs[c] = A.state_sys[(long)c * B + b];
  }
  const auto pre = load_pre<Sys, real>(P, A.pars_env, b);
  constexpr int DCMAX = GENERIC ? NCHI * (NCHI + 1) / 2 + NCHI : 1;
  real wreg[DCMAX];
  if (GENERIC) {
    const bool has_w = P.mode != RCG_MODE_MPC && A.w != nullptr;
#pragma unroll
    for (int i = 0; i < DCMAX; ++i) wreg[i] = (has_w && i < P.dc) ? A.w[(long)i * B + b] : (real)0;
  }
  auto wget = [&](int i) -> real { return wreg[GENERIC ? i : 0]; };
  const CandKey key = cand_subkey(A.seed, A.env_id_base + b, A.episode_idx[b], A.step_idx[b]);

  for (int i = lane; i < R; i += 64) {
    real v;
    if (A.centre_in) {
      int j = i;
      if (A.shift) j = (i + DU < R) ? i + DU : i;  // u_k <- u_{k+1}, the last step repeated
      v = A.centre_in[b * R + j];
    } else {
      v = A.u0[0];
#pragma unroll
      for (int q = 1; q < DU; ++q) v = (i % DU == q) ? A.u0[q] : v;  // (a dynamic index would copy the argument block to scratch)
    }
    centre[i] = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  const int n_tiles = (K + 63) / 64, half = cand_ps_first(K);
  real bestJ = inf_r<real>();
  int bestI = 0;
  for (int r = 0; r < A.rounds; ++r) {
    const int round = A.round0 + r;
    real sigma[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) sigma[c] = ((real)0.5 * (P.hi[c] - P.lo[c])) * (real)__builtin_ldexp(1.0, -round);  // 2^-round, exact
    bestJ = inf_r<real>();
    bestI = 0x7fffffff;
    for (int tg = 0; tg < n_tiles; tg += TPC) {  // TPC tiles share the lane's held draw
      float H[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (tg * 64 < half)  // wave-uniform: some candidate of this tile group is a held one
        cand_normals8(key, (uint32_t)lane, (uint32_t)(tg / TPC), (uint32_t)round, 1u, H);
#pragma unroll
      for (int tt = 0; tt < TPC; ++tt) {
        const int t = tg + tt;
        if (t >= n_tiles) break;  // wave-uniform
        const int k = t * 64 + lane;
        const bool valid = k < K;
        float xih[DU];
#pragma unroll
        for (int c = 0; c < DU; ++c) xih[c] = H[tt * DU + c];
        const bool fix0 = t == 0, fix1 = t == 0 && round == 0;  // wave-uniform: the tile holds candidate 0 / candidate 1 = u0
        real u0[DU], J;
        real row_keep[RC > 0 ? RC : 1];
        if constexpr (RC > 0) {
          real* const row = row_keep;
          cand_row<DU, real, RC>(P, key, k, K, R, round, [&](int i) { return centre[i]; }, sigma, xih,
                                 [&](int i, real v) { row[i] = v; });
          if (fix0) {
#pragma unroll
            for (int i = 0; i < RC; ++i) row[i] = (k == 0) ? centre[i] : ((fix1 && k == 1) ? A.u0[i % DU] : row[i]);
          }
          J = rollout_dispatch<Sys, real, GENERIC, TGT, true, NC>(P, pre, N, xs, y0, row, nullptr, wget, u0);
        } else {
          // the lane writes ITS row and then reads only that row back: no cross-lane hazard on the tile; `centre` is
          // written between rounds only
          cand_row<DU, real, 0>(P, key, k, K, R, round, [&](int i) { return centre[i]; }, sigma, xih,
                                [&](int i, real v) { myrow[i] = v; });
          if (fix0 && k < 2) {
            for (int i = 0; i < R; i += DU) {
#pragma unroll
              for (int c = 0; c < DU; ++c)
                if (k == 0 || fix1) myrow[i + c] = (k == 0) ? centre[i + c] : A.u0[c];
            }
          }
          J = rollout_dispatch<Sys, real, GENERIC, TGT, true>(P, pre, N, xs, y0, myrow, nullptr, wget, u0);
        }
        const real Jc = (J != J) ? inf_r<real>() : J;  // NaN counts as +inf
        if (valid && (Jc < bestJ || bestI == 0x7fffffff)) {
          bestJ = Jc;
          bestI = k;
          if constexpr (RC > 0) {  // the lane's best row of the round so far (1 + 1/2 + 1/3 + ... writes per round)
#pragma unroll
            for (int i = 0; i < RC; ++i) myrow[i] = row_keep[i];
          }
        }
      }
    }
    wave_argmin(bestJ, bestI);  // lower J, then lower index; every lane ends with the winner (f32: DPP + readlane on a packed key)
    // the winner's row becomes the centre
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (bestI != 0) {  // wave-uniform; candidate 0 IS the centre
      if constexpr (RC > 0) {  // it sits in the winning lane's slot of the tile
        const real* const wrow = tile + (size_t)(bestI & 63) * R;
        for (int i = lane; i < R; i += 64) centre[i] = wrow[i];
      } else {  // lane i regenerates element i (from the same counters)
        for (int i0 = 0; i0 < R; i0 += 64) {
          const int i = i0 + lane;
          if (i < R) centre[i] = cand_element<DU, real>(P, key, bestI, K, i, round, centre[i], sigma, A.u0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }

  for (int i = lane; i < R; i += 64)
    if (A.u_best) A.u_best[b * R + i] = centre[i];
  if (lane == 0) {
    real a[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) {
      a[c] = centre[c];
      if (A.action_out) A.action_out[(long)c * B + b] = a[c];
    }
    if (A.best_J) A.best_J[b] = bestJ;
    if (A.best_idx) A.best_idx[b] = bestI;
    if (A.accum) A.accum[b] = accum_update<Sys, TGT, real>(P, y0, a, A.accum[b]);
    if (A.step_rw) A.step_rw[b] += 1;
  }
}

// The producer alone (rcg_candidates_sample): the rows k_actor_search evaluates in `round` around `centre` [B][N][du] (nullptr:
// u0 tiled) -> cand [B][K][N][du], for inspection, tests, and callers that want to stream them through rcg_actor_cost /
// rcg_actor_argmin.  Grid (B, ceil(K / KB)): a block works on KB consecutive candidates (a multiple of 64) of ONE env, in two
// phases.  (1) Its threads compute every draw those rows need exactly once - one per lane and group of TPC held tiles, ceil(R / 8)
// per per-step row - and park the eight normals of each in LDS.  (2) Its threads walk the block's slab of the tensor in 16-byte
// pieces, consecutive threads consecutive pieces (a wave's store instruction is one contiguous 1-KiB segment, non-temporal),
// fetching each piece's normals from LDS.  History (C2: 1.34 GB): one thread per row, 80-byte strides between lanes: 1.3 ms;
// one thread per 4-element chunk on a flat 64-bit index: 0.72 ms; a block inside one env, one ten-round call per 4 elements:
// 0.46 ms (round 4); one thread per draw, each writing its 32 bytes: 1.0 ms with non-temporal stores (half lines), 0.39 ms
// with plain ones - 3 threads of a held row repeated the same draw; this form: DESIGN.md 5.
__host__ __device__ constexpr int cand_block_draws(int KB, int n_draws, int du) {  // LDS slots (of eight floats) a block needs
  return 64 * ((KB / 64 + 8 / du - 1) / (8 / du) + 1) + KB * n_draws;
}

template <int DU, typename real>
__global__ __launch_bounds__(256) void k_cand_sample(real* cand, const real* centre_in, const int32_t* episode_idx,
                                                     const int32_t* step_idx, int K, int round, int R, uint64_t seed,
                                                     int64_t env_id_base, real u00, real u01, int KB, const KParams<real> P) {
  constexpr int TPC = CandGeom<DU>::TPC;
  constexpr int PER = 16 / (int)sizeof(real);  // elements per 16-byte piece
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const slots = reinterpret_cast<float*>(smem_raw);  // [n_h + n_p][8]
  const int n_draws = (R + 7) / 8, ps0 = cand_ps_first(K);
  const long b = blockIdx.x;
  const int k0 = (int)blockIdx.y * KB, k1 = min(K, k0 + KB);  // this block's candidates (k0 a multiple of 64)
  const int tid = (int)threadIdx.x;
  // (every thread derives the env's key itself: one ten-round call per wave instead of a broadcast through LDS behind a
  // barrier - the block is latency-bound, not issue-bound)
  const CandKey key = cand_subkey(seed, env_id_base + b, episode_idx[b], step_idx[b]);
  // (1) the draws: held groups g0 .. g1 (64 lanes each) of the tiles that hold a held candidate, then the per-step rows
  const int kh1 = min(k1, ps0);                       // held candidates of the block: [k0, kh1)
  const int g0 = (k0 >> 6) / TPC, g1 = kh1 > k0 ? ((kh1 - 1) >> 6) / TPC : g0 - 1;
  const int n_h = 64 * (g1 - g0 + 1);
  const int kp0 = max(k0, ps0);                       // per-step candidates of the block: [kp0, k1)
  const int n_p = k1 > kp0 ? (k1 - kp0) * n_draws : 0;
  for (int d = tid; d < n_h + n_p; d += (int)blockDim.x) {
    float n8[8];
    if (d < n_h) {
      cand_normals8(key, (uint32_t)(d & 63), (uint32_t)(g0 + (d >> 6)), (uint32_t)round, 1u, n8);
    } else {
      const int q = d - n_h, kr = q / n_draws, j = q - kr * n_draws;
      cand_normals8(key, (uint32_t)(kp0 + kr), (uint32_t)j, (uint32_t)round, 0u, n8);
    }
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 lo4 = {n8[0], n8[1], n8[2], n8[3]}, hi4 = {n8[4], n8[5], n8[6], n8[7]};
    *reinterpret_cast<v4*>(slots + 8 * d) = lo4;
    *reinterpret_cast<v4*>(slots + 8 * d + 4) = hi4;
  }
  __syncthreads();
  // (2) the slab [k0, k1) x R, in pieces of PER elements (one element at a time when a row is not a whole number of pieces)
  real sigma[DU], u0[RCG_MAX_DU] = {u00, u01};
#pragma unroll
  for (int c = 0; c < DU; ++c) sigma[c] = ((real)0.5 * (P.hi[c] - P.lo[c])) * (real)__builtin_ldexp(1.0, -round);  // 2^-round, exact
  real* const slab = cand + (b * K + k0) * (long)R;
  const real* const cen = centre_in ? centre_in + b * R : nullptr;
  const int n_el = (k1 - k0) * R;
  const float inv_r = 1.0f / (float)R;
  auto element = [&](int k, int i, int c) -> real {  // element i (input c) of candidate k
    const real cv = cen ? cen[i] : u0[c];
    float xi;
    if (k >= ps0) {
      xi = slots[8 * (n_h + (k - kp0) * n_draws + (i >> 3)) + (i & 7)];
    } else {
      const int t = k >> 6;
      xi = slots[8 * (64 * (t / TPC - g0) + (k & 63)) + (t % TPC) * DU + c];
    }
    const real v = clamp_fin(fma_r(sigma[c], (real)xi, cv), P.lo[c], P.hi[c]);
    return (k == 0) ? cv : ((k == 1 && round == 0) ? u0[c] : v);
  };
  auto split = [&](int e0, int& k, int& i) {  // e0 = (k - k0) R + i
    int kr = (int)(((float)e0 + 0.5f) * inv_r);
    int ii = e0 - kr * R;
    if (ii < 0) {
      kr -= 1;
      ii += R;
    } else if (ii >= R) {
      kr += 1;
      ii -= R;
    }
    k = k0 + kr;
    i = ii;
  };
  const bool vec = (R % PER) == 0 && (reinterpret_cast<uintptr_t>(cand) & 15) == 0 && (PER % DU) == 0;
  if (vec) {
    typedef real vecq __attribute__((ext_vector_type(PER)));
    for (int p = tid; p * PER < n_el; p += (int)blockDim.x) {
      int k, i;
      split(p * PER, k, i);
      // the piece's normals in one LDS read (i is a multiple of PER, hence of DU, and a piece never straddles a draw)
      float xi[PER];
      if (k >= ps0) {
        typedef float vecf __attribute__((ext_vector_type(PER)));
        const vecf n = *reinterpret_cast<const vecf*>(slots + 8 * (n_h + (k - kp0) * n_draws + (i >> 3)) + (i & 7));
#pragma unroll
        for (int e = 0; e < PER; ++e) xi[e] = n[e];
      } else {
        const int t = k >> 6;
        const float* const hs = slots + 8 * (64 * (t / TPC - g0) + (k & 63)) + (t % TPC) * DU;
        float xh[DU];
#pragma unroll
        for (int c = 0; c < DU; ++c) xh[c] = hs[c];
#pragma unroll
        for (int e = 0; e < PER; ++e) xi[e] = xh[e % DU];
      }
      vecq x;
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const int c = e % DU;
        const real cv = cen ? cen[i + e] : u0[c];
        x[e] = clamp_fin(fma_r(sigma[c], (real)xi[e], cv), P.lo[c], P.hi[c]);
        if (k < 2) x[e] = (k == 0) ? cv : ((round == 0) ? u0[c] : x[e]);
      }
      __builtin_nontemporal_store(x, reinterpret_cast<vecq*>(slab + (long)p * PER));
    }
  } else {
    for (int e0 = tid; e0 < n_el; e0 += (int)blockDim.x) {
      int k, i;
      split(e0, k, i);
      slab[e0] = element(k, i, i % DU);
    }
  }
}

}  // namespace rcg
