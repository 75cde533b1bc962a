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
//            k < K / 2 one draw per input held over the horizon, k >= K / 2 one draw per input and step
//   xi       Philox4x32-10.  Key of an env and tick: words 0, 1 of Philox(counter = (env id lo, env id hi, EPISODE_IDX,
//            STEP_IDX), key = (seed lo ^ 'CAND', seed hi)) - an env's stream does not depend on batch size, sharding or
//            launch geometry.  Chunk j of candidate k in round r: Philox(counter = (k, j, r, 0), that key) -> four 24-bit
//            uniforms u = (m + 0.5) 2^-24 (float32) -> two Box-Muller pairs -> the normals of row elements 4 j .. 4 j + 3.
//            The integer stream is bit-exact against the oracle; logarithm, square root, sine and cosine are the hardware's
//            float32 forms (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 on the revolution u itself), whatever the handle's
//            element type: a candidate agrees with the oracle's float64 evaluation to 1e-5 sigma (tests/test_hip_search.py).
#pragma once
#include "rcg_disturb.hpp"
#include "rcg_kernels.hpp"

namespace rcg {

struct CandKey {
  uint32_t k0, k1;
};

__device__ __forceinline__ CandKey cand_subkey(uint64_t seed, int64_t env_id, int32_t episode, int32_t step) {
  const uint64_t e = (uint64_t)env_id;
  const PhiloxOut o = philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)episode, (uint32_t)step,
                                    (uint32_t)seed ^ 0x43414E44u, (uint32_t)(seed >> 32));
  return CandKey{o.w[0], o.w[1]};
}

// four standard normals of (candidate k, chunk j, round r) under an env's key
__device__ __forceinline__ void cand_normals4(const CandKey& key, int k, int j, int round, float* xi) {
  const PhiloxOut o = philox4x32_10((uint32_t)k, (uint32_t)j, (uint32_t)round, 0u, key.k0, key.k1);
  float u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)  // (m + 0.5) 2^-24: the product is exact, one rounding in the sum (numpy float32 does the same)
    u[i] = __builtin_fmaf((float)(o.w[i] >> 8), 5.9604644775390625e-08f, 2.98023223876953125e-08f);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float r = __builtin_amdgcn_sqrtf(-2.0f * (__builtin_amdgcn_logf(u[2 * p]) * 0.693147180559945309f));
    xi[2 * p] = r * __builtin_amdgcn_cosf(u[2 * p + 1]);  // argument in revolutions
    xi[2 * p + 1] = r * __builtin_amdgcn_sinf(u[2 * p + 1]);
  }
}

// Elements 4 j .. 4 j + 3 of candidate row k in round `round` (see the rule above).  `ce`: the centre's values at those
// four elements (only the first `n_valid` are read); `xi0`: chunk 0's normals, which a held-over-the-horizon candidate
// uses at every step.
template <int DU, typename real>
__device__ __forceinline__ void cand_chunk(const KParams<real>& P, const CandKey& key, int k, int K, int j, int round,
                                           const real* ce, int n_valid, const real* sigma, const real* u0,
                                           const float* xi0, real* v) {
  static_assert(4 % DU == 0, "a chunk of four elements starts on a step boundary");
  const bool per_step = k >= (K >> 1);
  float xi[4];
  if (per_step && k > 0) {
    if (j == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) xi[e] = xi0[e];
    } else {
      cand_normals4(key, k, j, round, xi);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) xi[e] = xi0[e % DU];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = e % DU;
    real val = 0;
    if (e < n_valid) {
      const real cv = ce[e];
      if (k == 0)
        val = cv;
      else if (k == 1 && round == 0)
        val = u0[c];
      else
        val = clamp_r<real>(fma_r(sigma[c], (real)xi[e], cv), P.lo[c], P.hi[c]);
    }
    v[e] = val;
  }
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

// per-wave LDS: tile [64][R] | centre [R]
__host__ __device__ constexpr int search_lds_reals(int R) { return 65 * R; }

template <typename Sys, typename real, bool GENERIC, bool TGT>
__global__ __launch_bounds__(256) void k_actor_search(const SearchArgs<real> A, const KParams<real> P) {
  constexpr int DS = Sys::DS, DU = Sys::DU, NCHI = DS + DU;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long b = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;  // wave == env
  const long B = P.B;
  if (b >= B) return;  // wave-uniform; no workgroup barrier below
  const int K = A.K, N = P.n_actor, R = N * DU;
  real* const tile = reinterpret_cast<real*>(smem_raw) + (size_t)wave_in_wg * search_lds_reals(R);
  real* const centre = tile + 64 * R;
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
      v = A.u0[i % DU];
    }
    centre[i] = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  const int n_tiles = (K + 63) / 64, n_chunks = (R + 3) / 4;
  real bestJ = inf_r<real>();
  int bestI = 0;
  for (int r = 0; r < A.rounds; ++r) {
    const int round = A.round0 + r;
    real sigma[DU];
#pragma unroll
    for (int c = 0; c < DU; ++c) sigma[c] = ((real)0.5 * (P.hi[c] - P.lo[c])) * (real)exp2(-(double)round);
    bestJ = inf_r<real>();
    bestI = 0x7fffffff;
    for (int t = 0; t < n_tiles; ++t) {
      const int k = t * 64 + lane;
      const bool valid = k < K;
      // the lane writes ITS row and then reads only that row back: no cross-lane hazard on the tile; `centre` is
      // written between rounds only
      float xi0[4];
      cand_normals4(key, k, 0, round, xi0);
      for (int j = 0; j < n_chunks; ++j) {
        real v[4];
        cand_chunk<DU, real>(P, key, k, K, j, round, centre + 4 * j, R - 4 * j, sigma, A.u0, xi0, v);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * j + e < R) myrow[4 * j + e] = v[e];
      }
      real u0[DU];
      const real J = rollout_dispatch<Sys, real, GENERIC, TGT, true>(P, pre, N, xs, y0, myrow, nullptr, wget, u0);
      const real Jc = (J != J) ? inf_r<real>() : J;  // NaN counts as +inf
      if (valid && (Jc < bestJ || bestI == 0x7fffffff)) {
        bestJ = Jc;
        bestI = k;
      }
    }
    for (int m = 1; m < 64; m <<= 1) {  // wave argmin: lower J, then lower index; every lane ends with the winner
      const real oJ = __shfl_xor(bestJ, m, 64);
      const int oI = __shfl_xor(bestI, m, 64);
      if ((oJ < bestJ) || (oJ == bestJ && oI < bestI)) {
        bestJ = oJ;
        bestI = oI;
      }
    }
    // the winner's row becomes the centre: lane i regenerates element i (its chunk from the same counters)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (bestI != 0) {  // wave-uniform; candidate 0 IS the centre
      real nv = 0;
      if (lane < R) {
        float xi0[4];
        cand_normals4(key, bestI, 0, round, xi0);
        real v[4];
        cand_chunk<DU, real>(P, key, bestI, K, lane >> 2, round, centre + 4 * (lane >> 2), R - 4 * (lane >> 2), sigma, A.u0, xi0, v);
        const int e = lane & 3;
        nv = e == 0 ? v[0] : (e == 1 ? v[1] : (e == 2 ? v[2] : v[3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();  // every lane has read the old centre
      if (lane < R) centre[lane] = nv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }

  if (lane < R && A.u_best) A.u_best[b * R + lane] = centre[lane];
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
// rcg_actor_argmin.  Grid (B, ceil(K * chunks / 256)): a block works inside ONE env - its key (one Philox call) is drawn once
// per block - and a thread produces one 4-element chunk of a row (16 bytes in f32), so a wave's store is one contiguous 1-KiB
// piece of the tensor.  History (C2: 1.34 GB): one thread per row, 80-byte strides between lanes: 1.3 ms; one thread per
// chunk on a flat 64-bit index (two 64-bit integer divisions and a double-precision exp2 per thread): 0.72 ms; this form:
// see DESIGN.md 5 - what remains is the generator (Philox's 32-bit multiplies, the Box-Muller transcendentals).
template <int DU, typename real>
__global__ __launch_bounds__(256) void k_cand_sample(real* cand, const real* centre_in, const int32_t* episode_idx,
                                                     const int32_t* step_idx, int K, int round, int R, uint64_t seed,
                                                     int64_t env_id_base, real u00, real u01, const KParams<real> P) {
  const unsigned n_chunks = (unsigned)(R + 3) / 4u;
  const long b = blockIdx.x;
  const unsigned t = blockIdx.y * blockDim.x + threadIdx.x;  // chunk (k, j) of env b
  __shared__ CandKey skey;
  if (threadIdx.x == 0) skey = cand_subkey(seed, env_id_base + b, episode_idx[b], step_idx[b]);
  __syncthreads();
  if (t >= (unsigned)K * n_chunks) return;
  const int k = (int)(t / n_chunks), j = (int)(t - (unsigned)k * n_chunks);
  const long row = b * K + k;
  const CandKey key = skey;
  real sigma[DU], u0[RCG_MAX_DU] = {u00, u01};
#pragma unroll
  for (int c = 0; c < DU; ++c) sigma[c] = ((real)0.5 * (P.hi[c] - P.lo[c])) * (real)__builtin_ldexp(1.0, -round);  // 2^-round, exact
  float xi0[4] = {0, 0, 0, 0};
  if (!(k >= (K >> 1) && k > 0 && j > 0)) cand_normals4(key, k, 0, round, xi0);  // (a per-step row draws chunk j itself)
  real ce[4], v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * j + e;
    ce[e] = i < R ? (centre_in ? centre_in[b * R + i] : u0[e % DU]) : (real)0;
  }
  cand_chunk<DU, real>(P, key, k, K, j, round, ce, R - 4 * j, sigma, u0, xi0, v);
  real* const out = cand + row * R + 4 * j;
  if ((R & 3) == 0 && (reinterpret_cast<uintptr_t>(cand) & 15) == 0) {  // whole, 16-byte aligned chunks: vector stores
    if constexpr (sizeof(real) == 4) {
      typedef real vec4 __attribute__((ext_vector_type(4)));
      vec4 q;
      q.x = v[0], q.y = v[1], q.z = v[2], q.w = v[3];
      __builtin_nontemporal_store(q, reinterpret_cast<vec4*>(out));
    } else {
      typedef real vec2 __attribute__((ext_vector_type(2)));
      vec2 a, c;
      a.x = v[0], a.y = v[1], c.x = v[2], c.y = v[3];
      __builtin_nontemporal_store(a, reinterpret_cast<vec2*>(out));
      __builtin_nontemporal_store(c, reinterpret_cast<vec2*>(out + 2));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * j + e < R) out[e] = v[e];
  }
}

}  // namespace rcg
