// rcg_dma_launch.hpp - definition of launch_dma<Sys, real, GROUP> (declared in rcg_actor_dma.hpp): picks the
// k_actor_dma<Sys, real, R, Sys::TGT, V> instance for a runtime row length and variant.  Included only by
// rcg_dma_inst.hip, which is compiled once per (system, element type, group): the ~700 kernel instances of the library
// are spread over 36 objects that build in parallel.
#pragma once

#include "rcg_actor_dma.hpp"
#include "rcg_actor_dma_packed.hpp"

namespace rcg {

template <typename Sys, typename real, int GROUP, int R>
static bool launch_dma_r(int r, int variant, dim3 grid, dim3 block, size_t lds, hipStream_t s, const ActorArgs<real>& A,
                         const KParams<real>& P, hipEvent_t ev_a, hipEvent_t ev_b) {
  if constexpr (R > dma_max_row<real>()) {
    return false;
  } else {
    if (r != R) return launch_dma_r<Sys, real, GROUP, R + 1>(r, variant, grid, block, lds, s, A, P, ev_a, ev_b);
    if constexpr (R % Sys::DU != 0) {
      return false;
    } else {
#define RCG_DMA_CASE(V)                                                                                              \
  case V: {                                                                                                          \
    auto fn = k_actor_dma<Sys, real, R, ((V) >= DMA_MPC_GEND ? true : Sys::TGT), V>;                                 \
    if (lds > 64 * 1024) /* f64 rows beyond 256 bytes: four 64-row tiles exceed the default dynamic-LDS limit */     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                (int)lds);                                                                           \
    if (ev_a)                                                                                                        \
      hipExtLaunchKernelGGL(fn, grid, block, (std::uint32_t)lds, s, ev_a, ev_b, 0, A, P);                            \
    else                                                                                                             \
      hipLaunchKernelGGL(fn, grid, block, lds, s, A, P);                                                             \
    return true;                                                                                                     \
  }
      if constexpr (GROUP == 0) {
        switch (variant) {
          RCG_DMA_CASE(DMA_MPC_G1)
          RCG_DMA_CASE(DMA_MPC)
        }
      } else if constexpr (GROUP == 6) {
        switch (variant) {
          RCG_DMA_CASE(DMA_MPC_GEND)
          RCG_DMA_CASE(DMA_MPC_GENF)
        }
      } else if constexpr (GROUP == 7) {
        switch (variant) {
          RCG_DMA_CASE(DMA_RQL_GEN_0 + RCG_CRITIC_QUAD_LIN)
          RCG_DMA_CASE(DMA_RQL_GEN_0 + RCG_CRITIC_QUADRATIC)
          RCG_DMA_CASE(DMA_RQL_GEN_0 + RCG_CRITIC_QUAD_NOMIX)
          RCG_DMA_CASE(DMA_RQL_GEN_0 + RCG_CRITIC_QUAD_MIX)
        }
      } else if constexpr (GROUP == 1) {
        switch (variant) {
          RCG_DMA_CASE(DMA_SQL_0 + RCG_CRITIC_QUAD_LIN)
          RCG_DMA_CASE(DMA_SQL_0 + RCG_CRITIC_QUADRATIC)
          RCG_DMA_CASE(DMA_SQL_0 + RCG_CRITIC_QUAD_NOMIX)
          RCG_DMA_CASE(DMA_SQL_0 + RCG_CRITIC_QUAD_MIX)
        }
      } else {
        switch (variant) {
          RCG_DMA_CASE(DMA_RQL_0 + RCG_CRITIC_QUAD_LIN)
          RCG_DMA_CASE(DMA_RQL_0 + RCG_CRITIC_QUADRATIC)
          RCG_DMA_CASE(DMA_RQL_0 + RCG_CRITIC_QUAD_NOMIX)
          RCG_DMA_CASE(DMA_RQL_0 + RCG_CRITIC_QUAD_MIX)
        }
      }
#undef RCG_DMA_CASE
      return false;
    }
  }
}

template <typename Sys, typename real, int GROUP>
bool launch_dma(int r, int variant, dim3 grid, dim3 block, size_t lds, hipStream_t s, const ActorArgs<real>& A,
                const KParams<real>& P, hipEvent_t ev_a, hipEvent_t ev_b) {
  return launch_dma_r<Sys, real, GROUP, 1>(r, variant, grid, block, lds, s, A, P, ev_a, ev_b);
}

// ---- k_actor_dma_packed (groups 3: the two MPC variants, 4: SQL x 4 structures, 5: RQL x 4), every row length ---------
template <typename Sys, typename real, int GROUP, int R>
static bool launch_dma_packed_r(int r, int variant, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                                const ActorArgs<real>& A, const KParams<real>& P, hipEvent_t ev_a, hipEvent_t ev_b) {
  if constexpr (R > dma_max_row<real>()) {
    return false;
  } else {
    if (r != R) return launch_dma_packed_r<Sys, real, GROUP, R + 1>(r, variant, grid, block, lds, s, A, P, ev_a, ev_b);
    if constexpr (R % Sys::DU != 0) {
      return false;
    } else {
#define RCG_DMAP_CASE(V)                                                                                             \
  case V: {                                                                                                          \
    if constexpr ((V) >= DMA_RQL_0 &&                                                                                \
                  !packed_critic_ok(dma_dc(((V) >= DMA_SQL_0 ? (V)-DMA_SQL_0 : (V)-DMA_RQL_0), Sys::DS, Sys::DU),    \
                                    (int)sizeof(real))) {                                                            \
      return false; /* too many weights for per-lane registers: no instance, the caller takes another kernel */      \
    } else {                                                                                                         \
      auto fn = k_actor_dma_packed<Sys, real, R, Sys::TGT, V>;                                                       \
      if (lds > 64 * 1024)                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                  (int)lds);                                                                         \
      if (ev_a)                                                                                                      \
        hipExtLaunchKernelGGL(fn, grid, block, (std::uint32_t)lds, s, ev_a, ev_b, 0, A, P);                          \
      else                                                                                                           \
        hipLaunchKernelGGL(fn, grid, block, lds, s, A, P);                                                           \
      return true;                                                                                                   \
    }                                                                                                                \
  }
      if constexpr (GROUP == 3) {
        switch (variant) {
          RCG_DMAP_CASE(DMA_MPC_G1)
          RCG_DMAP_CASE(DMA_MPC)
        }
      } else if constexpr (GROUP == 4) {
        switch (variant) {
          RCG_DMAP_CASE(DMA_SQL_0 + RCG_CRITIC_QUAD_LIN)
          RCG_DMAP_CASE(DMA_SQL_0 + RCG_CRITIC_QUADRATIC)
          RCG_DMAP_CASE(DMA_SQL_0 + RCG_CRITIC_QUAD_NOMIX)
          RCG_DMAP_CASE(DMA_SQL_0 + RCG_CRITIC_QUAD_MIX)
        }
      } else {
        switch (variant) {
          RCG_DMAP_CASE(DMA_RQL_0 + RCG_CRITIC_QUAD_LIN)
          RCG_DMAP_CASE(DMA_RQL_0 + RCG_CRITIC_QUADRATIC)
          RCG_DMAP_CASE(DMA_RQL_0 + RCG_CRITIC_QUAD_NOMIX)
          RCG_DMAP_CASE(DMA_RQL_0 + RCG_CRITIC_QUAD_MIX)
        }
      }
#undef RCG_DMAP_CASE
      return false;
    }
  }
}

template <typename Sys, typename real, int GROUP>
bool launch_dma_packed(int r, int variant, dim3 grid, dim3 block, size_t lds, hipStream_t s, const ActorArgs<real>& A,
                       const KParams<real>& P, hipEvent_t ev_a, hipEvent_t ev_b) {
  return launch_dma_packed_r<Sys, real, GROUP, 1>(r, variant, grid, block, lds, s, A, P, ev_a, ev_b);
}

}  // namespace rcg
