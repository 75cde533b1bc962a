// rcg_dma_inst.hip - the k_actor_dma instances of ONE (system, element type, group), selected by the Makefile:
//   -DRCG_INST_SYS=Sys3WRobot|Sys3WRobotNI|Sys2Tank  -DRCG_INST_REAL=float|double  -DRCG_INST_GROUP=0|1|2|3|4|5|6|7
// (group 0: MPC gamma == 1, MPC discounted; group 1: SQL x 4 critic structures; group 2: RQL x 4; groups 3 / 4 / 5: k_actor_dma_packed, the two MPC variants / SQL x 4 / RQL x 4; group 6: k_actor_dma for MPC with a cost structure
// no preset has - DMA_MPC_GEND, DMA_MPC_GENF; group 7: the same for RQL x 4 critic structures).
#include "rcg_dma_launch.hpp"

#if !defined(RCG_INST_SYS) || !defined(RCG_INST_REAL) || !defined(RCG_INST_GROUP)
#error "compile with -DRCG_INST_SYS=... -DRCG_INST_REAL=... -DRCG_INST_GROUP=... (see the Makefile)"
#endif

#if RCG_INST_GROUP >= 3 && RCG_INST_GROUP <= 5
template bool rcg::launch_dma_packed<rcg::RCG_INST_SYS, RCG_INST_REAL, RCG_INST_GROUP>(int, int, dim3, dim3, size_t, hipStream_t,
                                                                                   const rcg::ActorArgs<RCG_INST_REAL>&,
                                                                                   const rcg::KParams<RCG_INST_REAL>&,
                                                                                   hipEvent_t, hipEvent_t);
#else
template bool rcg::launch_dma<rcg::RCG_INST_SYS, RCG_INST_REAL, RCG_INST_GROUP>(int, int, dim3, dim3, size_t, hipStream_t,
                                                                            const rcg::ActorArgs<RCG_INST_REAL>&,
                                                                            const rcg::KParams<RCG_INST_REAL>&, hipEvent_t,
                                                                            hipEvent_t);
#endif
