// rcg_sys_inst.hip - the system-templated launchers (rcg_sysops.hpp) and, through them, every kernel of ONE system, split
// into parts that build in parallel (the Makefile compiles this file once per system and part):
//   -DRCG_SYS=Sys3WRobot|Sys3WRobotNI|Sys2Tank  -DRCG_SYS_VT=kVt3WRobot|kVt3WRobotNI|kVt2Tank  -DRCG_SYS_PART=0..4
//   part 0: the table of host function pointers, the light launchers (operators, nominal controller), op_sim_step (k_sim*),
//           op_critic_update (k_critic_fit);  1: op_actor (k_actor and the dispatch to the k_actor_dma objects);
//   2: op_ticks (k_ticks, k_ticks_pk);  3: op_ticks_mem (k_ticks_mem);  4: op_optimize, op_search (k_actor_opt, k_actor_search).
// A launcher is instantiated in exactly one part and declared `extern template` in the others.
#include "rcg_sysops.hpp"

#if !defined(RCG_SYS) || !defined(RCG_SYS_VT) || !defined(RCG_SYS_PART)
#error "compile with -DRCG_SYS=... -DRCG_SYS_VT=... -DRCG_SYS_PART=... (see the Makefile)"
#endif

#define RCG_S rcg::RCG_SYS
#define RCG_OP_SIM(X) X int rcg::op_sim_step<RCG_S>(rcg_handle*, int32_t);
#define RCG_OP_FIT(X) X int rcg::op_critic_update<RCG_S>(rcg_handle*, int32_t, int32_t, int32_t);
#define RCG_OP_ACTOR(X)                                                                                                \
  X int rcg::op_actor<RCG_S>(rcg_handle*, const char*, const void*, int, const void*, const void*, const void*, void*, \
                             void*, void*, int32_t*, bool, bool);
#define RCG_OP_TICKS(X) X int rcg::op_ticks<RCG_S>(rcg_handle*, int32_t, int32_t, const void*);
#define RCG_OP_TICKS_MEM(X) X int rcg::op_ticks_mem<RCG_S>(rcg_handle*, int32_t, int32_t, const void*);
#define RCG_OP_OPT(X)                                                                                                  \
  X int rcg::op_optimize<RCG_S>(rcg_handle*, int32_t, const void*, const void*, const void*, int, void*, void*, void*, \
                                int32_t*, bool, bool);                                                                 \
  X int rcg::op_search<RCG_S>(rcg_handle*, int32_t, int32_t, int32_t, const void*, const void*, const void*, int,      \
                              void*, void*, void*, int32_t*, bool, bool);

#if RCG_SYS_PART != 0
RCG_OP_SIM(extern template)
RCG_OP_FIT(extern template)
#endif
#if RCG_SYS_PART != 1
RCG_OP_ACTOR(extern template)
#endif
#if RCG_SYS_PART != 2
RCG_OP_TICKS(extern template)
#endif
#if RCG_SYS_PART != 3
RCG_OP_TICKS_MEM(extern template)
#endif
#if RCG_SYS_PART != 4
RCG_OP_OPT(extern template)
#endif

#if RCG_SYS_PART == 0
RCG_OP_SIM(template)
RCG_OP_FIT(template)
// Both compilation passes instantiate the launchers (the device pass learns from them which kernels to emit); the table of
// host function pointers itself exists in the host pass only.
template struct rcg::SysInstances<RCG_S>;
#if !defined(__HIP_DEVICE_COMPILE__)
const SysVTable RCG_SYS_VT = rcg::SysInstances<RCG_S>::table();
#endif
#elif RCG_SYS_PART == 1
RCG_OP_ACTOR(template)
#elif RCG_SYS_PART == 2
RCG_OP_TICKS(template)
#elif RCG_SYS_PART == 3
RCG_OP_TICKS_MEM(template)
#elif RCG_SYS_PART == 4
RCG_OP_OPT(template)
#else
#error "RCG_SYS_PART must be 0 .. 4"
#endif
