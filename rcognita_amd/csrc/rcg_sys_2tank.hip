// rcg_sys_2tank.hip - every system-templated kernel and launcher of librcg.so instantiated for Sys2Tank
// (rcognita/systems.py).  One translation unit per environment so the library builds in parallel.
#include "rcg_sysops.hpp"

// Both compilation passes instantiate the launchers (the device pass learns from them which kernels to
// emit); the table of host function pointers itself exists in the host pass only.
template struct rcg::SysInstances<rcg::Sys2Tank>;
#if !defined(__HIP_DEVICE_COMPILE__)
const SysVTable kVt2Tank = rcg::SysInstances<rcg::Sys2Tank>::table();
#endif
