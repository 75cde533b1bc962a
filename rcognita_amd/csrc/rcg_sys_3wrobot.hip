// rcg_sys_3wrobot.hip - every system-templated kernel and launcher of librcg.so instantiated for Sys3WRobot
// (rcognita/systems.py).  One translation unit per environment so the library builds in parallel.
#include "rcg_sysops.hpp"

// Both compilation passes instantiate the launchers (the device pass learns from them which kernels to
// emit); the table of host function pointers itself exists in the host pass only.
template struct rcg::SysInstances<rcg::Sys3WRobot>;
#if !defined(__HIP_DEVICE_COMPILE__)
const SysVTable kVt3WRobot = rcg::SysInstances<rcg::Sys3WRobot>::table();
#endif
