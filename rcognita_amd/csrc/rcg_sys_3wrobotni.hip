// rcg_sys_3wrobotni.hip - every system-templated kernel and launcher of librcg.so instantiated for Sys3WRobotNI
// (rcognita/systems.py).  One translation unit per environment so the library builds in parallel.
#include "rcg_sysops.hpp"

// Both compilation passes instantiate the launchers (the device pass learns from them which kernels to
// emit); the table of host function pointers itself exists in the host pass only.
template struct rcg::SysInstances<rcg::Sys3WRobotNI>;
#if !defined(__HIP_DEVICE_COMPILE__)
const SysVTable kVt3WRobotNI = rcg::SysInstances<rcg::Sys3WRobotNI>::table();
#endif
