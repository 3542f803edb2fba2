// One instantiation of the step kernel by itself (tools/dbg/loop_spills.sh): -DKN=16 -DKV=true | -DKN=16 -DKV=false | -DKN=32 -DKV=false
#include <hip/hip_runtime.h>
#include "../../include/snk.h"
#include "../../bullet-envs_amd/csrc/snk_device.hpp"
#ifndef KN
#define KN 16
#define KV true
#endif
template __global__ void snk::env_step_sched_kernel<KN, KV>(snk::StepArgs);
