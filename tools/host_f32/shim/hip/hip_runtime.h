// Host stand-in for <hip/hip_runtime.h>: tools/host_f32 compiles mrs_device.hpp for the CPU (diagnostics only).
#pragma once
#include <cmath>
#include <cstdint>
using std::isnan;
#define __device__
#define __host__
#define __forceinline__ inline
static inline float shim_med3(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }
static inline bool shim_class_normal(float x) { return std::isnormal(x); }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
#define __builtin_amdgcn_exp2f(x) (::exp2f(x))
#define __builtin_amdgcn_fmed3f(a, b, c) shim_med3(a, b, c)
#define __builtin_amdgcn_ballot_w64(x) ((x) ? 1ull : 0ull)
#define __builtin_amdgcn_class(x, m) shim_class_normal(x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
