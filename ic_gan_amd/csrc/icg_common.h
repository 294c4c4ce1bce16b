// Shared helpers for the gfx950 kernels of libicgan_hip.so (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/icgan_hip.h"

extern thread_local int g_icg_last_hip_error;

static inline int icg_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_icg_last_hip_error = (int)e;
    return ICG_ERR_LAUNCH;
  }
  return ICG_OK;
}

#define ICG_REQUIRE(cond) \
  do {                    \
    if (!(cond)) return ICG_ERR_ARG; \
  } while (0)

// residual operand of an epilogue: 0 added as is, 1 added with nearest x2 upsampling on read, 2 not added: ReLU mask
static inline int icg_res_mode(unsigned flags) {
  return (flags & ICG_RES_RELU_MASK) ? 2 : ((flags & ICG_RES_UPSAMPLE2X) ? 1 : 0);
}

static inline int64_t icg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Launch geometry of the streaming kernels (transforms, elementwise, bias_act / upfirdn2d): a FULL grid -- one block-iteration per
// workgroup -- up to 2^22 workgroups.  Hand-written copies (tools/hbm_copy.hip, tools/hbm_bench.py) stream at 6.35 TB/s that way on
// MI355X and at 4.8 - 5.5 TB/s as grid-stride loops over 2048 - 8192 workgroups; the kernels keep their loops for larger problems.
// (ICG_GRID_CAP in the environment: measurement switch, read once per process)
#include <stdlib.h>
static inline long icg_grid_cap() {
  static const long cap = [] { const char* e = getenv("ICG_GRID_CAP"); const long v = e ? atol(e) : 0; return v > 0 ? v : (1L << 22); }();
  return cap;
}
#define ICG_GRID_CAP icg_grid_cap()

// 64-wide wavefront reductions (gfx950: wave = 64 lanes)
// Ordering point for LDS data exchanged between the lanes of ONE wavefront (wave-private LDS regions): tells the compiler that
// the LDS stores before it are visible to the LDS loads after it.  A wave executes in lockstep and the LDS queue is in order
// per wave, so this costs nothing at run time -- it pins what the hardware already does against instruction scheduling.
__device__ __forceinline__ void icg_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
