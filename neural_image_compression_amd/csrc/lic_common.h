// Shared helpers for the gfx950 kernels of liblic_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lic.h"

#define LIC_EXPORT extern "C" __attribute__((visibility("default")))

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern thread_local int g_lic_last_hip_error;

static inline int lic_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_lic_last_hip_error = (int)e;
    return LIC_ERR_LAUNCH;
  }
  return LIC_OK;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Column pitch of the packed fp32 MFMA weight operand (lic_pack_weight / lic_prep_run / lic_igemm must agree):
// whole 64-column wave pairs, so that every layer wider than one 32-column tile has a full-N tiling and runs
// the branch-free LDS-DMA variant -- the 288-wide hyper-decoder layers (Components.py:101-103) and the 80-wide
// column matrix of the RGB head ran the ragged register-staged variant at half its speed with ceil32.
static inline int lic_npad_f32(int N) { return N <= 32 ? 32 : ((N + 63) / 64) * 64; }

// grid for grid-stride elementwise kernels: enough blocks to fill 256 CUs x 8, no more
static inline int ew_grid(int64_t n, int per_block) {
  int64_t g = cdiv64(n, per_block);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float lic_softplus(float v) { return v > 20.0f ? v : log1pf(expf(v)); }
__device__ __forceinline__ float lic_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float lic_softplus_grad(float v) { return v > 20.0f ? 1.0f : lic_sigmoid(v); }

// compressai NonNegativeParametrizer forward: LowerBound(p, bound)^2 - pedestal.
// One definition for lic_gdn_reparam and lic_prep's transform, so both produce the same bits.
__device__ __forceinline__ float lic_reparam(float p, float bound, float pedestal) {
  const float v = p > bound ? p : bound;
  return v * v - pedestal;
}

// wave64 sum via DPP-free shuffles
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
