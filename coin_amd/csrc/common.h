// Shared device helpers for libcoin_hip (gfx950 only: wave = 64, MFMA, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/coin_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define COIN_WAVE 64

#define COIN_CHECK_PTR(p) \
  do {                    \
    if ((p) == nullptr) return COIN_EINVAL; \
  } while (0)

static inline int coin_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? COIN_OK : (int)e;
}

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  typedef f32x4 type;
};
template <>
struct Vec16<bf16_t> {
  static constexpr int N = 8;
  typedef bf16x8 type;
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (wid == 0) {
    r = lane < nw ? red[lane] : 0.f;
    r = wave_reduce_sum(r);
  }
  __syncthreads();
  return r;  // valid in wave 0
}
