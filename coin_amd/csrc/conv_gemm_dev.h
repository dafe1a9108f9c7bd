// Device helpers shared by the GEMM cores (conv_gemm_p8.hip: 256 x 256 persistent tiles; conv_gemm_s4.hip: 128 x 128 tiles for the
// backbone's small maps): range-checked buffer LDS-DMA, raw barriers, bf16 pair unpacking, the statistics epilogue's cross-lane sums.
#pragma once
#include "common.h"

// Range-checked buffer LDS-DMA of 16 bytes per lane: address = base + voff (per lane) + soff (wave-uniform); a voff at or beyond
// `bytes` writes zeros.  (A plain function on purpose: called with these builtins directly, function TEMPLATES are rejected by the host
// pass of hipcc 7.2 with a bare "substitution failure".)
__device__ __forceinline__ void blds16(const void* base, unsigned bytes, unsigned voff, unsigned soff, void* lds_wave_base) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

constexpr unsigned P8_OOB = 0x80000000u;  // a buffer offset beyond every operand (host: sizes < 2^31): the range-checked DMA writes zeros

#define P8_SCHED() __builtin_amdgcn_sched_barrier(0)
// raw barrier (no counter is waited for: LDS-DMA stays in flight across it) + a compiler-level memory fence
#define P8_BAR()                      \
  do {                                \
    asm volatile("" ::: "memory");    \
    __builtin_amdgcn_s_barrier();     \
    asm volatile("" ::: "memory");    \
  } while (0)
// epilogue barrier: this wave's LDS accesses are complete, then the workgroup barrier; global loads / stores / DMA are NOT waited for
#define P8_LDS_SYNC()                                   \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
    P8_SCHED();                                         \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
// 8 bf16 -> 4 pairs of floats (element 2k in .x, 2k + 1 in .y): one shift / one mask per element, exact
__device__ __forceinline__ void p8_pairs(const bf16x8& v, f32x2 (&o)[4]) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k].x = __builtin_bit_cast(float, w[k] << 16);
    o[k].y = __builtin_bit_cast(float, w[k] & 0xffff0000u);
  }
}

// t[i] <- sum of t[i] over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four 16-lane rows of the wave), in every lane.
// v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src;  v_permlane16_swap: odd rows of vdst <-> even rows of src.
// With vdst = src = x the two results add up to the pair sums.  (`s_nop 1`: a vector-ALU write of a swap operand needs two wait
// states before the swap reads it; inline asm is not covered by the compiler's hazard recogniser.)
__device__ __forceinline__ void p8_rows_sum(float (&t)[8]) {
  float u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = t[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t[i]), "+v"(u[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t[i] += u[i];
    u[i] = t[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(t[i]), "+v"(u[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] += u[i];
}

// row index of a pooled residual (see P8Args::rp_w): output pixel `grow` of a [*, rp_h, rp_w] grid -> row of the floor-pooled
// [*, rp_h / 2, rp_w / 2] grid and the weight of its contribution (0.25, or 0 where the pixel falls off the pooled map).
// magic = ceil(2^32 / d): umulhi(x, magic) is floor(x / d) or one more (the excess x * (magic * d - 2^32) / (d * 2^32) is below 1 for
// every 32-bit x) -- the fix-up makes both quotients exact for every map size (round-4 ADVICE: without it [2, 200, 336] maps lost
// the last pixel of an image)
__device__ __forceinline__ size_t p8_pooled_row(int grow, int rp_h, int rp_w, unsigned magic_hw, unsigned magic_w, float& scale) {
  int n = (int)__umulhi((unsigned)grow, magic_hw);
  n -= (unsigned)n * (unsigned)(rp_h * rp_w) > (unsigned)grow ? 1 : 0;
  const int rem = grow - n * (rp_h * rp_w);
  int h = (int)__umulhi((unsigned)rem, magic_w);
  h -= h * rp_w > rem ? 1 : 0;
  const int w = rem - h * rp_w;
  const int oh = h >> 1, ow = w >> 1, OH = rp_h >> 1, OW = rp_w >> 1;
  scale = (oh < OH && ow < OW) ? 0.25f : 0.f;
  return ((size_t)n * OH + (oh < OH ? oh : OH - 1)) * OW + (ow < OW ? ow : OW - 1);
}
