// bf16 GEMM / implicit-GEMM convolution for the res5 bottlenecks on the RoI tiles (gfx950).
//
// Replaces the library convolutions of `backbone.layer4` on the RoI tiles (coin/modeling/utils.py:77-90,184-186 run through
// coin/modeling/roi_heads/clip_roi_heads.py:172-176): the 1x1 convolutions are plain NHWC GEMMs [R*hw, Cin] x [Cout, Cin]^T,
// the 3x3 / pad 1 convolutions implicit GEMMs with K = 9*Cin (tap-major, the channels-last weight layout).  Forward and
// data-gradient use this kernel (dgrad = the same contraction with the weight re-laid [Cin][flipped tap][Cout]).
//
// Why hand-written: at these shapes (M = 100 352 or 401 408 rows, N = 512..2048, K = 512..4608) the 1x1 GEMMs sit ON the HBM
// roofline (342 FLOP/B for layer4.0.conv1 against a machine balance of ~400): what bounds them is how the output tile leaves
// the chip and what else can be done while it is on chip.  So:
//   * 256 x 128 x 64 block tile, 8 waves (4 x 2), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 accumulators;
//   * operands staged by 16-byte LDS-DMA (global_load_lds) into a 3-deep ring (3 x 48 KiB), XOR-swizzled on the SOURCE address
//     (rule 21) so that the ds_read_b128 fragment reads are conflict-free; ONE raw s_barrier per K-step, counted
//     `s_waitcnt vmcnt(6)`: the loads of step t+1 stay in flight across the barrier while step t is multiplied and step t+2
//     is issued;
//   * the 3x3 gather is done by the DMA's per-lane source address (a shifted pixel, or a zero page outside the image);
//   * MFMA operands swapped (D = W-frag x A-frag) so that a lane ends up with 4 CONSECUTIVE output channels of one pixel:
//     the tile is packed to bf16 in registers, transposed through LDS once and stored as whole 256-byte rows;
//   * the BatchNorm statistics of the output (per-channel pivoted sum and sum of squares of the STORED bf16 values, per row
//     tile) are taken during that store pass -- the separate statistics pass over the activation (coin_bn_stats) disappears.
#include <stdlib.h>
#include "common.h"
#include "conv_gemm_p8.h"

namespace {

constexpr int GM = 256, GN = 128, GK = 64;
constexpr int A_BYTES = GM * GK * 2;              // 32 KiB
constexpr int B_BYTES = GN * GK * 2;              // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;    // 48 KiB
constexpr int NSTAGE = 3;
constexpr int C_ROW_BYTES = GN * 2 + 16;          // padded row of the output tile image

__device__ __attribute__((aligned(256))) unsigned char coin_zero_page[256];  // source of the 3x3 taps that fall outside the image

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

struct ARows {             // the 4 A-tile rows this thread stages (fixed for the whole K loop)
  const bf16_t* base[4];   // row start (1x1: the matrix row; 3x3: the centre pixel's channel 0)
  unsigned taps[4];        // 3x3: bit t set = tap t lies inside the image
};

template <bool GATHER3, bool STATS>
__global__ __launch_bounds__(512) void conv_gemm_bf16_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ Cmat, int ldc, int M, int N,
    int K, int H, int W, int Cin, float* __restrict__ stats, int64_t stats_rows, int tiles_m, int tiles_n,
    const bf16_t* __restrict__ Rmat, int ldr) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware tile order: consecutive tiles along N (sharing the A panel) stay on one XCD
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * GM, n0 = tn * GN;
  const int wr = wave >> 1, wc = wave & 1;  // 4 x 2 waves

  // ---- per-thread staging addresses
  const int rl = lane >> 3;            // row within the 8-row group one DMA instruction writes
  const int sc = (lane & 7) ^ rl;      // logical 16-byte chunk that must land at physical chunk lane & 7
  ARows ar;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int row = m0 + (wave * 4 + j) * 8 + rl;
    row = row < M ? row : M - 1;
    if (GATHER3) {
      const int hw = H * W;
      const int nb = row / hw, rem = row - nb * hw;
      const int oy = rem / W, ox = rem - oy * W;
      ar.base[j] = A + (size_t)row * Cin;
      unsigned m = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = oy + t / 3 - 1, xx = ox + t % 3 - 1;
        m |= (yy >= 0 && yy < H && xx >= 0 && xx < W ? 1u : 0u) << t;
      }
      ar.taps[j] = m;
    } else {
      ar.base[j] = A + (size_t)row * lda;
      ar.taps[j] = 0;
    }
  }
  const bf16_t* brow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = n0 + (wave * 2 + j) * 8 + rl;
    row = row < N ? row : N - 1;
    brow[j] = B + (size_t)row * ldb;
  }
  const int kc = GATHER3 ? Cin / GK : 1;  // K-steps per tap

  auto stage = [&](int t, int slot) {
    char* sa = lds + slot * STAGE_BYTES;
    char* sb = sa + A_BYTES;
    int koff = t * GK, tap = 0;
    long long shift = 0;
    if (GATHER3) {
      tap = t / kc;
      koff = (t - tap * kc) * GK;
      shift = ((long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * Cin;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t* src = ar.base[j] + koff + sc * 8;
      if (GATHER3) src = ((ar.taps[j] >> tap) & 1u) ? src + shift : reinterpret_cast<const bf16_t*>(coin_zero_page) + sc * 8;
      glds16(src, sa + (wave * 4 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(brow[j] + (size_t)t * GK + sc * 8, sb + (wave * 2 + j) * 1024);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = K / GK;
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  const int fr = lane & 15, fq = lane >> 4;
  for (int t = 0; t < nt; ++t) {
    // stage t has landed once at most the 6 loads of stage t+1 are outstanding; the barrier publishes every wave's DMA writes
    // and guarantees that slot (t+2) % 3 == (t-1) % 3 is no longer being read
    if (t + 1 < nt)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) stage(t + 2, (t + 2) % NSTAGE);
    const char* la = lds + (t % NSTAGE) * STAGE_BYTES;
    const char* lb = la + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(la, wr * 64 + i * 16 + fr, kk * 4 + fq);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = frag(lb, wc * 64 + j * 16 + fr, kk * 4 + fq);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // operands swapped: D[n][m] -> lane fr = output row m, registers = 4 consecutive output columns n
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- epilogue: registers -> bf16 tile image in LDS -> whole rows to HBM (+ statistics of the stored values)
  __syncthreads();  // every wave is done with the staging ring
  char* ct = lds;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wr * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wc * 64 + j * 16 + fq * 4;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[i][j][r];
      *reinterpret_cast<bf16x4*>(ct + row * C_ROW_BYTES + col * 2) = o;
    }
  }
  __syncthreads();
  const int chunk = threadIdx.x & 15, rsub = threadIdx.x >> 4;  // 16 chunks of 8 columns per row, 32 rows per pass
  const int gcol = n0 + chunk * 8;
  float s1[8], s2[8], piv[8];
  if (STATS) {
    const bf16x8 p = *reinterpret_cast<const bf16x8*>(ct + chunk * 16);  // the tile's first row is every thread's pivot
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      piv[i] = (float)p[i];
      s1[i] = s2[i] = 0.f;
    }
  }
#pragma unroll
  for (int p = 0; p < GM / 32; ++p) {
    const int row = p * 32 + rsub;
    const int grow = m0 + row;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(ct + row * C_ROW_BYTES + chunk * 16);
    if (Rmat != nullptr && grow < M && gcol < N) {  // C = A.B^T + R: bf16 product, then a bf16 add (what two separate launches store)
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(Rmat + (size_t)grow * ldr + gcol);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (bf16_t)((float)v[i] + (float)r[i]);
    }
    if (grow < M && gcol < N) *reinterpret_cast<bf16x8*>(Cmat + (size_t)grow * ldc + gcol) = v;
    if (STATS && (int64_t)grow < stats_rows) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = (float)v[i] - piv[i];
        s1[i] += d;
        s2[i] += d * d;
      }
    }
  }
  if (STATS) {
    // threads with equal `chunk`: lanes l, l^16, l^32 inside a wave, then the 8 waves through LDS (fixed order)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s1[i] += __shfl_xor(s1[i], 16, 64);
      s2[i] += __shfl_xor(s2[i], 16, 64);
      s1[i] += __shfl_xor(s1[i], 32, 64);
      s2[i] += __shfl_xor(s2[i], 32, 64);
    }
    __syncthreads();  // the tile image has been consumed
    float* red = reinterpret_cast<float*>(lds + GM * C_ROW_BYTES);  // [8 waves][16 chunks][16]
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red[(wave * 16 + lane) * 16 + i] = s1[i];
        red[(wave * 16 + lane) * 16 + 8 + i] = s2[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < GN) {  // one thread per column of the tile
      const int c = threadIdx.x, ch = c >> 3, ci = c & 7;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a1 += red[(w * 16 + ch) * 16 + ci];
        a2 += red[(w * 16 + ch) * 16 + 8 + ci];
      }
      if (n0 + c < N) {
        float* __restrict__ part = stats + (size_t)tm * 3 * N + n0 + c;
        part[0] = (float)*reinterpret_cast<const bf16_t*>(ct + c * 2);  // pivot
        part[N] = a1;
        part[2 * (size_t)N] = a2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// 256 x 256 x 32 variant (N % 256 == 0): the 256 x 128 kernel above is bound by the L2 -> LDS stream (measured: 48 KiB per K-step
// arrive at ~42 GB/s per CU, 1.13 us per step against 0.43 us of MFMA time).  A square tile needs 1.5x fewer operand bytes per
// FLOP: (256 + 256) x 32 x 2 B = 32 KiB per 4.2 MFLOP step.  8 waves as 2 (M) x 4 (N), each 128 x 64 = 8 x 4 accumulators
// (128 VGPRs); 4-deep ring of 32 KiB stages (128 KiB), the loads of steps t+1 and t+2 stay in flight across the barrier of step t
// (`s_waitcnt vmcnt(8)`).  Rows are 64 B (32 k): one DMA instruction covers 16 rows; chunk c of row r is stored at physical
// chunk c ^ ((-(r >> 2)) & 3), which makes every 16-lane group of the ds_read_b128 fragment reads hit 16 distinct 16-byte slots
// (lane groups of MI355X_MICROARCH.md section LDS).  The epilogue goes through LDS in two 128-column halves.
// ------------------------------------------------------------------------------------------
constexpr int QM = 256, QN = 256, QK = 32;
constexpr int QA_BYTES = QM * QK * 2;             // 16 KiB
constexpr int QSTAGE_BYTES = QA_BYTES * 2;        // 32 KiB
constexpr int QNSTAGE = 4;

__device__ __forceinline__ bf16x8 qfrag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4));
}

template <bool GATHER3, bool STATS>
__global__ __launch_bounds__(512) void conv_gemm256_bf16_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ Cmat, int ldc, int M, int N,
    int K, int H, int W, int Cin, float* __restrict__ stats, int64_t stats_rows, int tiles_m, int tiles_n,
    const bf16_t* __restrict__ Rmat, int ldr) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * QM, n0 = tn * QN;
  const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves

  // ---- per-thread staging addresses: one DMA instruction = 16 rows x 64 B; this thread: row (lane >> 2) of the group, and the
  // logical chunk that must land at physical chunk lane & 3
  const int rl = lane >> 2;
  const int sc = (lane & 3) ^ ((0 - (rl >> 2)) & 3);
  const bf16_t* abase[2];
  unsigned ataps[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = m0 + (wave * 2 + j) * 16 + rl;
    row = row < M ? row : M - 1;
    if (GATHER3) {
      const int hw = H * W;
      const int nb = row / hw, rem = row - nb * hw;
      const int oy = rem / W, ox = rem - oy * W;
      abase[j] = A + (size_t)row * Cin;
      unsigned m = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = oy + t / 3 - 1, xx = ox + t % 3 - 1;
        m |= (yy >= 0 && yy < H && xx >= 0 && xx < W ? 1u : 0u) << t;
      }
      ataps[j] = m;
    } else {
      abase[j] = A + (size_t)row * lda;
      ataps[j] = 0;
    }
  }
  const bf16_t* brow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = n0 + (wave * 2 + j) * 16 + rl;
    row = row < N ? row : N - 1;
    brow[j] = B + (size_t)row * ldb;
  }
  const int kc = GATHER3 ? Cin / QK : 1;  // K-steps per tap

  auto stage = [&](int t, int slot) {
    char* sa = lds + slot * QSTAGE_BYTES;
    char* sb = sa + QA_BYTES;
    int koff = t * QK, tap = 0;
    long long shift = 0;
    if (GATHER3) {
      tap = t / kc;
      koff = (t - tap * kc) * QK;
      shift = ((long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * Cin;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16_t* src = abase[j] + koff + sc * 8;
      if (GATHER3) src = ((ataps[j] >> tap) & 1u) ? src + shift : reinterpret_cast<const bf16_t*>(coin_zero_page) + sc * 8;
      glds16(src, sa + (wave * 2 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(brow[j] + (size_t)t * QK + sc * 8, sb + (wave * 2 + j) * 1024);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = K / QK;
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  if (nt > 2) stage(2, 2);
  const int fr = lane & 15, fq = lane >> 4;
  for (int t = 0; t < nt; ++t) {
    // stage t has landed once at most the 4 + 4 loads of stages t+1, t+2 are outstanding
    if (t + 2 < nt)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 1 < nt)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 3 < nt) stage(t + 3, (t + 3) % QNSTAGE);
    const char* la = lds + (t % QNSTAGE) * QSTAGE_BYTES;
    const char* lb = la + QA_BYTES;
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = qfrag(lb, wc * 64 + j * 16 + fr, fq);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = qfrag(la, wr * 128 + i * 16 + fr, fq);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // swapped: lane = row m, regs = 4 columns n
    __builtin_amdgcn_s_setprio(0);
  }

  // ---- epilogue in two 128-column halves: waves wc = {0,1} then {2,3} park their tiles in LDS, all threads store whole rows
  const int chunk = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  char* ct = lds;
  float* red = reinterpret_cast<float*>(lds + QM * C_ROW_BYTES);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();  // ring (half 0) / previous half's image and reduction scratch (half 1) are free
    if ((wc >> 1) == half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wr * 128 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = (wc & 1) * 64 + j * 16 + fq * 4;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[i][j][r];
          *reinterpret_cast<bf16x4*>(ct + row * C_ROW_BYTES + col * 2) = o;
        }
      }
    }
    __syncthreads();
    const int gcol = n0 + half * 128 + chunk * 8;
    float s1[8], s2[8], piv[8];
    if (STATS) {
      const bf16x8 p = *reinterpret_cast<const bf16x8*>(ct + chunk * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        piv[i] = (float)p[i];
        s1[i] = s2[i] = 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < QM / 32; ++p) {
      const int row = p * 32 + rsub;
      const int grow = m0 + row;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(ct + row * C_ROW_BYTES + chunk * 16);
      if (Rmat != nullptr && grow < M && gcol < N) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(Rmat + (size_t)grow * ldr + gcol);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)((float)v[i] + (float)r[i]);
      }
      if (grow < M && gcol < N) *reinterpret_cast<bf16x8*>(Cmat + (size_t)grow * ldc + gcol) = v;
      if (STATS && (int64_t)grow < stats_rows) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = (float)v[i] - piv[i];
          s1[i] += d;
          s2[i] += d * d;
        }
      }
    }
    if (STATS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s1[i] += __shfl_xor(s1[i], 16, 64);
        s2[i] += __shfl_xor(s2[i], 16, 64);
        s1[i] += __shfl_xor(s1[i], 32, 64);
        s2[i] += __shfl_xor(s2[i], 32, 64);
      }
      if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[(wave * 16 + lane) * 16 + i] = s1[i];
          red[(wave * 16 + lane) * 16 + 8 + i] = s2[i];
        }
      }
      __syncthreads();
      if (threadIdx.x < 128) {
        const int c = threadIdx.x, ch = c >> 3, ci = c & 7;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          a1 += red[(w * 16 + ch) * 16 + ci];
          a2 += red[(w * 16 + ch) * 16 + 8 + ci];
        }
        const int gc = n0 + half * 128 + c;
        if (gc < N) {
          float* __restrict__ part = stats + (size_t)tm * 3 * N + gc;
          part[0] = (float)*reinterpret_cast<const bf16_t*>(ct + c * 2);
          part[N] = a1;
          part[2 * (size_t)N] = a2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient ("TN" contraction over the pixels):  dW[co][(tap, ci)] = sum_m gy[m][co] * xcol[m][(tap, ci)]
// Both operands are stored pixel-major (the NHWC activations), i.e. the contraction index is the ROW index of both: the MFMA
// fragments (8 consecutive k per lane) are columns of the staged tiles, read with ds_read_b64_tr_b16 (the LDS transpose read).
// Block = 256 output channels x 256 input channels of ONE tap x one slice of the pixels; 8 waves (2 x 4), 128 x 64 per wave;
// K-step = 32 pixels: gy[32][256] and x[32][256] rows (512 B each) arrive by LDS-DMA into a 4-deep ring; 16-byte chunk c of row r
// lands at chunk c ^ ((r & 7) << 1) (source-side swizzle), which spreads the 8 rows a half-wave reads per transpose read over
// all 64 banks.  Fragment element j of lane group g is pixel 4g + j (j < 4) / 16 + 4g + j - 4 (j >= 4) of the step -- the same
// permutation for both operands, so the contraction is unchanged.  Pixels beyond M and 3x3 taps outside the image read a zero
// page.  Each slice writes an fp32 slab; a second kernel sums the slabs in slice order (bit-reproducible, no atomics).
// ------------------------------------------------------------------------------------------
constexpr int WB = 256, WK = 32, WROW = 512, WTILE = WK * WROW;  // 16 KiB per operand tile
constexpr int WNSTAGE = 4;

__device__ __attribute__((aligned(512))) unsigned char coin_zero_row[512];

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int col0, int lane) {
  // columns col0 .. col0+15 (lane & 15) x this lane group's 8 pixels
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int col = col0 + 4 * p;
  const int r0 = 4 * g + q, r1 = 16 + 4 * g + q;
  const int chunk = col >> 3, sub = (col & 7) * 2;
  const char* a0 = tile + r0 * WROW + ((chunk ^ ((r0 & 7) << 1)) << 4) + sub;
  const char* a1 = tile + r1 * WROW + ((chunk ^ ((r1 & 7) << 1)) << 4) + sub;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool GATHER3>
__global__ __launch_bounds__(512) void conv_wgrad_bf16_kernel(
    const bf16_t* __restrict__ GY, const bf16_t* __restrict__ X, float* __restrict__ slab, int M, int Cout, int Cin, int Ktot, int H, int W,
    int tiles_co, int tiles_k, int m_chunk) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x % (tiles_co * tiles_k), slice = blockIdx.x / (tiles_co * tiles_k);
  const int tco = tile / tiles_k, tk = tile - tco * tiles_k;
  const int co0 = tco * WB, k0 = tk * WB;
  const int tap = GATHER3 ? k0 / Cin : 0, ci0 = GATHER3 ? k0 - tap * Cin : k0;
  const long long shift = GATHER3 ? ((long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * Cin : 0;
  const int m_begin = slice * m_chunk, m_end = (m_begin + m_chunk) < M ? (m_begin + m_chunk) : M;
  const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves: 128 (co) x 64 (ci) each

  // staging: one DMA instruction = 2 rows x 512 B; this wave stages rows 4*wave .. 4*wave+3 of each operand tile per step
  const int srow = lane >> 5;  // row within the instruction
  const int pch = lane & 31;   // physical 16-byte chunk within the row

  auto stage = [&](int step, int slot) {
    char* sg = lds + slot * 2 * WTILE;
    char* sx = sg + WTILE;
    const int mb = m_begin + step * WK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 4 + j * 2 + srow;      // row of the tile
      const int m = mb + r;
      const int lch = pch ^ ((r & 7) << 1);        // logical chunk that must land at physical chunk pch
      const bool in = m < m_end;
      const bf16_t* gsrc = in ? GY + (size_t)m * Cout + co0 + lch * 8 : reinterpret_cast<const bf16_t*>(coin_zero_row) + lch * 8;
      glds16(gsrc, sg + (wave * 4 + j * 2) * WROW);
      bool ok = in;
      if (GATHER3 && in) {
        const int hw = H * W;
        const int rem = m % hw;
        const int oy = rem / W, ox = rem - oy * W;
        const int yy = oy + tap / 3 - 1, xx = ox + tap % 3 - 1;
        ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      }
      const bf16_t* xsrc = ok ? X + ((long long)m * Cin + shift) + ci0 + lch * 8 : reinterpret_cast<const bf16_t*>(coin_zero_row) + lch * 8;
      glds16(xsrc, sx + (wave * 4 + j * 2) * WROW);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsteps = (m_end - m_begin + WK - 1) / WK;
  if (nsteps > 0) stage(0, 0);
  if (nsteps > 1) stage(1, 1);
  if (nsteps > 2) stage(2, 2);
  for (int t = 0; t < nsteps; ++t) {
    if (t + 2 < nsteps)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 3 < nsteps) stage(t + 3, (t + 3) % WNSTAGE);
    const char* lg = lds + (t % WNSTAGE) * 2 * WTILE;
    const char* lx = lg + WTILE;
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(lx, wc * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = tr_frag(lg, wr * 128 + i * 16, lane);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  // D[row = co][col = ci]: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* __restrict__ out = slab + (size_t)slice * Cout * Ktot;
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = k0 + wc * 64 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = co0 + wr * 128 + i * 16 + fq * 4 + r;
        out[(size_t)row * Ktot + col] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int slices) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 a = *reinterpret_cast<const f32x4*>(slab + i);
  for (int s = 1; s < slices; ++s) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(slab + (size_t)s * n + i);
    a += b;
  }
  *reinterpret_cast<f32x4*>(dw + i) = a;
}

// mean / rstd (+ running statistics) from the per-row-tile pivoted partials: tile t holds n_t = clamp(rows - tile_rows t, 0, tile_rows) values
// per channel as (pivot p, S1 = sum(x - p), S2 = sum((x - p)^2)) -> (mean_t, M2_t) -> Chan's pairwise update, in a fixed order.
// Block = 16 channels x 64 tile lanes (thread = channel c16 + 16 * tile lane): a thread folds tiles lane, lane + 64, ... (a serial
// chain of ~tiles/64 dependent updates; the [401 408, 512] res5 outputs have 1568 tiles), then one thread per channel folds the 64
// partial results in lane order.  N/16 workgroups (32..128) instead of N/64: the launch was latency-bound at 8 workgroups.
// CPB = channels per workgroup, 1024 / CPB tile lanes; SF_U tiles per round.
template <int CPB, int SF_U>
__global__ __launch_bounds__(1024) void conv_stats_finalize_kernel(const float* __restrict__ part, int tiles_m, int N, int64_t rows,
                                                                    float eps, float momentum, float* __restrict__ mean,
                                                                    float* __restrict__ rstd, float* __restrict__ running_mean,
                                                                    float* __restrict__ running_var, int64_t* __restrict__ nbt, int tile_rows) {
  constexpr int TL = 1024 / CPB;
  __shared__ float red[TL][3][CPB];
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;   // nn.BatchNorm2d's counter rides along (it was a launch of its own)
  const int cl = threadIdx.x % CPB, tl = threadIdx.x / CPB;  // tile lane 0..TL-1
  const int c = blockIdx.x * CPB + cl;
  float n_a = 0.f, mu_a = 0.f, m2_a = 0.f;
  auto fold = [&](float n_b, float mu_b, float m2_b) {
    if (n_b == 0.f) return;
    const float n = n_a + n_b, d = mu_b - mu_a;
    mu_a += d * (n_b / n);
    m2_a += m2_b + d * d * (n_a * n_b / n);
    n_a = n;
  };
  if (c < N) {
    for (int t0 = tl; t0 < tiles_m; t0 += TL * SF_U) {  // SF_U tiles per round: their loads are in flight together (the chain is latency-bound)
      float pv[SF_U], s1[SF_U], s2[SF_U], nb[SF_U];
#pragma unroll
      for (int u = 0; u < SF_U; ++u) {
        const int t = t0 + TL * u;
        const int64_t left = rows - (int64_t)t * tile_rows;
        nb[u] = t < tiles_m ? (float)(left <= 0 ? 0 : (left < tile_rows ? left : tile_rows)) : 0.f;
        const float* __restrict__ p = part + (size_t)(t < tiles_m ? t : 0) * 3 * N + c;
        pv[u] = p[0];
        s1[u] = p[N];
        s2[u] = p[2 * (size_t)N];
      }
#pragma unroll
      for (int u = 0; u < SF_U; ++u) {
        const float n_b = nb[u];
        if (n_b == 0.f) continue;
        fold(n_b, pv[u] + s1[u] / n_b, fmaxf(s2[u] - s1[u] * s1[u] / n_b, 0.f));
      }
    }
  }
  red[tl][0][cl] = n_a;
  red[tl][1][cl] = mu_a;
  red[tl][2][cl] = m2_a;
  __syncthreads();
  // the TL partial results of a channel in lane order, 16 at a time (two levels when TL > 64)
  if (TL > 64) {
    if (tl < TL / 16) {
      n_a = mu_a = m2_a = 0.f;
      for (int w = tl * 16; w < tl * 16 + 16; ++w) fold(red[w][0][cl], red[w][1][cl], red[w][2][cl]);
    }
    __syncthreads();
    if (tl < TL / 16) {
      red[tl][0][cl] = n_a;
      red[tl][1][cl] = mu_a;
      red[tl][2][cl] = m2_a;
    }
    __syncthreads();
  }
  if (tl != 0 || c >= N) return;
  n_a = mu_a = m2_a = 0.f;
  for (int w = 0; w < (TL > 64 ? TL / 16 : TL); ++w) fold(red[w][0][cl], red[w][1][cl], red[w][2][cl]);
  const float var = n_a > 0.f ? m2_a / n_a : 0.f;
  mean[c] = mu_a;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = n_a > 1.f ? m2_a / (n_a - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu_a;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

}  // namespace

#ifdef COIN_LAB
int coin_conv_gemm_no_s4 = 0;       // lab hook: 1 = the round-5 dispatch (no small-map core) for same-box A/B runs of the whole step
extern "C" void coin_lab_set_no_s4(int v) { coin_conv_gemm_no_s4 = v; }   // (tools/ab_bench.py; exported by the lab library only)
int coin_conv_gemm_force_impl = 0;  // lab hook (tools/gemm_lab.hip): 0 = default, 1 = p8, 2 = sq, 3 = rect, 4 = s4 (128 x 128 x 32)
#else
static constexpr int coin_conv_gemm_force_impl = 0, coin_conv_gemm_no_s4 = 0;   // the product library has no implementation switch
#endif

extern "C" size_t coin_conv_gemm_stats_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)((M + 127) / 128) * 3 * (size_t)N * sizeof(float);   // enough for either tile height (128-row tiles: the small-map core)
}

extern "C" size_t coin_conv_gemm_workspace_bytes(int M, int N, int K) {
  const size_t p8 = coin_p8_nt_workspace_bytes(M, N, K), s4 = coin_s4_nt_workspace_bytes(M, N, K);
  return p8 > s4 ? p8 : s4;
}

// Which core serves a launch: 1 = the persistent 256 x 256 x 64 kernel, 4 = the 128 x 128 x 32 small-map kernel, 0 = the round-2 kernels.
// One function for the dispatch and for coin_conv_gemm_stats_tile_rows, so that the caller's reading of the partials cannot disagree
// with the kernel that wrote them.
static int conv_gemm_pick(int lda, int mode, int Cin, int ldb, int M, int N, int K) {
  if (coin_conv_gemm_force_impl) {
    if (coin_conv_gemm_force_impl == 4) return coin_s4_nt_ok(M, N, K, mode, Cin, lda, ldb) ? 4 : 0;
    if (coin_conv_gemm_force_impl == 1) return coin_p8_nt_ok(M, N, K, mode, Cin, lda, ldb) ? 1 : 0;
    return 0;
  }
  if (!coin_conv_gemm_no_s4 && coin_s4_nt_wanted(M, N, K) && coin_s4_nt_ok(M, N, K, mode, Cin, lda, ldb)) return 4;
  if (coin_p8_nt_ok(M, N, K, mode, Cin, lda, ldb)) return 1;
  return 0;
}

extern "C" int coin_conv_gemm_stats_tile_rows(int lda, int mode, int Cin, int ldb, int M, int N, int K) {
  return conv_gemm_pick(lda, mode, Cin, ldb, M, N, K) == 4 ? 128 : 256;
}

extern "C" int coin_conv_gemm_bf16(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc,
                                   const void* R, int ldr, int M, int N, int K, float* stats, int64_t stats_rows, void* stream) {
  return coin_conv_gemm_bf16_ws(A, lda, mode, H, W, Cin, B, ldb, C, ldc, R, ldr, M, N, K, stats, stats_rows, nullptr, 0, stream);
}

// C = bf16(bf16(A.B^T) + 0.25 * R[pooled pixel]): R is the gradient of a 2x2 / stride-2 average pool of the OUTPUT pixel grid [*, out_h, out_w]
// (rows of R = pixels of [*, out_h / 2, out_w / 2], floor).  The data gradient of a Bottleneck's conv1 takes the downsample branch's
// gradient this way: the avg-pool backward (an [M, N] tensor written and read back) disappears into the epilogue's residual add.
// Served by the persistent 8-phase kernel only: COIN_ESHAPE where that kernel does not fit (the caller materialises the pool gradient).
extern "C" int coin_conv_gemm_bf16_rpool(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc,
                                         const void* R, int ldr, int out_h, int out_w, int M, int N, int K, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!A || !B || !C || !R) return COIN_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0 || ldb < K || ldc < N || ldr < N || (mode != 0 && mode != 1)) return COIN_EINVAL;
  if (out_h < 2 || out_w < 2 || M % (out_h * out_w)) return COIN_EINVAL;
  if (K % GK || ldb % 8 || ldc % 8 || N % 8 || ldr % 8) return COIN_ESHAPE;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || ((uintptr_t)R & 15)) return COIN_EALIGN;
  if (mode == 0) {
    if (lda < K) return COIN_EINVAL;
    if (lda % 8) return COIN_ESHAPE;
  } else {
    if (H <= 0 || W <= 0 || Cin <= 0 || M % (H * W)) return COIN_EINVAL;
    if (Cin % GK || K != 9 * Cin) return COIN_ESHAPE;
  }
  const int pick = conv_gemm_pick(lda, mode, Cin, ldb, M, N, K);
  if (pick == 4)
    return coin_s4_nt_launch(A, lda, mode, H, W, Cin, B, ldb, C, ldc, R, ldr, M, N, K, nullptr, 0, ((uintptr_t)workspace & 15) ? nullptr : workspace,
                             workspace_bytes, (hipStream_t)stream, out_h, out_w);
  if (pick != 1) return COIN_ESHAPE;
  return coin_p8_nt_launch(A, lda, mode, H, W, Cin, B, ldb, C, ldc, R, ldr, M, N, K, nullptr, 0, ((uintptr_t)workspace & 15) ? nullptr : workspace,
                           workspace_bytes, (hipStream_t)stream, out_h, out_w);
}

extern "C" int coin_conv_gemm_bf16_ws(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc,
                                      const void* R, int ldr, int M, int N, int K, float* stats, int64_t stats_rows, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (!A || !B || !C) return COIN_EINVAL;
  if (M < 0 || N < 0 || K <= 0 || ldb < K || ldc < N || (mode != 0 && mode != 1) || (R && ldr < N)) return COIN_EINVAL;
  if (M == 0 || N == 0) return COIN_OK;
  if (K % GK || ldb % 8 || ldc % 8 || N % 8 || (R && ldr % 8)) return COIN_ESHAPE;  // GK = 64 is also a multiple of the square tile's K-step
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || ((uintptr_t)R & 15)) return COIN_EALIGN;
  const bf16_t* rm = (const bf16_t*)R;
  if (mode == 0) {
    if (lda < K) return COIN_EINVAL;
    if (lda % 8) return COIN_ESHAPE;
  } else {
    if (H <= 0 || W <= 0 || Cin <= 0 || M % (H * W)) return COIN_EINVAL;
    if (Cin % GK || K != 9 * Cin) return COIN_ESHAPE;
  }
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* b = (const bf16_t*)B;
  bf16_t* c = (bf16_t*)C;
  // the small-map core / the persistent 8-phase core where the shape fits them; lab builds can force "sq" (the 256x256x32 kernel of
  // round 2) / "rect" (256x128x64)
  const int pick = conv_gemm_pick(lda, mode, Cin, ldb, M, N, K);
  if (pick == 4)
    return coin_s4_nt_launch(A, lda, mode, H, W, Cin, B, ldb, C, ldc, R, ldr, M, N, K, stats, (long long)stats_rows,
                             ((uintptr_t)workspace & 15) ? nullptr : workspace, workspace_bytes, st);
  if (pick == 1)
    return coin_p8_nt_launch(A, lda, mode, H, W, Cin, B, ldb, C, ldc, R, ldr, M, N, K, stats, (long long)stats_rows,
                             ((uintptr_t)workspace & 15) ? nullptr : workspace, workspace_bytes, st);
  const int force_rect = coin_conv_gemm_force_impl == 3;
  if (N % QN == 0 && !force_rect) {
    const int tm = (M + QM - 1) / QM, tn = N / QN;
    const size_t lds = (size_t)QNSTAGE * QSTAGE_BYTES;
#define COIN_LAUNCH_Q(G3, ST)                                                                                                  \
  do {                                                                                                                         \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)conv_gemm256_bf16_kernel<G3, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    conv_gemm256_bf16_kernel<G3, ST><<<tm * tn, 512, lds, st>>>(a, lda, b, ldb, c, ldc, M, N, K, H, W, Cin, stats, stats_rows, tm, tn, rm, ldr); \
  } while (0)
    if (mode == 1) {
      if (stats) COIN_LAUNCH_Q(true, true); else COIN_LAUNCH_Q(true, false);
    } else {
      if (stats) COIN_LAUNCH_Q(false, true); else COIN_LAUNCH_Q(false, false);
    }
#undef COIN_LAUNCH_Q
    return coin_launch_status();
  }
  const int tm = (M + GM - 1) / GM, tn = (N + GN - 1) / GN;
  const size_t lds = (size_t)NSTAGE * STAGE_BYTES;
#define COIN_LAUNCH(G3, ST)                                                                                                   \
  do {                                                                                                                        \
    static bool attr_set = false;                                                                                             \
    if (!attr_set) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)conv_gemm_bf16_kernel<G3, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set = true;                                                                                                        \
    }                                                                                                                         \
    conv_gemm_bf16_kernel<G3, ST><<<tm * tn, 512, lds, st>>>(a, lda, b, ldb, c, ldc, M, N, K, H, W, Cin, stats, stats_rows, tm, tn, rm, ldr); \
  } while (0)
  if (mode == 1) {
    if (stats) COIN_LAUNCH(true, true); else COIN_LAUNCH(true, false);
  } else {
    if (stats) COIN_LAUNCH(false, true); else COIN_LAUNCH(false, false);
  }
#undef COIN_LAUNCH
  return coin_launch_status();
}

extern "C" int coin_conv_gemm_stats_finalize(const float* partials, int M, int N, int64_t rows, int tile_rows, float eps, float momentum, float* mean,
                                             float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream) {
  if (!partials || !mean || !rstd || M <= 0 || N <= 0 || rows <= 0 || rows > M || (tile_rows != 128 && tile_rows != 256)) return COIN_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return COIN_EINVAL;
  // 8 channels x 128 tile lanes per workgroup, 8 tiles per round (tools/statsfin_bench.py, us per call incl. the wrapper's ~8 us floor:
  // [401408, 512] 16.3, [100352, 2048] 11.4, [100352, 512] 8.3; 16 channels x 64 lanes: 24.6 / 14.9 / 14.4; 4 x 256: 20.5 / 20.9 / 8.4)
  const int tiles = (M + tile_rows - 1) / tile_rows;
  conv_stats_finalize_kernel<8, 8><<<(N + 7) / 8, 1024, 0, (hipStream_t)stream>>>(partials, tiles, N, rows, eps, momentum, mean, rstd, running_mean,
                                                                                 running_var, num_batches_tracked, tile_rows);
  return coin_launch_status();
}

static int wgrad_slices(int M, int tiles) {
  int s = (768 + tiles - 1) / tiles;  // ~3 blocks per CU
  const int max_s = (M + 4095) / 4096; // at least 4096 pixels per slice
  s = s < max_s ? s : max_s;
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

extern "C" size_t coin_conv_wgrad_workspace_bytes(int M, int Cout, int Ktot) {
  if (M <= 0 || Cout <= 0 || Ktot <= 0 || Cout % 128 || Ktot % 128) return 0;
  const size_t s4 = coin_s4_tn_workspace_bytes(M, Cout, Ktot);
  const size_t p8 = coin_p8_tn_workspace_bytes(M, Cout, Ktot);  // any of the kernels may be selected at launch
  const size_t both = s4 > p8 ? s4 : p8;
  if (Cout % WB || Ktot % WB) return both;   // odd multiples of 128: not the sliced kernel
  const size_t sliced = (size_t)wgrad_slices(M, (Cout / WB) * (Ktot / WB)) * Cout * (size_t)Ktot * sizeof(float);
  return sliced > both ? sliced : both;
}

extern "C" int coin_conv_wgrad_bf16(const void* GY, const void* X, int mode, int H, int W, int Cin, int M, int Cout, int Ktot, float* dW,
                                    void* workspace, void* stream) {
  if (!GY || !X || !dW || !workspace || M <= 0 || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return COIN_EINVAL;
  if (Cout % 128 || Cin % 128) return COIN_ESHAPE;
  if (mode == 0 ? Ktot != Cin : (Ktot != 9 * Cin || H <= 0 || W <= 0 || M % (H * W))) return COIN_EINVAL;
  if (((uintptr_t)GY & 15) || ((uintptr_t)X & 15) || ((uintptr_t)dW & 15) || ((uintptr_t)workspace & 15)) return COIN_EALIGN;
  const bool use_old = coin_conv_gemm_force_impl ? (coin_conv_gemm_force_impl != 1 && coin_conv_gemm_force_impl != 4) : false;   // lab builds only
  if (!use_old && coin_s4_tn_ok(M, Cout, Cin, Ktot, mode) &&
      (coin_conv_gemm_force_impl == 4 || (coin_conv_gemm_force_impl == 0 && !coin_conv_gemm_no_s4 && coin_s4_tn_wanted(M, Cout, Cin))))
    return coin_s4_tn_launch(GY, X, mode, H, W, Cin, M, Cout, Ktot, dW, workspace, (hipStream_t)stream);
  if (!use_old && coin_p8_tn_ok(M, Cout, Cin, Ktot, mode))
    return coin_p8_tn_launch(GY, X, mode, H, W, Cin, M, Cout, Ktot, dW, workspace, (hipStream_t)stream);
  if (Cout % WB || Cin % WB) return COIN_ESHAPE;   // the sliced kernel below needs whole 256 x 256 tiles
  const int tco = Cout / WB, tk = Ktot / WB;
  const int slices = wgrad_slices(M, tco * tk);
  int m_chunk = (M + slices - 1) / slices;
  m_chunk = (m_chunk + WK - 1) / WK * WK;
  const size_t lds = (size_t)WNSTAGE * 2 * WTILE;
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)workspace;
  if (mode == 1) {
    static bool set3 = false;
    if (!set3) { (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set3 = true; }
    conv_wgrad_bf16_kernel<true><<<tco * tk * slices, 512, lds, st>>>((const bf16_t*)GY, (const bf16_t*)X, slab, M, Cout, Cin, Ktot, H, W, tco, tk, m_chunk);
  } else {
    static bool set1 = false;
    if (!set1) { (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set1 = true; }
    conv_wgrad_bf16_kernel<false><<<tco * tk * slices, 512, lds, st>>>((const bf16_t*)GY, (const bf16_t*)X, slab, M, Cout, Cin, Ktot, H, W, tco, tk, m_chunk);
  }
  const long long n = (long long)Cout * Ktot;
  wgrad_reduce_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(slab, dW, n, slices);
  return coin_launch_status();
}
