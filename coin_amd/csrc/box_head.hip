// Box-head kernels for gfx950: MFMA GEMM (bf16 and exact-f32), 2-D transpose, bias/activation
// backward, and the cosine-similarity classifier.
//
// Replaces FastRCNNOutputLayers.forward / do_classify (coin/modeling/roi_heads/fast_rcnn.py:318-353):
// the `trans` MLP (Linear 2048->1024, LeakyReLU, 1024->1024, LeakyReLU, 1024->2048), `cls_score`
// 2048->D, L2-normalised cosine logits against the text embeddings, and `bbox_pred` 2048->4.
//
// GEMM design (bf16): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 MFMA
// 16x16x32 accumulators.  Both operands are K-contiguous ("NT": Y = X.W^T is nn.Linear's layout), so
// a fragment is one 16-byte LDS read.  Tiles are staged with 16-byte global_load_lds (LDS-DMA, no
// VGPR round trip) into a lane-linear image; the XOR swizzle that makes the ds_read_b128 fragment
// reads bank-conflict free is applied on the per-lane SOURCE address and again on the read
// (cdna_hip_programming.md rule 21).  Two LDS buffers: tile t+1 is in flight while tile t is
// multiplied.  The f32 path uses v_mfma_f32_16x16x4_f32 (bit-exact fp32 FMA chain in k order) and
// exists for the 1e-4 fp32 parity runs, not for throughput.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// bf16 NT GEMM
// ------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  // lane i lands at lds_wave_base + 16*i
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage one [128 x 64] bf16 tile (rows row0.., k-columns k0..k0+63) of a row-major matrix with
// leading dimension ld into `lds_tile` (16 KiB).  Rows >= nrows are clamped (their products only
// reach outputs that are never stored).  Image: row r at byte r*128, 16-byte chunk c stored at
// physical chunk c ^ (r & 7).
template <int NW>   // waves of the workgroup (4 or 8): each stages 16 / NW of the tile's 8-row groups
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ mat, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
  const int rl = lane >> 3;                 // row within the 8-row group written by one instruction
  const int c = (lane & 7) ^ rl;            // logical chunk that must land at physical chunk lane&7
#pragma unroll
  for (int j = 0; j < 16 / NW; ++j) {
    const int rgrp = wave * (16 / NW) + j;  // 8-row group index (16 groups per tile)
    int row = row0 + rgrp * 8 + rl;
    row = row < nrows ? row : nrows - 1;
    const bf16_t* src = mat + (size_t)row * ld + k0 + c * 8;
    glds16(src, lds_tile + rgrp * 1024);
  }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* lds_tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// NW = 4: 2 x 2 waves of 64 x 64 (the throughput shape: two workgroups per CU hide each other's waits).  NW = 8: 2 x 4 waves of 64 x 32
// for launches with no more tiles than CUs (the box head's M = 2048 GEMMs): two waves per SIMD, so that one wave's 12 LDS fragment
// reads run under the other's 16 MFMAs -- with one wave per SIMD they ran back to back (~512 + 512 cycles per half K-step).
template <typename OutT, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_nt_bf16_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, OutT* __restrict__ Cmat, int ldc,
    int M, int N, int K, const float* __restrict__ bias, int act, float alpha, int tiles_m, int tiles_n) {
  constexpr int WC = NW / 2;          // waves along N
  constexpr int NJ = 4 / (WC / 2);    // 16-column fragments per wave: 4 (NW = 4) or 2 (NW = 8)
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // [buf][A|B]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware tile order: consecutive tiles along N (sharing the A panel) stay on one XCD.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave / WC, wc = wave % WC;

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = K / BK;
  stage_tile<NW>(A, lda, m0, M, 0, lds, wave, lane);
  stage_tile<NW>(B, ldb, n0, N, 0, lds + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    char* la = lds + cur * 2 * TILE_BYTES;
    char* lb = la + TILE_BYTES;
    if (t + 1 < nt) {
      char* na = lds + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile<NW>(A, lda, m0, M, (t + 1) * BK, na, wave, lane);
      stage_tile<NW>(B, ldb, n0, N, (t + 1) * BK, na + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds_frag(la, wr * 64 + i * 16 + fr, kk * 4 + fq);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = lds_frag(lb, wc * (16 * NJ) + j * 16 + fr, kk * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wc * (16 * NJ) + j * 16 + fr;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + i * 16 + fq * 4 + r;
        if (row >= M) continue;
        float v = acc[i][j][r] + bv;
        if (act == COIN_ACT_LEAKY_RELU) v = v > 0.f ? v : v * alpha;
        else if (act == COIN_ACT_RELU) v = v > 0.f ? v : 0.f;
        Cmat[(size_t)row * ldc + col] = (OutT)v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// f32 NT GEMM on v_mfma_f32_16x16x4_f32 (exact fp32).  64x64x16 tile, 4 waves (2x2), 32x32 per wave.
// ------------------------------------------------------------------------------------------
constexpr int FM = 64, FN = 64, FK = 16, FLD = FK + 1;

template <typename OutT>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, OutT* __restrict__ Cmat, int ldc,
    int M, int N, int K, const float* __restrict__ bias, int act, float alpha) {
  __shared__ float sa[FM * FLD];
  __shared__ float sb[FN * FLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.y * FM, n0 = blockIdx.x * FN;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lr = threadIdx.x >> 2, lc = (threadIdx.x & 3) * 4;  // 64 rows x 4 float4 per tile
  const int fr = lane & 15, fq = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += FK) {
    {
      int ra = m0 + lr; ra = ra < M ? ra : M - 1;
      int rb = n0 + lr; rb = rb < N ? rb : N - 1;
      const f32x4 va = *reinterpret_cast<const f32x4*>(A + (size_t)ra * lda + k0 + lc);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(B + (size_t)rb * ldb + k0 + lc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sa[lr * FLD + lc + i] = va[i];
        sb[lr * FLD + lc + i] = vb[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FK; kk += 4) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = sa[(wr * 32 + i * 16 + fr) * FLD + kk + fq];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = sb[(wc * 32 + j * 16 + fr) * FLD + kk + fq];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 32 + j * 16 + fr;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + i * 16 + fq * 4 + r;
        if (row >= M) continue;
        float v = acc[i][j][r] + bv;
        if (act == COIN_ACT_LEAKY_RELU) v = v > 0.f ? v : v * alpha;
        else if (act == COIN_ACT_RELU) v = v > 0.f ? v : 0.f;
        Cmat[(size_t)row * ldc + col] = (OutT)v;
      }
  }
}

// ------------------------------------------------------------------------------------------
// transpose
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int M, int N) {
  __shared__ T tile[64][65];
  const int bx = blockIdx.x * 64, by = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int row = by + r, col = bx + tx;
    if (row < M && col < N) tile[r][tx] = in[(size_t)row * N + col];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int orow = bx + r, ocol = by + tx;  // out[N][M]
    if (orow < N && ocol < M) out[(size_t)orow * M + ocol] = tile[tx][r];
  }
}

// Data-gradient layout of every convolution / linear weight in ONE launch: dst[ci][ks-1-ky][ks-1-kx][co] = src[co][ky][kx][ci]
// (bf16; src = the channels-last compute shadow of a conv weight, or a linear weight [N, K] with ks = 1, i.e. its transpose).
// The dgrad GEMMs (coin_conv_gemm_bf16 / coin_gemm_nt with B = W^T) read these; before, every backward call re-laid its weight with
// flip + permute + copy launches (46 + 11 per step).  grid = (max tiles over the table, entries); 64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void weight_dgrad_layout_kernel(const coin_wd_tensor* __restrict__ table) {
  __shared__ uint16_t tile[64][72];   // row pitch 144 B: 16-byte aligned rows, column reads spread over the banks
  const coin_wd_tensor t = table[blockIdx.y];
  const int tco = (t.cout + 63) >> 6, tci = (t.cin + 63) >> 6;
  const int per_tap = tco * tci;
  if ((int)blockIdx.x >= per_tap * t.ks * t.ks) return;
  const int tap = blockIdx.x / per_tap, rem = blockIdx.x - tap * per_tap;
  const int co0 = (rem / tci) << 6, ci0 = (rem % tci) << 6;
  const int ftap = t.ks * t.ks - 1 - tap;   // (ks-1-ky) * ks + (ks-1-kx)
  const int taps = t.ks * t.ks;
  // load: thread -> (row = co, 16-byte chunk of 8 ci); cin % 8 == 0
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = it * 256 + threadIdx.x, r = idx >> 3, c = (idx & 7) << 3;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (co0 + r < t.cout && ci0 + c < t.cin) v = *reinterpret_cast<const uint4*>(t.src + ((size_t)(co0 + r) * taps + tap) * t.cin + ci0 + c);
    *reinterpret_cast<uint4*>(&tile[r][c]) = v;
  }
  __syncthreads();
  // store: thread -> (row = ci, 8 consecutive co); cout % 8 == 0
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = it * 256 + threadIdx.x, r = idx >> 3, c = (idx & 7) << 3;
    if (ci0 + r < t.cin && co0 + c < t.cout) {
      uint16_t o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = tile[c + k][r];
      uint4 v;
      v.x = o[0] | ((uint32_t)o[1] << 16);
      v.y = o[2] | ((uint32_t)o[3] << 16);
      v.z = o[4] | ((uint32_t)o[5] << 16);
      v.w = o[6] | ((uint32_t)o[7] << 16);
      *reinterpret_cast<uint4*>(t.dst + ((size_t)(ci0 + r) * taps + ftap) * t.cout + co0 + c) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// bias + activation backward
// ------------------------------------------------------------------------------------------
// Workgroup (bx, by) owns rows [256 by, 256 by + 256) x columns [64 bx, 64 bx + 64): dZ, and the column sums of its rows as ONE
// partial row part[by][col] (no atomics: `bias_sum_kernel` adds the partials to dbias in block order, so the bias gradient is
// bit-reproducible).  V = columns per thread: 16-byte accesses where N, ld and the pointers allow it, else 1.
template <typename T, int V>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* __restrict__ dC, const T* __restrict__ Cm,
                                                           T* __restrict__ dZ, int ld, int M, int N,
                                                           float* __restrict__ part, int act, float alpha) {
  constexpr int CL = 64 / V;          // column lanes
  constexpr int RL = 256 / CL;        // row lanes
  typedef T vec_t __attribute__((ext_vector_type(V)));
  __shared__ float red[RL][64];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int col = blockIdx.x * 64 + cl * V;
  const int r0 = blockIdx.y * 256;
  float s[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s[i] = 0.f;
  if (col < N) {
    for (int r = r0 + rl; r < M && r < r0 + 256; r += RL) {
      const size_t o = (size_t)r * ld + col;
      const vec_t gin = *reinterpret_cast<const vec_t*>(dC + o);
      float gf[V];
#pragma unroll
      for (int i = 0; i < V; ++i) gf[i] = (float)gin[i];
      if (act != COIN_ACT_NONE) {
        const vec_t c = *reinterpret_cast<const vec_t*>(Cm + o);
#pragma unroll
        for (int i = 0; i < V; ++i) gf[i] = (float)c[i] > 0.f ? gf[i] : (act == COIN_ACT_LEAKY_RELU ? gf[i] * alpha : 0.f);
      }
      if (dZ) {
        vec_t g;
#pragma unroll
        for (int i = 0; i < V; ++i) g[i] = (T)gf[i];
        *reinterpret_cast<vec_t*>(dZ + o) = g;
      }
#pragma unroll
      for (int i = 0; i < V; ++i) s[i] += gf[i];   // the fp32 value, as before (not the rounded store)
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) red[rl][cl * V + i] = s[i];
  __syncthreads();
  if (part && threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < N) {
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < RL; ++k) t += red[k][threadIdx.x];
    part[(size_t)blockIdx.y * N + blockIdx.x * 64 + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void bias_sum_kernel(const float* __restrict__ part, int nby, int N, float* __restrict__ dbias) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  float t = 0.f;
  for (int y = 0; y < nby; ++y) t += part[(size_t)y * N + col];
  dbias[col] += t;   // ACCUMULATED (one writer per column)
}

// ------------------------------------------------------------------------------------------
// cosine logits
// ------------------------------------------------------------------------------------------
constexpr int COS_MAX_K = 64;

template <typename T>
__global__ __launch_bounds__(256) void cosine_fwd_kernel(const T* __restrict__ feats, int ldf,
                                                         const float* __restrict__ text, int R, int D, int Kc,
                                                         float inv_scale, float* __restrict__ scores,
                                                         float* __restrict__ inv_norm_f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const T* __restrict__ f = feats + (size_t)r * ldf;
  float ff = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = (float)f[d];
    ff += v * v;
  }
  ff = wave_reduce_sum(ff);
  const float inf = 1.0f / sqrtf(ff);
  if (lane == 0 && inv_norm_f) inv_norm_f[r] = inf;
  for (int k = 0; k < Kc; ++k) {
    const float* __restrict__ t = text + (size_t)k * D;
    float dot = 0.f, tt = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float tv = t[d];
      dot += (float)f[d] * tv;
      tt += tv * tv;
    }
    dot = wave_reduce_sum(dot);
    tt = wave_reduce_sum(tt);
    if (lane == 0) scores[(size_t)r * Kc + k] = dot * inf / sqrtf(tt) * inv_scale;
  }
}

// One block = 16 rows; thread t owns feature columns d = t, t+256, ...
template <typename T>
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ ds, const T* __restrict__ feats,
                                                         int ldf, const float* __restrict__ text,
                                                         const float* __restrict__ scores,
                                                         const float* __restrict__ inv_norm_f, int R, int D, int Kc,
                                                         float inv_scale, T* __restrict__ d_feats,
                                                         float* __restrict__ d_text /* this block's partial [Kc][D] */) {
  __shared__ float s_inv_t[COS_MAX_K];
  __shared__ float s_ds[16][COS_MAX_K];
  __shared__ float s_sc[16][COS_MAX_K];
  __shared__ float s_dot[16];  // sum_k ds*s per row
  __shared__ float red[16];
  const int r0 = blockIdx.x * 16;
  const int nr = (R - r0) < 16 ? (R - r0) : 16;
  // text inverse norms
  for (int k = 0; k < Kc; ++k) {
    float tt = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
      const float v = text[(size_t)k * D + d];
      tt += v * v;
    }
    tt = block_reduce_sum(tt, red);
    if (threadIdx.x == 0) s_inv_t[k] = 1.0f / sqrtf(tt);
  }
  for (int i = threadIdx.x; i < 16 * Kc; i += 256) {
    const int rr = i / Kc, k = i - rr * Kc;
    const bool ok = rr < nr;
    s_ds[rr][k] = ok ? ds[(size_t)(r0 + rr) * Kc + k] : 0.f;
    s_sc[rr][k] = ok ? scores[(size_t)(r0 + rr) * Kc + k] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float a = 0.f;
    for (int k = 0; k < Kc; ++k) a += s_ds[threadIdx.x][k] * s_sc[threadIdx.x][k];
    s_dot[threadIdx.x] = a;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float tn[COS_MAX_K > 16 ? 16 : COS_MAX_K];  // processed in groups of 16 classes to bound registers
    for (int kb = 0; kb < Kc; kb += 16) {
      const int kn = (Kc - kb) < 16 ? (Kc - kb) : 16;
      float dt[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        tn[k] = k < kn ? text[(size_t)(kb + k) * D + d] * s_inv_t[kb + k] : 0.f;
        dt[k] = 0.f;
      }
      for (int rr = 0; rr < nr; ++rr) {
        const float fn = (float)feats[(size_t)(r0 + rr) * ldf + d] * inv_norm_f[r0 + rr];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k < kn) dt[k] += s_ds[rr][kb + k] * (inv_scale * fn - tn[k] * s_sc[rr][kb + k]);
        }
      }
      if (d_text) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < kn) d_text[((size_t)blockIdx.x * Kc + kb + k) * D + d] = dt[k] * s_inv_t[kb + k];   // joined in block order by cosine_dtext_sum_kernel
      }
    }
    if (d_feats) {
      for (int rr = 0; rr < nr; ++rr) {
        const float inf = inv_norm_f[r0 + rr];
        const float fn = (float)feats[(size_t)(r0 + rr) * ldf + d] * inf;
        float a = 0.f;
        for (int k = 0; k < Kc; ++k) a += s_ds[rr][k] * text[(size_t)k * D + d] * s_inv_t[k];
        d_feats[(size_t)(r0 + rr) * ldf + d] = (T)(inf * (inv_scale * a - fn * s_dot[rr]));
      }
    }
  }
}

// d_text[k][d] += sum over the row blocks' partials, in block order (no float atomics: bit-reproducible)
__global__ __launch_bounds__(256) void cosine_dtext_sum_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ d_text) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int b = 0; b < nblk; ++b) t += part[(size_t)b * n + i];
  d_text[i] += t;
}

}  // namespace

extern "C" int coin_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                            const float* bias, int act, float act_alpha, int dtype, int out_dtype, void* stream) {
  if (!A || !B || !C) return COIN_EINVAL;
  if (M < 0 || N < 0 || K <= 0 || lda < K || ldb < K || ldc < N) return COIN_EINVAL;
  if (act < COIN_ACT_NONE || act > COIN_ACT_RELU) return COIN_EINVAL;
  if ((out_dtype != COIN_F32 && out_dtype != COIN_BF16) || (dtype != COIN_F32 && dtype != COIN_BF16)) return COIN_EINVAL;
  if (M == 0 || N == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == COIN_BF16) {
    if (K % BK || lda % 8 || ldb % 8) return COIN_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return COIN_EALIGN;
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    static int ncu = 0;
    if (!ncu) {
      int dev = 0, v = 0;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
#define GO(OT, NW) gemm_nt_bf16_kernel<OT, NW><<<tm * tn, NW * 64, 0, st>>>((const bf16_t*)A, lda, (const bf16_t*)B, ldb, (OT*)C, ldc, M, N, K, bias, act, act_alpha, tm, tn)
    const bool small = tm * tn <= ncu;
    if (out_dtype == COIN_BF16) {
      if (small) GO(bf16_t, 8); else GO(bf16_t, 4);
    } else {
      if (small) GO(float, 8); else GO(float, 4);
    }
#undef GO
  } else {
    if (K % FK || lda % 4 || ldb % 4) return COIN_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return COIN_EALIGN;
    dim3 grid((N + FN - 1) / FN, (M + FM - 1) / FM);
    if (out_dtype == COIN_BF16)
      gemm_nt_f32_kernel<bf16_t><<<grid, 256, 0, st>>>((const float*)A, lda, (const float*)B, ldb, (bf16_t*)C, ldc, M,
                                                        N, K, bias, act, act_alpha);
    else
      gemm_nt_f32_kernel<float><<<grid, 256, 0, st>>>((const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, M, N,
                                                       K, bias, act, act_alpha);
  }
  return coin_launch_status();
}

extern "C" int coin_transpose2d(const void* in, void* out, int M, int N, int dtype, void* stream) {
  if (!in || !out || M < 0 || N < 0) return COIN_EINVAL;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  if (M == 0 || N == 0) return COIN_OK;
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == COIN_F32)
    transpose_kernel<float><<<grid, 256, 0, st>>>((const float*)in, (float*)out, M, N);
  else
    transpose_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, M, N);
  return coin_launch_status();
}

extern "C" int coin_weight_dgrad_layout(const coin_wd_tensor* table, int num_tensors, int max_tiles, void* stream) {
  if (num_tensors < 0 || max_tiles < 0 || (num_tensors > 0 && !table)) return COIN_EINVAL;
  if (num_tensors == 0 || max_tiles == 0) return COIN_OK;
  if (num_tensors > 65535) return COIN_ESHAPE;
  weight_dgrad_layout_kernel<<<dim3(max_tiles, num_tensors), 256, 0, (hipStream_t)stream>>>(table);
  return coin_launch_status();
}

extern "C" int coin_bias_act_bwd(const void* dC, const void* C, void* dZ, int ld, int M, int N, float* dbias, int act,
                                 float act_alpha, int dtype, void* workspace, void* stream) {
  if (!dC || M < 0 || N < 0 || ld < N) return COIN_EINVAL;
  if (act != COIN_ACT_NONE && !C) return COIN_EINVAL;
  if (act < COIN_ACT_NONE || act > COIN_ACT_RELU) return COIN_EINVAL;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  if (dbias && !workspace) return COIN_EINVAL;   // ceil(M / 256) * N floats of partial column sums
  if (M == 0 || N == 0) return COIN_OK;
  dim3 grid((N + 63) / 64, (M + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  float* part = dbias ? (float*)workspace : nullptr;
  const int v = dtype == COIN_F32 ? 4 : 8;
  const bool wide = N % v == 0 && ld % v == 0 && !(((uintptr_t)dC | (uintptr_t)C | (uintptr_t)dZ) & 15);
#define GO(T, V) bias_act_bwd_kernel<T, V><<<grid, 256, 0, st>>>((const T*)dC, (const T*)C, (T*)dZ, ld, M, N, part, act, act_alpha)
  if (dtype == COIN_F32) {
    if (wide) GO(float, 4); else GO(float, 1);
  } else {
    if (wide) GO(bf16_t, 8); else GO(bf16_t, 1);
  }
#undef GO
  if (dbias) bias_sum_kernel<<<(N + 255) / 256, 256, 0, st>>>(part, (int)grid.y, N, dbias);
  return coin_launch_status();
}

extern "C" int coin_cosine_logits_fwd(const void* feats, int ldf, const float* text, int R, int D, int Kc,
                                      float inv_scale, float* scores, float* inv_norm_f, int dtype, void* stream) {
  if (!feats || !text || !scores || R < 0 || D <= 0 || Kc <= 0 || ldf < D) return COIN_EINVAL;
  if (Kc > COS_MAX_K) return COIN_ESHAPE;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  if (R == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (R + 3) / 4;
  if (dtype == COIN_F32)
    cosine_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)feats, ldf, text, R, D, Kc, inv_scale, scores, inv_norm_f);
  else
    cosine_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)feats, ldf, text, R, D, Kc, inv_scale, scores,
                                                     inv_norm_f);
  return coin_launch_status();
}

extern "C" int coin_cosine_logits_bwd(const float* d_scores, const void* feats, int ldf, const float* text,
                                      const float* scores, const float* inv_norm_f, int R, int D, int Kc,
                                      float inv_scale, void* d_feats, float* d_text, int dtype, void* workspace, void* stream) {
  if (!d_scores || !feats || !text || !scores || !inv_norm_f) return COIN_EINVAL;
  if (d_text && !workspace) return COIN_EINVAL;   // ceil(R / 16) * Kc * D floats: the row blocks' partial text gradients
  if (R < 0 || D <= 0 || Kc <= 0 || ldf < D) return COIN_EINVAL;
  if (Kc > COS_MAX_K) return COIN_ESHAPE;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  if (R == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (R + 15) / 16;
  if (dtype == COIN_F32)
    cosine_bwd_kernel<float><<<grid, 256, 0, st>>>(d_scores, (const float*)feats, ldf, text, scores, inv_norm_f, R, D, Kc,
                                                    inv_scale, (float*)d_feats, d_text ? (float*)workspace : nullptr);
  else
    cosine_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(d_scores, (const bf16_t*)feats, ldf, text, scores, inv_norm_f, R, D,
                                                     Kc, inv_scale, (bf16_t*)d_feats, d_text ? (float*)workspace : nullptr);
  if (d_text) cosine_dtext_sum_kernel<<<(Kc * D + 255) / 256, 256, 0, st>>>((const float*)workspace, grid, Kc * D, d_text);
  return coin_launch_status();
}
