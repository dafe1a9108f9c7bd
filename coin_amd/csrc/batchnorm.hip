// Fused train-mode BatchNorm (+ residual add) (+ ReLU) (+ 2x2 average pool) for channels-last activations, gfx950.
//
// Replaces the nn.BatchNorm2d -> (+= identity) -> ReLU -> nn.AvgPool2d chains of the CLIP Bottleneck
// (coin/modeling/utils.py:77-90) for the trainable stages (layer2, layer3 and res5 = layer4 on the RoI tiles,
// clip_roi_heads.py:172-176).  On the RoI head these chains stream ~3 GB of bf16 activations per step; done as
// separate library launches (statistics, normalise, add, relu, pool and their five backward kernels) they cost more
// device time than the res5 convolutions.  Here:
//   forward  = 1 statistics pass (read x) + 1 apply pass (read x [, residual], write y; y already pooled if pool=2)
//   backward = 1 reduction pass (read x, dy [, y]) + 1 dx pass (read x, dy [, y], write dx [, d_residual]);
//              y is read only when the forward added a residual, otherwise the ReLU mask is recomputed from x
// All four are pure HBM streams: lane = 8 (bf16) or 4 (f32) consecutive channels = one 16-byte access, rows are
// grid-strided, per-channel sums are kept in registers, combined through LDS and flushed with one float atomic per
// (block, channel).  Statistics are accumulated in fp32 around a per-channel pivot (the first row) so that
// E[x^2]-E[x]^2 does not cancel; running_mean / running_var follow nn.BatchNorm2d (momentum, unbiased variance).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;

template <typename T>
struct Ld {
  static constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type vec;
  static __device__ __forceinline__ void load(const T* p, float (&o)[V]) {
    const vec v = *reinterpret_cast<const vec*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = (float)v[i];
  }
  static __device__ __forceinline__ void store(T* p, const float (&o)[V]) {
    vec v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = (T)o[i];
    *reinterpret_cast<vec*>(p) = v;
  }
};

// Thread -> (channel group, row slot).  tpr = threads per row = C / V (<= 256 here; larger C loops over groups).
struct Map {
  int cg, slot, rows_per_iter, ngroups_iter;
};

// ------------------------------------------------------------------------------------------ statistics
// sums[0..C) = sum(x - pivot), sums[C..2C) = sum((x - pivot)^2), pivot[c] = x[0][c]
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ x, int64_t M, int C,
                                                               float* __restrict__ sums) {
  constexpr int V = Ld<T>::V;
  extern __shared__ float red[];  // [BN_THREADS][2*V]
  const int ncg = C / V;
  const int tpr = ncg < BN_THREADS ? ncg : BN_THREADS;
  const int rpi = BN_THREADS / tpr;
  const int slot = threadIdx.x / tpr, cg0 = threadIdx.x - slot * tpr;
  const bool active = slot < rpi;
  for (int cg = cg0; cg < ncg; cg += tpr) {
    float s1[V], s2[V], piv[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    Ld<T>::load(x + (size_t)cg * V, piv);
    if (active) {
#pragma unroll 4
      for (int64_t r = (int64_t)blockIdx.x * rpi + slot; r < M; r += (int64_t)gridDim.x * rpi) {
        float v[V];
        Ld<T>::load(x + (size_t)r * C + (size_t)cg * V, v);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float d = v[i] - piv[i];
          s1[i] += d;
          s2[i] += d * d;
        }
      }
    }
    // combine the row slots that share this channel group
#pragma unroll
    for (int i = 0; i < V; ++i) {
      red[threadIdx.x * 2 * V + i] = s1[i];
      red[threadIdx.x * 2 * V + V + i] = s2[i];
    }
    __syncthreads();
    if (slot == 0) {
      for (int s = 1; s < rpi; ++s) {
        const float* o = red + (size_t)(s * tpr + cg0) * 2 * V;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          s1[i] += o[i];
          s2[i] += o[V + i];
        }
      }
      float* __restrict__ part = sums + (size_t)blockIdx.x * 2 * C;  // per-block partials: no atomics, fixed order
#pragma unroll
      for (int i = 0; i < V; ++i) {
        part[cg * V + i] = s1[i];
        part[C + cg * V + i] = s2[i];
      }
    }
    __syncthreads();
  }
}

// mean / rstd from the per-block pivoted partial sums (+ running statistics).  Block = 64 channels x 16 waves: waves 0-7 sum
// the first-moment partials p = w, w+8, ..., waves 8-15 the second-moment ones, each with coalesced 256-byte reads; the 8 + 8
// wave results are combined through LDS in a fixed order (bit-reproducible).
template <typename T>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const T* __restrict__ x, const float* __restrict__ sums, int nparts, int64_t M,
                                                            int C, float eps, float momentum, float* __restrict__ mean,
                                                            float* __restrict__ rstd, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, int64_t* __restrict__ nbt) {
  __shared__ float red[16][64];
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int which = wave >> 3, w8 = wave & 7;
  float a = 0.f;
  if (c < C) {
    const float* __restrict__ src = sums + (size_t)which * C + c;
#pragma unroll 4
    for (int p = w8; p < nparts; p += 8) a += src[(size_t)p * 2 * C];
  }
  red[wave][lane] = a;
  __syncthreads();
  if (wave != 0 || c >= C) return;
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a1 += red[k][lane];
    a2 += red[8 + k][lane];
  }
  const float piv = (float)x[c];
  const float m1 = a1 / (float)M;
  const float var = fmaxf(a2 / (float)M - m1 * m1, 0.f);
  const float mu = piv + m1;
  mean[c] = mu;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// ------------------------------------------------------------------------------------------ apply (forward)
// Streaming kernels below share one thread mapping: the launch has gridDim.x*BN_THREADS = a multiple of ncg = C/V threads,
// so a thread keeps ONE 16-byte channel group for its whole life (per-channel constants are computed once, in registers)
// and walks rows with a constant stride: no integer division and no per-channel table reads inside the loop.
struct RowWalk {
  int cg;
  int64_t row0, rstep;
};
__device__ __forceinline__ RowWalk row_walk(int ncg) {
  const int64_t tid = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * BN_THREADS;
  RowWalk r;
  r.cg = (int)(tid % ncg);
  r.row0 = tid / ncg;
  r.rstep = nthreads / ncg;
  return r;
}

// x: [N, H, W, C]; y: [N, H/POOL, W/POOL, C] (floor); residual (POOL == 1 only): same shape as y
template <typename T, int POOL>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const T* __restrict__ residual,
                                                               T* __restrict__ y, int N, int H, int W, int C, int relu,
                                                               uint8_t* __restrict__ mask) {
  constexpr int V = Ld<T>::V;
  const int ncg = C / V;
  const RowWalk rw = row_walk(ncg);
  const int OH = H / POOL, OW = W / POOL;
  const int64_t rows = (int64_t)N * OH * OW;
  float sc[V], sh[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = rw.cg * V + i;
    sc[i] = rstd[c] * gamma[c];
    sh[i] = beta[c] - mean[c] * sc[i];
  }
  const size_t coff = (size_t)rw.cg * V;
  if (POOL == 1) {
    if (residual) {
#pragma unroll 2
      for (int64_t r = rw.row0; r < rows; r += rw.rstep) {
        float v[V], rr[V], o[V];
        Ld<T>::load(x + (size_t)r * C + coff, v);
        Ld<T>::load(residual + (size_t)r * C + coff, rr);
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          o[i] = v[i] * sc[i] + sh[i] + rr[i];
          m |= (o[i] > 0.f ? 1u : 0u) << i;
          if (relu) o[i] = fmaxf(o[i], 0.f);
        }
        Ld<T>::store(y + (size_t)r * C + coff, o);
        if (mask) mask[(size_t)r * ncg + rw.cg] = (uint8_t)m;   // bit i: channel i of this 16-byte group passed the ReLU
      }
    } else {
#pragma unroll 4
      for (int64_t r = rw.row0; r < rows; r += rw.rstep) {
        float v[V], o[V];
        Ld<T>::load(x + (size_t)r * C + coff, v);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          o[i] = v[i] * sc[i] + sh[i];
          if (relu) o[i] = fmaxf(o[i], 0.f);
        }
        Ld<T>::store(y + (size_t)r * C + coff, o);
      }
    }
  } else {
    for (int64_t r = rw.row0; r < rows; r += rw.rstep) {
      const int r32 = (int)r;  // rows < 2^31 (checked on the host)
      const int ow = r32 % OW;
      const int t = r32 / OW;
      const int oh = t % OH, n = t / OH;
      float o[V];
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float v[V];
          Ld<T>::load(x + (((size_t)n * H + (2 * oh + dy)) * W + (2 * ow + dx)) * C + coff, v);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            float tt = v[i] * sc[i] + sh[i];
            if (relu) tt = fmaxf(tt, 0.f);
            // the un-pooled activation is rounded to the storage type before pooling, as nn.AvgPool2d sees it
            o[i] += (float)(T)tt;
          }
        }
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] *= 0.25f;
      Ld<T>::store(y + (size_t)r * C + coff, o);
    }
  }
}

// pool == 0: y[n][c] = mean over the H*W positions of relu?( x*sc + sh [+ residual] )  -- the RoI head's spatial mean
// (clip_roi_heads.py:207-208 `x.mean(dim=[2,3])`) folded into the last BatchNorm of res5: the [N,H,W,C] activation is
// never written.  One thread owns one (n, 16-byte channel group) and walks the H*W rows: coalesced, no cross-lane reduction.
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_mean_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const T* __restrict__ residual,
                                                                    T* __restrict__ y, int N, int HW, int C, int relu,
                                                                    uint8_t* __restrict__ mask) {
  constexpr int V = Ld<T>::V;
  const int ncg = C / V;
  const int cg = blockIdx.x * BN_THREADS + threadIdx.x;
  const int n = blockIdx.y;
  if (cg >= ncg) return;
  float sc[V], sh[V], acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = cg * V + i;
    sc[i] = rstd[c] * gamma[c];
    sh[i] = beta[c] - mean[c] * sc[i];
    acc[i] = 0.f;
  }
  const size_t base = (size_t)n * HW * C + (size_t)cg * V;
#pragma unroll 7
  for (int r = 0; r < HW; ++r) {
    float v[V];
    Ld<T>::load(x + base + (size_t)r * C, v);
    if (residual) {
      float rr[V];
      Ld<T>::load(residual + base + (size_t)r * C, rr);
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = v[i] * sc[i] + sh[i] + rr[i];
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = v[i] * sc[i] + sh[i];
    }
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      m |= (v[i] > 0.f ? 1u : 0u) << i;
      acc[i] += relu ? fmaxf(v[i], 0.f) : v[i];
    }
    if (mask) mask[((size_t)n * HW + r) * ncg + cg] = (uint8_t)m;
  }
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] *= inv;
  Ld<T>::store(y + (size_t)n * C + (size_t)cg * V, acc);
}

// Upstream gradient g of the pre-activation at row r = (n,h,w) of x, channel group cg (through pool / ReLU mask).
//   POOL 1: dy has x's shape.  The mask comes from the saved output `yout` when the forward added a residual, otherwise it
//           is recomputed from x (same expression as bn_apply_kernel) and y is never read.
//   POOL 2: dy is [N,H/2,W/2,C]; mask recomputed from x.
//   POOL 0: dy is [N,C] (global mean); `yout` carries the forward's RESIDUAL input (or NULL); mask recomputed.
// MODE (compile time, so that the row loops are branch-free and the loads of several rows are issued together): bit 0 = ReLU,
// bit 1 = `yout` is present (POOL 1: the saved output supplies the mask; POOL 0: the forward's residual input),
// bit 2 = the forward's ReLU bit mask is present (one byte per 16-byte channel group and row; replaces `yout` in both roles: a
// residual block's backward then reads 1/16 of the bytes it read from the saved output, twice).
template <typename T, int POOL, int MODE>
__device__ __forceinline__ void upstream(const T* __restrict__ dy, const T* __restrict__ yout_, const float (&xv)[Ld<T>::V],
                                         const float (&sc)[Ld<T>::V], const float (&sh)[Ld<T>::V], int64_t r, int H, int W, int C,
                                         size_t coff, float (&g)[Ld<T>::V], const uint8_t* __restrict__ mask = nullptr) {
  constexpr int V = Ld<T>::V;
  constexpr bool relu = (MODE & 1) != 0;
  constexpr bool HAS_Y = (MODE & 2) != 0;
  constexpr bool HAS_M = (MODE & 4) != 0;
  const T* __restrict__ yout = yout_;
  if (HAS_M && (POOL == 1 || POOL == 0)) {
    const unsigned m = mask[(size_t)r * (C / V) + coff / V];
    if (POOL == 1) {
      Ld<T>::load(dy + (size_t)r * C + coff, g);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = (!relu || ((m >> i) & 1u)) ? g[i] : 0.f;
    } else {
      const int hw = H * W;
      Ld<T>::load(dy + (size_t)(r / hw) * C + coff, g);
      const float inv = 1.0f / (float)hw;
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = (!relu || ((m >> i) & 1u)) ? g[i] * inv : 0.f;
    }
    return;
  }
  if (POOL == 1) {
    const size_t off = (size_t)r * C + coff;
    Ld<T>::load(dy + off, g);
    if (relu && HAS_Y) {
      float yv[V];
      Ld<T>::load(yout + off, yv);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
    } else if (relu) {
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = (xv[i] * sc[i] + sh[i] > 0.f) ? g[i] : 0.f;
    }
  } else if (POOL == 0) {
    const int hw = H * W;
    const int n = (int)(r / hw);
    Ld<T>::load(dy + (size_t)n * C + coff, g);
    const float inv = 1.0f / (float)hw;
    float rr[V];
    if (HAS_Y) Ld<T>::load(yout + (size_t)r * C + coff, rr);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float pre = xv[i] * sc[i] + sh[i] + (HAS_Y ? rr[i] : 0.f);
      g[i] = (relu && !(pre > 0.f)) ? 0.f : g[i] * inv;
    }
  } else {
    const int r32 = (int)r;
    const int w = r32 % W;
    const int t = r32 / W;
    const int h = t % H, n = t / H;
    const int OH = H / 2, OW = W / 2;
    if ((h >> 1) >= OH || (w >> 1) >= OW) {
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = 0.f;
      return;
    }
    Ld<T>::load(dy + (((size_t)n * OH + (h >> 1)) * OW + (w >> 1)) * C + coff, g);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      g[i] *= 0.25f;
      if (relu && !(xv[i] * sc[i] + sh[i] > 0.f)) g[i] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------ backward reductions
// dsums[0..C) = sum g (= dbeta), dsums[C..2C) = sum g * xhat (= dgamma)
template <typename T, int POOL, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    const T* __restrict__ yout, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int N, int H, int W, int C,
                                                                    int relu, float* __restrict__ dsums, const uint8_t* __restrict__ mask) {
  constexpr int V = Ld<T>::V;
  extern __shared__ float red[];
  const int ncg = C / V;
  const int tpr = ncg < BN_THREADS ? ncg : BN_THREADS;
  const int rpi = BN_THREADS / tpr;
  const int slot = threadIdx.x / tpr, cg0 = threadIdx.x - slot * tpr;
  const bool active = slot < rpi;
  const int64_t M = (int64_t)N * H * W;
  for (int cg = cg0; cg < ncg; cg += tpr) {
    float sc[V], sh[V], mu[V], rs[V], db[V], dg[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = cg * V + i;
      mu[i] = mean[c];
      rs[i] = rstd[c];
      sc[i] = rs[i] * gamma[c];
      sh[i] = beta[c] - mu[i] * sc[i];
      db[i] = dg[i] = 0.f;
    }
    const size_t coff = (size_t)cg * V;
    if (active) {
#pragma unroll 4
      for (int64_t r = (int64_t)blockIdx.x * rpi + slot; r < M; r += (int64_t)gridDim.x * rpi) {
        float xv[V], g[V];
        Ld<T>::load(x + (size_t)r * C + coff, xv);
        upstream<T, POOL, MODE>(dy, yout, xv, sc, sh, r, H, W, C, coff, g, mask);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          db[i] += g[i];
          dg[i] += g[i] * (xv[i] - mu[i]) * rs[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      red[threadIdx.x * 2 * V + i] = db[i];
      red[threadIdx.x * 2 * V + V + i] = dg[i];
    }
    __syncthreads();
    if (slot == 0) {
      for (int s = 1; s < rpi; ++s) {
        const float* o = red + (size_t)(s * tpr + cg0) * 2 * V;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          db[i] += o[i];
          dg[i] += o[V + i];
        }
      }
      float* __restrict__ part = dsums + (size_t)(blockIdx.x + 1) * 2 * C;  // slot 0 holds the final sums
#pragma unroll
      for (int i = 0; i < V; ++i) {
        part[cg * V + i] = db[i];
        part[C + cg * V + i] = dg[i];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(float* __restrict__ dsums, int nparts, int C) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float a = 0.f;
  if (i < 2 * C) {
#pragma unroll 4
    for (int p = 1 + wave; p <= nparts; p += 16) a += dsums[(size_t)p * 2 * C + i];
  }
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && i < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    dsums[i] = t;
  }
}

// dx = gamma * rstd * (g - dbeta/M - xhat * dgamma/M);  d_residual = g (forward had a residual)
template <typename T, int POOL, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                const T* __restrict__ yout, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ dsums,
                                                                int N, int H, int W, int C, int relu, T* __restrict__ dx,
                                                                T* __restrict__ dres, const uint8_t* __restrict__ mask) {
  constexpr int V = Ld<T>::V;
  const int ncg = C / V;
  const RowWalk rw = row_walk(ncg);
  const int64_t M = (int64_t)N * H * W;
  const float invM = 1.0f / (float)M;
  float sc[V], sh[V], mu[V], rs[V], kb[V], kg[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = rw.cg * V + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    sc[i] = rs[i] * gamma[c];
    sh[i] = beta[c] - mu[i] * sc[i];
    kb[i] = dsums[c] * invM;
    kg[i] = dsums[C + c] * invM;
  }
  const size_t coff = (size_t)rw.cg * V;
#pragma unroll 4
  for (int64_t r = rw.row0; r < M; r += rw.rstep) {
    float xv[V], g[V], o[V];
    Ld<T>::load(x + (size_t)r * C + coff, xv);
    upstream<T, POOL, MODE>(dy, yout, xv, sc, sh, r, H, W, C, coff, g, mask);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xhat = (xv[i] - mu[i]) * rs[i];
      o[i] = sc[i] * (g[i] - kb[i] - xhat * kg[i]);
    }
    Ld<T>::store(dx + (size_t)r * C + coff, o);
    if (dres) Ld<T>::store(dres + (size_t)r * C + coff, g);
  }
}

// ------------------------------------------------------------------------------------------ plain 2x2 average pool
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void avgpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                                   int C) {
  constexpr int V = Ld<T>::V;
  const int ncg = C / V, OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)N * OH * OW * ncg;
  for (int64_t idx = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BN_THREADS) {
    const int cg = (int)(idx % ncg);
    const int64_t orow = idx / ncg;
    const int ow = (int)(orow % OW), oh = (int)((orow / OW) % OH), n = (int)(orow / ((int64_t)OW * OH));
    float o[V];
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float v[V];
        Ld<T>::load(x + (((size_t)n * H + (2 * oh + dy)) * W + (2 * ow + dx)) * C + (size_t)cg * V, v);
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] += v[i];
      }
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] *= 0.25f;
    Ld<T>::store(y + (size_t)orow * C + (size_t)cg * V, o);
  }
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void avgpool2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                                                   int C) {
  constexpr int V = Ld<T>::V;
  const int ncg = C / V, OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)N * H * W * ncg;
  for (int64_t idx = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BN_THREADS) {
    const int cg = (int)(idx % ncg);
    const int64_t r = idx / ncg;
    const int w = (int)(r % W), h = (int)((r / W) % H), n = (int)(r / ((int64_t)W * H));
    float g[V];
    if ((h >> 1) < OH && (w >> 1) < OW) {
      Ld<T>::load(dy + (((size_t)n * OH + (h >> 1)) * OW + (w >> 1)) * C + (size_t)cg * V, g);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] *= 0.25f;
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = 0.f;
    }
    Ld<T>::store(dx + (size_t)r * C + (size_t)cg * V, g);
  }
}

int bn_check(const void* x, int N, int H, int W, int C, int pool, int dtype) {
  if (!x || N <= 0 || H <= 0 || W <= 0 || C <= 0) return COIN_EINVAL;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  if (pool != 0 && pool != 1 && pool != 2) return COIN_EINVAL;
  const int v = dtype == COIN_F32 ? 4 : 8;
  if (C % v) return COIN_ESHAPE;
  if (C / v > BN_THREADS && (C / v) % BN_THREADS) return COIN_ESHAPE;  // uniform trip count of the channel-group loops
  if (C / v < BN_THREADS && BN_THREADS % (C / v)) return COIN_ESHAPE;   // a thread keeps one channel group (row-walk kernels)
  if ((int64_t)N * H * W >= (int64_t)1 << 31) return COIN_ESHAPE;
  if ((uintptr_t)x & 15) return COIN_EALIGN;
  return COIN_OK;
}

int stream_grid(int64_t work_items) {
  int64_t g = (work_items + BN_THREADS - 1) / BN_THREADS;
  if (g > 256 * 16) g = 256 * 16;
  return (int)(g < 1 ? 1 : g);
}

// Grid for the row-walk kernels: gridDim.x * BN_THREADS must be a multiple of ncg (bn_check guarantees ncg <= BN_THREADS and
// a power-of-two-free divisor relation, or ncg = k * BN_THREADS), and enough workgroups to fill 256 CUs several times over.
#ifdef COIN_LAB
int g_bn_grid_cap = 0;   // lab hook (tools/bnbench_small.py): workgroups of the row-walk kernels
int g_bn_parts_cap = 0;  // lab hook: workgroups (= partial sums) of the backward's reduction pass
#else
constexpr int g_bn_grid_cap = 0, g_bn_parts_cap = 0;
#endif

int walk_grid(int64_t rows, int ncg) {
  const int unit = ncg <= BN_THREADS ? 1 : ncg / BN_THREADS;  // blocks per full row of channel groups
  int64_t g = (rows * ncg + BN_THREADS - 1) / BN_THREADS;
  // Workgroups of the row-walk kernels.  Round 6 (tools/bnbench_small.py, lab hook): the res5 tensors (>= 16 M channel groups: 411 MB) stream
  // 12-19 % faster in the apply pass and 4-8 % faster in the backward's dx pass with 24 576 workgroups than with 4 096 ([2048,14,14,512]:
  // apply 178 -> 145 us, backward 420 -> 364 us together with 768 reduction workgroups; torch's own elementwise kernels, which launch one
  // small workgroup per 16 KiB, move the same bytes at 5.9-6.0 TB/s) -- 4 096 equal workgroups are 2.7 rounds of the 1 536 the chip holds at
  // this kernel's occupancy, and the last partial round idles a third of it.  Tensors below that size are faster with the coarser grid
  // ([4,100,167,512]: 32.6 vs 35.9 us): their launch is ramp-bound, not tail-bound.
  const int cap = g_bn_grid_cap > 0 ? g_bn_grid_cap : (rows * ncg >= (16LL << 20) ? 24576 : 256 * 16);
  if (g > cap) g = cap;
  g = (g + unit - 1) / unit * unit;
  return (int)(g < unit ? unit : g);
}

}  // namespace

#ifdef COIN_LAB
extern "C" void coin_lab_set_bn_grid(int v) { g_bn_grid_cap = v; }
extern "C" void coin_lab_set_bn_parts(int v) { g_bn_parts_cap = v; }
extern "C" void coin_lab_set_bn_old(int v) { g_bn_grid_cap = v ? 4096 : 0; g_bn_parts_cap = v ? 512 : 0; }   // tools/ab_bench.py bn_old: round 5's launch shapes
extern "C" void coin_lab_set_no_s4(int v);         // conv_gemm.hip
extern "C" void coin_lab_set_roi_bwd_old(int v);   // roi_align.hip
// tools/ab_bench.py round5_kernels: every kernel-level change of round 6 off at once (GEMM dispatch, BatchNorm launch shapes, RoIAlign
// backward).  Defined here because tools/gemm_lab links the GEMM objects only.
extern "C" void coin_lab_set_round5_kernels(int v) {
  coin_lab_set_no_s4(v);
  coin_lab_set_bn_old(v);
  coin_lab_set_roi_bwd_old(v);
}
#endif

#define BN_DISPATCH(dtype, EXPR_F32, EXPR_BF16) \
  do {                                          \
    if ((dtype) == COIN_F32) {                  \
      typedef float T;                          \
      EXPR_F32;                                 \
    } else {                                    \
      typedef bf16_t T;                         \
      EXPR_BF16;                                \
    }                                           \
  } while (0)

extern "C" int coin_bn_stats(const void* x, int N, int H, int W, int C, float eps, float momentum, float* sums_workspace,
                             float* mean, float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int dtype,
                             void* stream) {
  int rc = bn_check(x, N, H, W, C, 1, dtype);
  if (rc) return rc;
  if (!sums_workspace || !mean || !rstd) return COIN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = (int64_t)N * H * W;
  const int v = dtype == COIN_F32 ? 4 : 8;
  const int ncg = C / v, tpr = ncg < BN_THREADS ? ncg : BN_THREADS, rpi = BN_THREADS / tpr;
  int64_t g = (M + rpi - 1) / rpi;
  if (g > 512) g = 512;   // (<= COIN_BN_MAX_PARTS: the workspace bound)
  const size_t lds = sizeof(float) * BN_THREADS * 2 * v;
#define GO(T) bn_stats_kernel<T><<<(int)g, BN_THREADS, lds, st>>>((const T*)x, M, C, sums_workspace); \
  bn_finalize_kernel<T><<<(C + 63) / 64, 1024, 0, st>>>((const T*)x, sums_workspace, (int)g, M, C, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked)
  BN_DISPATCH(dtype, GO(float), GO(bf16_t));
#undef GO
  return coin_launch_status();
}

extern "C" int coin_bn_apply_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 const void* residual, void* y, uint8_t* relu_mask, int N, int H, int W, int C, int relu, int pool, int dtype,
                                 void* stream) {
  int rc = bn_check(x, N, H, W, C, pool, dtype);
  if (rc) return rc;
  if (!mean || !rstd || !gamma || !beta || !y) return COIN_EINVAL;
  if (pool == 2 && residual) return COIN_ESHAPE;
  if (pool == 2 && (H < 2 || W < 2)) return COIN_ESHAPE;
  const int v = dtype == COIN_F32 ? 4 : 8;
  hipStream_t st = (hipStream_t)stream;
  if (pool == 0) {
    if (N > 65535) return COIN_ESHAPE;
    dim3 grid((C / v + BN_THREADS - 1) / BN_THREADS, N);
#define GO(T) bn_apply_mean_kernel<T><<<grid, BN_THREADS, 0, st>>>((const T*)x, mean, rstd, gamma, beta, (const T*)residual, (T*)y, N, H * W, C, relu, relu_mask)
    BN_DISPATCH(dtype, GO(float), GO(bf16_t));
#undef GO
    return coin_launch_status();
  }
  const int grid = walk_grid((int64_t)N * (H / pool) * (W / pool), C / v);
#define GO(T)                                                                                                                     \
  if (pool == 1)                                                                                                                  \
    bn_apply_kernel<T, 1><<<grid, BN_THREADS, 0, st>>>((const T*)x, mean, rstd, gamma, beta, (const T*)residual, (T*)y, N, H, W, C, relu, residual ? relu_mask : nullptr); \
  else                                                                                                                            \
    bn_apply_kernel<T, 2><<<grid, BN_THREADS, 0, st>>>((const T*)x, mean, rstd, gamma, beta, (const T*)residual, (T*)y, N, H, W, C, relu, nullptr)
  BN_DISPATCH(dtype, GO(float), GO(bf16_t));
#undef GO
  return coin_launch_status();
}

extern "C" int coin_bn_bwd(const void* x, const void* dy, const void* y, const uint8_t* relu_mask, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int N, int H, int W, int C, int relu, int pool,
                           float* dsums /* [2C]: dbeta, dgamma */, void* dx, void* d_residual, int dtype, void* stream) {
  int rc = bn_check(x, N, H, W, C, pool, dtype);
  if (rc) return rc;
  if (!dy || !mean || !rstd || !gamma || !beta || !dsums || !dx) return COIN_EINVAL;
  if (relu_mask && pool == 2) return COIN_ESHAPE;
  if (relu && pool == 1 && d_residual && !y && !relu_mask) return COIN_EINVAL;  // with a residual the ReLU mask is only in the saved output / bit mask
  if (pool == 0 && d_residual && !y && !relu_mask) return COIN_EINVAL;          // pool == 0: `y` is the forward's residual input
  if (pool == 2 && d_residual) return COIN_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == COIN_F32 ? 4 : 8;
  const int64_t M = (int64_t)N * H * W;
  const int ncg = C / v, tpr = ncg < BN_THREADS ? ncg : BN_THREADS, rpi = BN_THREADS / tpr;
  int64_t g = (M + rpi - 1) / rpi;
  // reduction workgroups: 768 (three per CU = one resident round at this kernel's 3 waves per SIMD) for the res5 tensors, 512 below
  const int parts_cap = g_bn_parts_cap > 0 ? g_bn_parts_cap : (M * ncg >= (16LL << 20) ? COIN_BN_MAX_PARTS : 512);
  if (g > parts_cap) g = parts_cap;
  const size_t lds = sizeof(float) * BN_THREADS * 2 * v;
  const int wg = walk_grid(M, ncg);
  const int mode = relu_mask ? ((relu ? 1 : 0) | 4) : ((relu ? 1 : 0) | (y ? 2 : 0));
#define GO2(T, P, MD)                                                                                                              \
  bn_bwd_reduce_kernel<T, P, MD><<<(int)g, BN_THREADS, lds, st>>>((const T*)x, (const T*)dy, (const T*)y, mean, rstd, gamma, beta, N, H, W, C, relu, dsums, relu_mask); \
  bn_bwd_finalize_kernel<<<(2 * C + 63) / 64, 1024, 0, st>>>(dsums, (int)g, C);                                                      \
  bn_bwd_dx_kernel<T, P, MD><<<wg, BN_THREADS, 0, st>>>((const T*)x, (const T*)dy, (const T*)y, mean, rstd, gamma, beta, dsums, N, H, W, C, relu, (T*)dx, (T*)d_residual, relu_mask)
#define GO1(T, P)             \
  if (mode == 0) {            \
    GO2(T, P, 0);             \
  } else if (mode == 1) {     \
    GO2(T, P, 1);             \
  } else if (mode == 2) {     \
    GO2(T, P, 2);             \
  } else if (mode == 3) {     \
    GO2(T, P, 3);             \
  } else if (mode == 4) {     \
    GO2(T, P, 4);             \
  } else {                    \
    GO2(T, P, 5);             \
  }
#define GO(T)                 \
  if (pool == 1) {            \
    GO1(T, 1);                \
  } else if (pool == 2) {     \
    GO1(T, 2);                \
  } else {                    \
    GO1(T, 0);                \
  }
  BN_DISPATCH(dtype, GO(float), GO(bf16_t));
#undef GO
#undef GO1
#undef GO2
  return coin_launch_status();
}

extern "C" int coin_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  int rc = bn_check(x, N, H, W, C, 2, dtype);
  if (rc) return rc;
  if (!y || H < 2 || W < 2) return COIN_EINVAL;
  const int v = dtype == COIN_F32 ? 4 : 8;
  hipStream_t st = (hipStream_t)stream;
#define GO(T) avgpool2_fwd_kernel<T><<<stream_grid((int64_t)N * (H / 2) * (W / 2) * (C / v)), BN_THREADS, 0, st>>>((const T*)x, (T*)y, N, H, W, C)
  BN_DISPATCH(dtype, GO(float), GO(bf16_t));
#undef GO
  return coin_launch_status();
}

extern "C" int coin_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream) {
  int rc = bn_check(dy, N, H, W, C, 2, dtype);
  if (rc) return rc;
  if (!dx || H < 2 || W < 2) return COIN_EINVAL;
  const int v = dtype == COIN_F32 ? 4 : 8;
  hipStream_t st = (hipStream_t)stream;
#define GO(T) avgpool2_bwd_kernel<T><<<stream_grid((int64_t)N * H * W * (C / v)), BN_THREADS, 0, st>>>((const T*)dy, (T*)dx, N, H, W, C)
  BN_DISPATCH(dtype, GO(float), GO(bf16_t));
#undef GO
  return coin_launch_status();
}
