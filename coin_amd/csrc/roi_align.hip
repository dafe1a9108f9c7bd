// RoIAlign forward / backward for gfx950.
//
// Replaces coin/modeling/roi_heads/clip_roi_heads.py:172-176 (`self.pooler(features, boxes)` ->
// detectron2 ROIPooler -> torchvision.ops.roi_align, aligned=True, sampling_ratio=0).
//
// Fast path = channels-last (NHWC): C is the contiguous axis, so every bilinear tap of one output
// pixel is one fully coalesced 16-byte-per-lane row read and the output row is one coalesced
// 16-byte-per-lane store.  The res4 map (8.5 MB bf16 / 17 MB f32 per view) is read through L2 /
// Infinity Cache; the output (R*ph*pw*C elements) is the HBM stream that bounds the kernel.
// All row-blocks of one RoI are placed on one XCD (blocks b and b+8 share an XCD) so that the
// RoI's footprint is fetched into a single L2.
//
// Backward (NHWC, bins up to 16 x 16: every shape the detector uses) is an atomic-free gather per 4 x 8 map tile x channel slab
// (roi_align_bwd_gather_kernel): the RoIs that touch the tile are walked in index order, the per-(RoI, tile) bilinear weight
// tables are built once per workgroup in LDS, every map element is stored exactly once in a fixed summation order
// (bit-reproducible).  Larger bins fall back to a separable per-RoI gather with one float atomic per footprint pixel
// (roi_align_bwd_nhwc_kernel; float atomics are bounded at ~1.3 TB/s of added bytes, MI355X_MICROARCH.md).
//
// The NCHW kernels are the layout-compatible (reference layout) path: one thread per element.
#include <stdlib.h>
#include "common.h"

// lab-only predicates (tools/roibench.py: kernel without its loads / stores) compile to `false` unless built with -DCOIN_LAB
#ifdef COIN_LAB
#define ROI_LAB(x) (x)
#else
#define ROI_LAB(x) false
#endif

namespace {

struct RoiGeom {
  int n;           // batch index
  float x0, y0;    // roi start (feature px, after the aligned offset)
  float bw, bh;    // bin size
  int gw, gh;      // sampling grid per bin
  float inv_count; // 1 / max(gw*gh, 1)
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ r, int ph, int pw, float scale,
                                            int sampling_ratio, int aligned) {
  RoiGeom g;
  g.n = (int)r[0];
  const float off = aligned ? 0.5f : 0.0f;
  g.x0 = r[1] * scale - off;
  g.y0 = r[2] * scale - off;
  float x1 = r[3] * scale - off;
  float y1 = r[4] * scale - off;
  float rw = x1 - g.x0, rh = y1 - g.y0;
  if (!aligned) {
    rw = fmaxf(rw, 1.0f);
    rh = fmaxf(rh, 1.0f);
  }
  g.bw = rw / (float)pw;
  g.bh = rh / (float)ph;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
  int cnt = g.gh * g.gw;
  g.inv_count = 1.0f / (float)(cnt > 1 ? cnt : 1);
  return g;
}

// One axis of torchvision's bilinear_interpolate: returns false if the coordinate is outside
// [-1, size]; otherwise low/high indices and the weights (hi_w multiplies `low`).
__device__ __forceinline__ bool axis_taps(float v, int size, int& lo, int& hi, float& w_lo, float& w_hi) {
  if (v < -1.0f || v > (float)size) return false;
  if (v <= 0.f) v = 0.f;
  lo = (int)v;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    v = (float)lo;
  } else {
    hi = lo + 1;
  }
  float l = v - (float)lo;
  w_hi = l;         // weight of `hi`
  w_lo = 1.0f - l;  // weight of `lo`
  return true;
}

template <typename T>
__device__ __forceinline__ void vec_fma(float (&acc)[Vec16<T>::N], const typename Vec16<T>::type& v, float w) {
#pragma unroll
  for (int i = 0; i < Vec16<T>::N; ++i) acc[i] += w * (float)v[i];
}

// summed bilinear weight that the samples of bin `b` put on map coordinate `k` (one axis); same ops as axis_taps
__device__ __forceinline__ float bin_weight(float start, float binsz, int grid, int b, int k, int size) {
  float wsum = 0.f;
  for (int i = 0; i < grid; ++i) {
    const float v = start + (float)b * binsz + ((float)i + 0.5f) * binsz / (float)grid;
    int lo, hi;
    float wl, wh;
    if (!axis_taps(v, size, lo, hi, wl, wh)) continue;
    if (lo == k) wsum += wl;
    if (hi == k) wsum += wh;
  }
  return wsum;
}

// ------------------------------------------------------------------------------------------
// forward, NHWC
// ------------------------------------------------------------------------------------------
// Pyramid levels of the multi-level pooler (FPN extension): RoI r is pooled from level roi_level[r] -- one launch for all levels.
struct RoiLevelTable {
  const void* feat[COIN_ROI_MAX_LEVELS];
  int H[COIN_ROI_MAX_LEVELS], W[COIN_ROI_MAX_LEVELS];
  float scale[COIN_ROI_MAX_LEVELS];
};

template <typename T, bool ML>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_kernel(
    const T* __restrict__ feat, const float* __restrict__ rois, T* __restrict__ out, int C, int H, int W, int R,
    int ph, int pw, float scale, int sampling_ratio, int aligned, const RoiLevelTable lv, const int* __restrict__ roi_level, int nlevels) {
  constexpr int VEC = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  // XCD-aware: the ph row-blocks of one RoI share blockIdx % 8.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, s = bid >> 3;
  const int roi = xcd + 8 * (s / ph);
  const int py = s % ph;
  if (roi >= R) return;
  if (ML) {
    int l = roi_level[roi];
    l = l < 0 ? 0 : (l >= nlevels ? nlevels - 1 : l);
    // (wave-uniform selects from the by-value table: scalar registers)
    feat = (const T*)lv.feat[0];
    H = lv.H[0];
    W = lv.W[0];
    scale = lv.scale[0];
#pragma unroll
    for (int k = 1; k < COIN_ROI_MAX_LEVELS; ++k)
      if (l == k) {
        feat = (const T*)lv.feat[k];
        H = lv.H[k];
        W = lv.W[k];
        scale = lv.scale[k];
      }
  }
  const RoiGeom g = roi_geom(rois + (size_t)roi * 5, ph, pw, scale, sampling_ratio, aligned);
  const int ncg = C / VEC;                     // 16-byte channel groups per pixel
  const int tpp = ncg < 256 ? ncg : 256;       // threads per pixel
  const int ppi = 256 / tpp;                   // pixels in flight per block iteration
  const int slot = threadIdx.x / tpp;
  const int cg0 = threadIdx.x - slot * tpp;
  if (slot >= ppi) return;
  const T* __restrict__ fmap = feat + (size_t)g.n * H * W * C;
  T* __restrict__ orow = out + ((size_t)roi * ph + py) * pw * C;
  const float ybase = g.y0 + (float)py * g.bh;
  for (int px = slot; px < pw; px += ppi) {
    const float xbase = g.x0 + (float)px * g.bw;
    for (int cg = cg0; cg < ncg; cg += tpp) {
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      const T* __restrict__ fc = fmap + (size_t)cg * VEC;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = ybase + ((float)iy + 0.5f) * g.bh / (float)g.gh;
        int yl, yh;
        float wyl, wyh;
        if (!axis_taps(y, H, yl, yh, wyl, wyh)) continue;
        const T* __restrict__ rl = fc + (size_t)yl * W * C;
        const T* __restrict__ rh = fc + (size_t)yh * W * C;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = xbase + ((float)ix + 0.5f) * g.bw / (float)g.gw;
          int xl, xh;
          float wxl, wxh;
          if (!axis_taps(x, W, xl, xh, wxl, wxh)) continue;
          const vec_t v1 = *reinterpret_cast<const vec_t*>(rl + (size_t)xl * C);
          const vec_t v2 = *reinterpret_cast<const vec_t*>(rl + (size_t)xh * C);
          const vec_t v3 = *reinterpret_cast<const vec_t*>(rh + (size_t)xl * C);
          const vec_t v4 = *reinterpret_cast<const vec_t*>(rh + (size_t)xh * C);
          vec_fma<T>(acc, v1, wyl * wxl);
          vec_fma<T>(acc, v2, wyl * wxh);
          vec_fma<T>(acc, v3, wyh * wxl);
          vec_fma<T>(acc, v4, wyh * wxh);
        }
      }
      vec_t o;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = (T)(acc[i] * g.inv_count);
      __builtin_nontemporal_store(o, reinterpret_cast<vec_t*>(orow + (size_t)px * C + (size_t)cg * VEC));
    }
  }
}



typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16 bytes of channels as pairs of floats (the arithmetic of the column walk runs on v_pk_mul_f32 / v_pk_fma_f32: the kernel is
// bound by vector-ALU issue -- a wave64 instruction occupies its 16-lane SIMD for 4 cycles -- and packed fp32 halves the count)
__device__ __forceinline__ void to_pairs(const bf16x8& v, f32x2 (&o)[4]) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k].x = __builtin_bit_cast(float, w[k] << 16);
    o[k].y = __builtin_bit_cast(float, w[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ void to_pairs(const f32x4& v, f32x2 (&o)[2]) {
  o[0].x = v[0];
  o[0].y = v[1];
  o[1].x = v[2];
  o[1].y = v[3];
}

// bin_weight with the per-sample step (bin size / grid) and the bin's start precomputed: no division in the loop.  The sample
// coordinate may differ from bin_weight's by one rounding (the interpolation is continuous in it).
__device__ __forceinline__ float bin_weight_step(float bin_start, float step, int grid, int k, int size) {
  float wsum = 0.f;
  for (int i = 0; i < grid; ++i) {
    const float v = bin_start + ((float)i + 0.5f) * step;
    int lo, hi;
    float wl, wh;
    if (!axis_taps(v, size, lo, hi, wl, wh)) continue;
    if (lo == k) wsum += wl;
    if (hi == k) wsum += wh;
  }
  return wsum;
}

// one sweep of the column-walk forward: NR consecutive map rows (weights wk, wave-uniform) x the columns [xmin, xmax]
template <typename T, int NR, int PG>
__device__ __forceinline__ void cols_sweep(const T* __restrict__ fy, size_t rs, int C, int xmin, int xmax, const float (&wk)[4],
                                           float bin_start, float step, int gw, int W, f32x2 (&acc)[PG][Vec16<T>::N / 2]) {
  constexpr int VP = Vec16<T>::N / 2;
  typedef typename Vec16<T>::type vec_t;
  // fold one column (its NR rows are in `v`) into the bins that have weight on it
  auto fold = [&](const vec_t (&v)[NR], int x) {
    const float wxv = bin_weight_step(bin_start, step, gw, x, W);
    if (__ballot(wxv != 0.f) == 0ull) return;
    f32x2 col[VP];
    {
      f32x2 f[VP];
      to_pairs(v[0], f);
      const f32x2 w0 = {wk[0], wk[0]};
#pragma unroll
      for (int k = 0; k < VP; ++k) col[k] = w0 * f[k];
    }
#pragma unroll
    for (int u = 1; u < NR; ++u) {
      f32x2 f[VP];
      to_pairs(v[u], f);
      const f32x2 wu = {wk[u], wk[u]};
#pragma unroll
      for (int k = 0; k < VP; ++k) col[k] = __builtin_elementwise_fma(wu, f[k], col[k]);
    }
#pragma unroll
    for (int p = 0; p < PG; ++p) {
      const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wxv), p));
      if (w != 0.f) {
        const f32x2 ww = {w, w};
#pragma unroll
        for (int k = 0; k < VP; ++k) acc[p][k] = __builtin_elementwise_fma(ww, col[k], acc[p][k]);
      }
    }
  };
  // two register sets, used alternately: the rows of column x + 1 are requested before column x is folded in, and no register
  // of a pending load is copied (a `v = vn` rotation made the compiler wait for the prefetch at the top of every iteration)
  vec_t va[NR], vb[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) va[u] = *reinterpret_cast<const vec_t*>(fy + u * rs + (size_t)xmin * C);
#pragma nounroll
  for (int x = xmin; x <= xmax; x += 2) {
    const int x1 = x + 1 <= xmax ? x + 1 : xmax;        // (past the end the last column is requested again: an L2 hit instead of a branch)
#pragma unroll
    for (int u = 0; u < NR; ++u) vb[u] = *reinterpret_cast<const vec_t*>(fy + u * rs + (size_t)x1 * C);
    fold(va, x);
    const int x2 = x + 2 <= xmax ? x + 2 : xmax;
#pragma unroll
    for (int u = 0; u < NR; ++u) va[u] = *reinterpret_cast<const vec_t*>(fy + u * rs + (size_t)x2 * C);
    if (x + 1 <= xmax) fold(vb, x + 1);
  }
}

// Column-walk forward (channels-last, pw in {7, 14}: the kernel the detector's poolers run).  RoIAlign is separable:
//   out[py][px][c] = inv_count * sum_x Wx[px][x] * ( sum_y Wy[py][y] * feat[y][x][c] ),
// Wy[py][y] / Wx[px][x] = the summed bilinear weights that the samples of bin row py / bin column px put on map row y / column x
// (bin_weight: the same tap rule, sample by sample).  One WAVE owns (RoI, py, 7 bins, 64 channel groups of 16 bytes) and keeps its 7
// output pixels as accumulators in registers (128 VGPRs: 4 waves per SIMD); it walks the DISTINCT map columns of its bins' footprint
// once, reading the (<= grid + 1) rows that bin row py touches -- the next column's rows are requested before the current one is
// folded in -- instead of 4 taps per sample per output pixel: a 14 x 14 pooling of a box of 8 x 8 map cells reads 18 rows of
// 16 bytes per lane and output row instead of 56; 20 x 20 cells: 63 instead of 224.  Weights are wave-uniform (lane p computes the
// column weights of bin p, v_readlane / v_readfirstlane turn them into scalars, zero weights are skipped by scalar branches).
// No LDS, no barrier.  Measured (tools/roibench.py, [4,50,83,1024] bf16, 2048 boxes, 14 x 14): boxes of 32-400 px 0.358 ms against
// 0.44-0.45 ms of the per-sample kernel, 300-800 px 0.87 against 1.34, 16-96 px 0.241 against 0.28; without its stores 0.25 / 0.85 /
// 0.14 ms, without its loads 0.14 ms (the 822 MB write stream at 5.9 TB/s).  The gather phase is bound by vector-ALU issue, not by
// L2 / Infinity Cache bandwidth or latency (SQ counters, profiles/r4_roi_pmc_sq.csv: 193 M wave-instructions x 4 cycles on 1 024
// SIMDs = 0.31 ms for the first version): packed fp32 arithmetic and division-free bin weights took it from 0.31 to 0.25 ms; 8-byte
// lanes (twice the instructions) 0.57 ms; pinning channel slabs to XCDs so that a map slab fits one L2 0.405 ms; a real prefetch
// (two register sets, counted `vmcnt`) instead of one the compiler had to wait for: no change.
template <typename T, bool ML, int SW /* map rows per sweep */>
__global__ __launch_bounds__(256) void roi_align_fwd_cols_kernel(
    const T* __restrict__ feat, const float* __restrict__ rois, T* __restrict__ out, int C, int H, int W, int R,
    int ph, int pw, float scale, int sampling_ratio, int aligned, const RoiLevelTable lv, const int* __restrict__ roi_level, int nlevels,
    int wpr /* waves per (RoI, py, 7-bin group): ceil(C / VEC / 64) */, int bpr /* blocks per RoI */, int variant /* lab: bit 2 no loads, bit 3 no stores */) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int PG = 7;                     // bins per wave: 7 accumulators x VEC channels in registers (pw = 7 or 14)
  typedef typename Vec16<T>::type vec_t;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, s = bid >> 3;
  const int ngrp = pw / PG;
  const int roi = xcd + 8 * (s / bpr);      // every block of one RoI runs on one XCD
  const int unit = (s % bpr) * 4 + (threadIdx.x >> 6);
  if (roi >= R || unit >= ph * ngrp * wpr) return;
  const int slab = unit % wpr, grp = (unit / wpr) % ngrp, py = unit / (wpr * ngrp);
  const int px0 = grp * PG;
  const int lane = threadIdx.x & 63;
  if (ML) {
    int l = roi_level[roi];
    l = l < 0 ? 0 : (l >= nlevels ? nlevels - 1 : l);
    feat = (const T*)lv.feat[0];
    H = lv.H[0];
    W = lv.W[0];
    scale = lv.scale[0];
#pragma unroll
    for (int k = 1; k < COIN_ROI_MAX_LEVELS; ++k)
      if (l == k) {
        feat = (const T*)lv.feat[k];
        H = lv.H[k];
        W = lv.W[k];
        scale = lv.scale[k];
      }
  }
  RoiGeom g = roi_geom(rois + (size_t)roi * 5, ph, pw, scale, sampling_ratio, aligned);
  {  // the geometry is wave-uniform: keep it in scalar registers (the float arithmetic above ran on the vector unit)
    auto uf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    g.n = __builtin_amdgcn_readfirstlane(g.n);
    g.gw = __builtin_amdgcn_readfirstlane(g.gw);
    g.gh = __builtin_amdgcn_readfirstlane(g.gh);
    g.x0 = uf(g.x0);
    g.y0 = uf(g.y0);
    g.bw = uf(g.bw);
    g.bh = uf(g.bh);
    g.inv_count = uf(g.inv_count);
  }
  const int ncg = C / VEC;
  const int cg = slab * 64 + lane;
  const int cgl = cg < ncg ? cg : ncg - 1;       // lanes past the last channel group repeat it (loads stay in bounds); not stored
  f32x2 acc[PG][VEC / 2];
#pragma unroll
  for (int p = 0; p < PG; ++p)
#pragma unroll
    for (int k = 0; k < VEC / 2; ++k) acc[p][k] = f32x2{0.f, 0.f};
  if (g.gh > 0 && g.gw > 0 && !ROI_LAB(variant & 4)) {
    // footprint of this output row (rows) and of this wave's 7 bins (columns); a superset is harmless (weights of untouched lines are 0)
    const float xstep = g.bw / (float)g.gw, ystep = g.bh / (float)g.gh;
    const float ya = g.y0 + (float)py * g.bh + 0.5f * ystep, yb = g.y0 + (float)py * g.bh + ((float)g.gh - 0.5f) * ystep;
    const float xa = g.x0 + (float)px0 * g.bw + 0.5f * xstep;
    const float xb = g.x0 + (float)(px0 + PG - 1) * g.bw + ((float)g.gw - 0.5f) * xstep;
    const float ylo = fminf(ya, yb), yhi = fmaxf(ya, yb), xlo = fminf(xa, xb), xhi = fmaxf(xa, xb);
    if (!(yhi < -1.0f || ylo > (float)H || xhi < -1.0f || xlo > (float)W)) {
      const int ymin = (int)fminf(fmaxf(floorf(ylo), 0.f), (float)(H - 1)), ymax = (int)fminf(fmaxf(floorf(yhi) + 1.f, 0.f), (float)(H - 1));
      const int xmin = (int)fminf(fmaxf(floorf(xlo), 0.f), (float)(W - 1)), xmax = (int)fminf(fmaxf(floorf(xhi) + 1.f, 0.f), (float)(W - 1));
      const T* __restrict__ fmap = feat + (size_t)g.n * H * W * C + (size_t)cgl * VEC;
      const int pxl = px0 + (lane < PG ? lane : PG - 1);
      const float xbin = g.x0 + (float)pxl * g.bw, ybin = g.y0 + (float)py * g.bh;   // start of this lane's bin column / this wave's bin row
      for (int y0 = ymin; y0 <= ymax; y0 += SW) {           // SW map rows per sweep (a bin row touches grid + 1 rows)
        const int nr = (ymax - y0 + 1) < SW ? (ymax - y0 + 1) : SW;
        float wk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < SW; ++u) {
          // (every lane computes the same value: wave-uniform by construction; readfirstlane makes it a scalar)
          const float w = u < nr ? bin_weight_step(ybin, ystep, g.gh, y0 + u, H) : 0.f;
          wk[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w)));
        }
        if (wk[0] == 0.f && wk[1] == 0.f && wk[2] == 0.f && wk[3] == 0.f) continue;
        const T* __restrict__ fy = fmap + (size_t)y0 * W * C;
        const size_t rs = (size_t)W * C;
        // One instantiation per row count: the loads of a sweep are UNCONDITIONAL inside it, so that the compiler can count them
        // (`s_waitcnt vmcnt(NR)`: the next column's rows stay in flight while the current column is folded in).  With the loads
        // under `if (weight != 0)` it had to wait for vmcnt(0) before every use and the prefetch bought nothing.
        if (nr == 1)
          cols_sweep<T, 1, PG>(fy, rs, C, xmin, xmax, wk, xbin, xstep, g.gw, W, acc);
        else if (SW == 2 || nr == 2)
          cols_sweep<T, 2, PG>(fy, rs, C, xmin, xmax, wk, xbin, xstep, g.gw, W, acc);
        else if (SW == 3 || nr == 3)
          cols_sweep<T, (SW >= 3 ? 3 : 2), PG>(fy, rs, C, xmin, xmax, wk, xbin, xstep, g.gw, W, acc);
        else
          cols_sweep<T, (SW >= 4 ? 4 : 2), PG>(fy, rs, C, xmin, xmax, wk, xbin, xstep, g.gw, W, acc);
      }
    }
  }
  if (cg < ncg && !ROI_LAB((variant & 8) && acc[0][0].x != 12345.f)) {
    T* __restrict__ orow = out + (((size_t)roi * ph + py) * pw + px0) * C + (size_t)cg * VEC;
#pragma unroll
    for (int p = 0; p < PG; ++p) {
      vec_t o;
#pragma unroll
      for (int k = 0; k < VEC / 2; ++k) {
        const f32x2 r = acc[p][k] * f32x2{g.inv_count, g.inv_count};
        o[2 * k] = (T)r.x;
        o[2 * k + 1] = (T)r.y;
      }
      *reinterpret_cast<vec_t*>(orow + (size_t)p * C) = o;   // (non-temporal stores: 0.41 against 0.36 ms at the benchmark's box sizes)
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, NHWC, bins larger than 16 x 16 only (the tile-gather kernel below serves the detector's 14 x 14 / 7 x 7 bins): separable gather
// with float atomics.
//   d feat[y][x][c] = sum_py sum_px Wy[py][y] * Wx[px][x] * gout[py][px][c] / count
// Wy[py][y] (resp. Wx) is the summed bilinear weight of all samples of bin row py that touch map row y.
// A block owns one RoI x 64 channels (lane = channel): it builds the two small weight tables in LDS,
// stages the RoI's [ph*pw][64] gradient tile in LDS once, then each wave walks footprint pixels,
// contracts the (few) bins that touch the pixel out of LDS and issues ONE 256-byte-contiguous float
// atomic per footprint pixel.  No per-tap atomics, no LDS atomics.
// ------------------------------------------------------------------------------------------
constexpr int BWD_CH = 64;  // channels per block (lane = channel)

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc_kernel(
    const T* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat, int C, int H, int W,
    int R, int ph, int pw, float scale, int sampling_ratio, int aligned, int tile_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);                       // [ph*pw][64]
  float* wy = reinterpret_cast<float*>(smem + tile_bytes);    // [ph][H]
  float* wx = wy + ph * H;                                    // [pw][W]
  int* ylo = reinterpret_cast<int*>(wx + pw * W);             // [H] first bin row touching y
  int* yhi = ylo + H;                                         // [H] last bin row (+1)
  int* xlo = yhi + H;                                         // [W]
  int* xhi = xlo + W;                                         // [W]
  int* box = xhi + W;                                         // fy0, fy1, fx0, fx1
  const int roi = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * BWD_CH + lane;
  const bool c_ok = c < C;
  const RoiGeom g = roi_geom(rois + (size_t)roi * 5, ph, pw, scale, sampling_ratio, aligned);
  if (g.gh <= 0 || g.gw <= 0) return;
  float* __restrict__ gmap = gfeat + (size_t)g.n * H * W * C;
  const T* __restrict__ go = gout + (size_t)roi * ph * pw * C;

  for (int i = threadIdx.x; i < ph * H + pw * W; i += 256) wy[i] = 0.f;  // wy and wx are adjacent
  // stage the gradient tile (coalesced over channels)
  for (int p = wave; p < ph * pw; p += 4) tile[p * BWD_CH + lane] = c_ok ? go[(size_t)p * C + c] : (T)0.f;
  __syncthreads();
  // one thread per bin row / bin column accumulates its own table row: no conflicts
  if ((int)threadIdx.x < ph) {
    const int py = threadIdx.x;
    float* row = wy + py * H;
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = g.y0 + (float)py * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      int lo, hi;
      float wl, wh;
      if (!axis_taps(y, H, lo, hi, wl, wh)) continue;
      row[lo] += wl;
      row[hi] += wh;
    }
  } else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + pw) {
    const int px = threadIdx.x - 64;
    float* row = wx + px * W;
    for (int ix = 0; ix < g.gw; ++ix) {
      const float x = g.x0 + (float)px * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
      int lo, hi;
      float wl, wh;
      if (!axis_taps(x, W, lo, hi, wl, wh)) continue;
      row[lo] += wl;
      row[hi] += wh;
    }
  }
  __syncthreads();
  // per map row / column: contiguous range of bins with non-zero weight
  for (int i = threadIdx.x; i < H + W; i += 256) {
    const bool isy = i < H;
    const int k = isy ? i : i - H;
    const int nb = isy ? ph : pw, ld = isy ? H : W;
    const float* tab = isy ? wy : wx;
    int lo = nb, hi = 0;
    for (int b = 0; b < nb; ++b) {
      if (tab[b * ld + k] != 0.f) {
        lo = b < lo ? b : lo;
        hi = b + 1;
      }
    }
    (isy ? ylo : xlo)[k] = lo;
    (isy ? yhi : xhi)[k] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = H, b = -1, l = W, r = -1;
    for (int y = 0; y < H; ++y)
      if (yhi[y] > ylo[y]) { a = y < a ? y : a; b = y; }
    for (int x = 0; x < W; ++x)
      if (xhi[x] > xlo[x]) { l = x < l ? x : l; r = x; }
    box[0] = a; box[1] = b; box[2] = l; box[3] = r;
  }
  __syncthreads();
  const int fy0 = box[0], fy1 = box[1], fx0 = box[2], fx1 = box[3];
  if (fy1 < fy0 || fx1 < fx0) return;
  const int fw = fx1 - fx0 + 1, npx = (fy1 - fy0 + 1) * fw;
  for (int q = wave; q < npx; q += 4) {
    const int y = fy0 + q / fw, x = fx0 + q % fw;
    const int pl = ylo[y], phh = yhi[y], ql = xlo[x], qh = xhi[x];
    if (phh <= pl || qh <= ql) continue;
    float acc = 0.f;
    for (int py = pl; py < phh; ++py) {
      const float a = wy[py * H + y];
      float rowacc = 0.f;
      for (int px = ql; px < qh; ++px) rowacc += wx[px * W + x] * (float)tile[(py * pw + px) * BWD_CH + lane];
      acc += a * rowacc;
    }
    if (c_ok) atomicAdd(gmap + ((size_t)y * W + x) * C + c, acc * g.inv_count);
  }
}

constexpr int LIST_CAP = 4096;  // RoIs per tile list (ordered compaction)

// ------------------------------------------------------------------------------------------
// backward, NHWC, atomic-free gather per map tile (the shipped NHWC path for ph, pw <= 16):
//   d feat[n][y][x][c] = sum over RoIs r of image n:  sum_py sum_px Wy_r[py][y] * Wx_r[px][x] * gout[r][py][px][c] / count_r
// A block owns a 4-row x 8-column tile of one image's gradient map x one channel part; wave w owns tile row w, a lane one 16-byte
// channel vector of it: 8 pixel accumulators x VEC channels in registers, every map element stored exactly once with plain
// stores, RoIs summed in index order (bit-reproducible; no float atomics: their ~1.3 TB/s ceiling bound the scatter form).
//   phase 0: ordered compaction of the RoIs whose footprint may touch the tile;
//   phase 1 (per 32 list entries, whole block): the bilinear weight tables of those RoIs restricted to the tile -- one thread
//            per (RoI, axis, bin) walks the bin's samples once (the forward's arithmetic) -- plus bit masks of the non-zero
//            entries.  Round 1's kernel (8x8 tile, wave-private tables) recomputed these per wave and per 128-channel block
//            (32x redundant) inside its load -> use chain; now they are off the critical path and the inner loop is loads + FMAs;
//   phase 2: each wave streams the gradient bins with non-zero weight on its row: 16 bytes per lane per bin (1 KiB per
//            wave-instruction), converted once and fanned out to the <= 3 tile columns they touch.
// Round 6 (tools/roibwd_bench.py on the RoIs of a real step, profiles/r4_real_rois.pt; lab switches DBG): of 0.46 ms the lists + tables +
// stores are 0.05, the loop skeleton + conversions 0.09, the loads +0.2 and the column fan-out +0.2 -- neither HBM (bf16 and f32 take
// the same time at twice the bytes) nor FMA issue bound the launch but each wave's serial chain load -> wait -> branch ladder, at 2-4
// waves per SIMD.  Hence ALG 1 (shipped for bf16): ALL bin rows with weight on a map row are combined first (py ascending, one fused
// multiply-add each, starting from 0) and fanned out to the tile columns ONCE per bin column -- a real step's boxes put 2.9 bin rows on
// a map row, so the ladder runs 2.9x less often -- with rounds of PYR x PXC = 2 x 7 bins in flight (70 % of the (RoI, row) pairs have
// <= 2 bin rows: 4 x 4 rounds were half empty) and 16-byte lanes (half the bin visits per byte): 0.46 -> 0.35 ms; with the weight tables of
// a list entry held one bin per LANE and turned into scalars by v_readlane (LT: no LDS read + wait + v_readfirstlane per bin column and
// bin row in the inner loops; same bits) 0.335 ms.  Measured and
// dropped (same tool): taller / wider tiles (8 x 8: 0.48, 8 x 16: 0.98 ms -- the column ladder and the table build grow with the tile
// while the workgroup count shrinks; the 1.5x re-read of bins that straddle 4 x 8 tiles is served by L2 / Infinity Cache and is NOT
// what bounds the launch); two map rows per wave (fewer bin visits, fewer waves: 0.51); a per-wave LDS ring filled by LDS-DMA with
// producer / consumer cursors (prefetch depth independent of the box geometry: 0.43-0.55 ms -- deeper rings cost occupancy, and ~70
// scalar + vector instructions per 1 KiB bin at 3 waves per SIMD bound it at any depth).
// ------------------------------------------------------------------------------------------
// Tile shape = template parameters: BT_ROWS map rows (= waves of the workgroup) x BT_COLS columns; BT_LC list entries per table chunk.

// VEC = channels per lane (4: 16-byte f32 / 8-byte bf16 loads; a wave covers 256 channels).  With 8 bf16 channels per lane the 1024-channel
// res4 gradient gave 1 144 workgroups whose longest (the central tiles, touched by ~40 % of an image's boxes) bounded the launch.
template <typename T, int VEC, int NW /* waves */, int BT_COLS, int BT_LC, int DBG = 0 /* lab: 1 no gather phase, 2 no loads, 4 no fan-out, 8 stores only */,
          int ALG = 0 /* 1: all bin rows of a map row are combined before the column fan-out */, int RPW = 1 /* map rows per wave (ALG 1) */,
          int PYR = 4, int PXC = 4 /* ALG 1: bin rows x bin columns in flight per round */,
          bool LT = false /* ALG 1: the weight tables of a list entry in registers, one bin per lane, read with v_readlane */>
__global__ __launch_bounds__(NW * 64) void roi_align_bwd_gather_kernel(
    const T* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat, int C, int H, int W, int R, int ph, int pw,
    float scale, int sampling_ratio, int aligned, int tiles_x, int tiles_y, int ntiles, int nparts, const int* __restrict__ roi_level, int level) {
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ unsigned short list[LIST_CAP];
  constexpr int NT = NW * 64, BT_ROWS = NW * RPW;
  static_assert(BT_LC * BT_ROWS + BT_LC <= NT && BT_COLS % 4 == 0 && BT_COLS <= 32 && (RPW == 1 || ALG == 1), "tile shape");
  __shared__ int wave_cnt[NW];
  __shared__ int list_n;
  __shared__ __attribute__((aligned(16))) float wyt[BT_LC][16][BT_ROWS];
  __shared__ __attribute__((aligned(16))) float wxt[BT_LC][16][BT_COLS];
  __shared__ unsigned xmk[BT_LC][16];      // bit t: bin column px has weight on tile column t
  __shared__ unsigned ymk[BT_LC][BT_ROWS]; // bit py: bin row py has weight on tile row r
  __shared__ int pxr[BT_LC][2];            // bin columns [lo, hi) with any weight on the tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware order: blocks b and b + 8 share an XCD (L2); give each XCD a contiguous run of tiles so that the bins which
  // straddle tile borders are re-read from the same L2 (speed only, placement is not assumed for correctness)
  const int nblk = ntiles * nparts;
  const int q = nblk / 8, rem = nblk % 8, xcd = blockIdx.x & 7;
  const int logical = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
  const int part = logical / ntiles, tile = logical - part * ntiles;
  const int tx0 = (tile % tiles_x) * BT_COLS;
  const int ty0 = ((tile / tiles_x) % tiles_y) * BT_ROWS;
  const int n = tile / (tiles_x * tiles_y);
  const int c0 = (part * 64 + lane) * VEC;
  const bool c_ok = c0 < C;

  float accr[RPW][BT_COLS][VEC];   // [row of the wave][tile column][channel]
  float (&acc)[BT_COLS][VEC] = accr[0];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int t = 0; t < BT_COLS; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) accr[rr][t][i] = 0.f;

  for (int base = 0; base < ((DBG & 8) ? 0 : R); base += LIST_CAP) {
    const int lim = (R - base) < LIST_CAP ? (R - base) : LIST_CAP;
    // ---- phase 0: ordered compaction of the RoIs of image n whose footprint may touch this tile
    if (threadIdx.x == 0) list_n = 0;
    __syncthreads();
    for (int i0 = 0; i0 < lim; i0 += NT) {
      const int i = i0 + threadIdx.x;
      bool hit = false;
      if (i < lim) {
        const RoiGeom g = roi_geom(rois + (size_t)(base + i) * 5, ph, pw, scale, sampling_ratio, aligned);
        if (g.n == n && g.gh > 0 && g.gw > 0 && (roi_level == nullptr || roi_level[base + i] == level)) {   // multi-level pooler: this level's RoIs only
          // conservative footprint: first / last sample of each axis, +-1 pixel for the bilinear taps (the border rows /
          // columns also collect the clamped samples from [-1, 0] and [size-1, size])
          const float ys0 = g.y0 + 0.5f * g.bh / (float)g.gh, ys1 = g.y0 + ((float)(ph - 1) + ((float)g.gh - 0.5f) / (float)g.gh) * g.bh;
          const float xs0 = g.x0 + 0.5f * g.bw / (float)g.gw, xs1 = g.x0 + ((float)(pw - 1) + ((float)g.gw - 0.5f) / (float)g.gw) * g.bw;
          const float ylo = fminf(ys0, ys1), yhi = fmaxf(ys0, ys1), xlo = fminf(xs0, xs1), xhi = fmaxf(xs0, xs1);
          hit = (yhi >= (float)(ty0 - 1)) && (ylo <= (float)(ty0 + BT_ROWS)) && (xhi >= (float)(tx0 - 1)) && (xlo <= (float)(tx0 + BT_COLS));
        }
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int off = list_n;
      for (int w = 0; w < wave; ++w) off += wave_cnt[w];
      if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
      __syncthreads();
      if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < NW; ++w) tot += wave_cnt[w];
        list_n += tot;
      }
      __syncthreads();
    }
    const int nlist = list_n;
    for (int lb = 0; lb < nlist; lb += BT_LC) {
      const int lcn = (nlist - lb) < BT_LC ? (nlist - lb) : BT_LC;
      // ---- phase 1: weight tables of the chunk's RoIs on this tile; task = (list entry, axis, bin)
      for (int task = threadIdx.x; task < lcn * 32; task += NT) {
        const int li = task >> 5, b = task & 15, isx = (task >> 4) & 1;
        const RoiGeom g = roi_geom(rois + (size_t)(base + (int)list[lb + li]) * 5, ph, pw, scale, sampling_ratio, aligned);
        if (isx) {
          float w[BT_COLS];
#pragma unroll
          for (int t = 0; t < BT_COLS; ++t) w[t] = 0.f;
          if (b < pw) {
            for (int ix = 0; ix < g.gw; ++ix) {
              const float v = g.x0 + (float)b * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
              int lo, hi;
              float wl, wh;
              if (!axis_taps(v, W, lo, hi, wl, wh)) continue;
#pragma unroll
              for (int t = 0; t < BT_COLS; ++t) {
                if (lo == tx0 + t) w[t] += wl;
                if (hi == tx0 + t) w[t] += wh;
              }
            }
          }
          unsigned m = 0;
#pragma unroll
          for (int t = 0; t < BT_COLS; ++t) {
            wxt[li][b][t] = w[t];
            m |= (w[t] != 0.f ? 1u : 0u) << t;
          }
          xmk[li][b] = m;
        } else {
          float w[BT_ROWS];
#pragma unroll
          for (int t = 0; t < BT_ROWS; ++t) w[t] = 0.f;
          if (b < ph) {
            for (int iy = 0; iy < g.gh; ++iy) {
              const float v = g.y0 + (float)b * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
              int lo, hi;
              float wl, wh;
              if (!axis_taps(v, H, lo, hi, wl, wh)) continue;
#pragma unroll
              for (int t = 0; t < BT_ROWS; ++t) {
                if (lo == ty0 + t) w[t] += wl;
                if (hi == ty0 + t) w[t] += wh;
              }
            }
          }
#pragma unroll
          for (int t = 0; t < BT_ROWS; ++t) wyt[li][b][t] = w[t] * g.inv_count;
        }
      }
      __syncthreads();
      if ((int)threadIdx.x < lcn * BT_ROWS) {
        const int li = threadIdx.x / BT_ROWS, r = threadIdx.x % BT_ROWS;
        unsigned m = 0;
        for (int py = 0; py < ph; ++py) m |= (wyt[li][py][r] != 0.f ? 1u : 0u) << py;
        ymk[li][r] = m;
      } else if ((int)threadIdx.x >= BT_LC * BT_ROWS && (int)threadIdx.x < BT_LC * BT_ROWS + lcn) {
        const int li = threadIdx.x - BT_LC * BT_ROWS;
        int lo = pw, hi = 0;
        for (int px = 0; px < pw; ++px)
          if (xmk[li][px]) {
            lo = px < lo ? px : lo;
            hi = px + 1;
          }
        pxr[li][0] = lo;
        pxr[li][1] = hi;
      }
      __syncthreads();
      // ---- phase 2: wave w gathers for tile row w
      for (int li = 0; li < ((DBG & 1) ? 0 : lcn); ++li) {
        unsigned ymr[RPW];   // per row of the wave: the bin rows with weight on it
        unsigned ym = 0;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          ymr[rr] = (unsigned)__builtin_amdgcn_readfirstlane((int)ymk[li][wave * RPW + rr]);
          ym |= ymr[rr];
        }
        const int pa = __builtin_amdgcn_readfirstlane(pxr[li][0]), pe = __builtin_amdgcn_readfirstlane(pxr[li][1]);
        if (ym == 0 || pa >= pe) continue;
        const int roi = base + (int)list[lb + li];
        const T* __restrict__ go = gout + (size_t)roi * ph * pw * C + (c_ok ? c0 : 0);
        if constexpr (ALG == 1) {
          const unsigned ym0 = ym;
          // LT: lane p holds bin p's entries of this list entry's tables -- the inner loops then turn them into scalars with v_readlane
          // instead of one LDS round trip (read, wait, v_readfirstlane) per bin column and bin row
          unsigned xml = 0;
          float wxl[BT_COLS], wyl[RPW];
          if constexpr (LT) {
            const int pb = lane & 15;
            xml = xmk[li][pb];
#pragma unroll
            for (int k = 0; k < BT_COLS / 4; ++k) {
              const f32x4 q4 = *reinterpret_cast<const f32x4*>(&wxt[li][pb][4 * k]);
              wxl[4 * k] = q4[0];
              wxl[4 * k + 1] = q4[1];
              wxl[4 * k + 2] = q4[2];
              wxl[4 * k + 3] = q4[3];
            }
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) wyl[rr] = wyt[li][pb][wave * RPW + rr];
          }
          for (int px = pa; px < pe; px += PXC) {
            // tf[rr][j] = sum over the bin rows py with weight on map row rr of Wy[py][rr] * gout[py][px + j]  (py ascending, PYR per
            // round: up to PYR * PXC gradient bins, one wave-instruction each, in flight), fanned out to the tile columns ONCE per bin column.
            // With RPW = 2 a bin that touches both rows of the wave is loaded (and converted) once, and one column ladder serves both.
            float tf[RPW][PXC][VEC];
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
              for (int j = 0; j < PXC; ++j)
#pragma unroll
                for (int i = 0; i < VEC; ++i) tf[rr][j][i] = 0.f;
            ym = ym0;
            while (ym) {
              int py[PYR];
              vec_t v[PYR][PXC];
#pragma unroll
              for (int k = 0; k < PYR; ++k) {
                py[k] = ym ? __builtin_ctz(ym) : -1;
                ym &= ym - 1;   // 0 stays 0
                if (py[k] < 0) continue;
                const T* __restrict__ grow = go + (size_t)py[k] * pw * C;
#pragma unroll
                for (int j = 0; j < PXC; ++j) {
                  const int pp = px + j < pe ? px + j : pe - 1;
                  v[k][j] = *reinterpret_cast<const vec_t*>(grow + (size_t)pp * C);
                }
              }
#pragma unroll
              for (int k = 0; k < PYR; ++k) {
                if (py[k] < 0) continue;
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                  if (RPW > 1 && !(ymr[rr] & (1u << py[k]))) continue;   // no weight on this row: nothing is added (not even 0 * x)
                  float a;
                  if constexpr (LT) a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyl[rr]), py[k]));
                  else a = wyt[li][py[k]][wave * RPW + rr];
#pragma unroll
                  for (int j = 0; j < PXC; ++j)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) tf[rr][j][i] += a * (float)v[k][j][i];
                }
              }
            }
#pragma unroll
            for (int j = 0; j < PXC; ++j) {
              if (px + j >= pe) break;
              unsigned mk;
              if constexpr (LT) mk = (unsigned)__builtin_amdgcn_readlane((int)xml, px + j);
              else mk = (unsigned)__builtin_amdgcn_readfirstlane((int)xmk[li][px + j]);
              if (mk == 0) continue;
              f32x4 wq[BT_COLS / 4];
              if constexpr (!LT) {
#pragma unroll
                for (int k = 0; k < BT_COLS / 4; ++k) wq[k] = *reinterpret_cast<const f32x4*>(&wxt[li][px + j][4 * k]);
              }
#pragma unroll
              for (int t = 0; t < BT_COLS; ++t) {
                if (mk & (1u << t)) {
                  asm volatile("; col taken");  // a real wave-uniform branch (see the forward kernel)
                  float wt;
                  if constexpr (LT) wt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wxl[t]), px + j));
                  else wt = wq[t >> 2][t & 3];
#pragma unroll
                  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) accr[rr][t][i] += wt * tf[rr][j][i];
                }
              }
            }
          }
          continue;
        }
        while (ym) {
          // two bin rows per round: up to 8 gradient bins (one wave-instruction each) in flight
          const int py0 = __builtin_ctz(ym);
          ym &= ym - 1;
          const int py1 = ym ? __builtin_ctz(ym) : -1;
          if (py1 >= 0) ym &= ym - 1;
          const float a0 = wyt[li][py0][wave];
          const float a1 = py1 >= 0 ? wyt[li][py1][wave] : 0.f;
          const T* __restrict__ grow0 = go + (size_t)py0 * pw * C;
          const T* __restrict__ grow1 = go + (size_t)(py1 >= 0 ? py1 : py0) * pw * C;
          for (int px = pa; px < pe; px += 4) {
            vec_t v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int pp = px + j < pe ? px + j : pe - 1;
              if constexpr ((DBG & 2) != 0) {   // no loads
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[j][i] = v[4 + j][i] = (T)(float)(pp + i);
                continue;
              }
              v[j] = *reinterpret_cast<const vec_t*>(grow0 + (size_t)pp * C);
              if (py1 >= 0) v[4 + j] = *reinterpret_cast<const vec_t*>(grow1 + (size_t)pp * C);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (px + j >= pe) break;
              unsigned mk = (unsigned)__builtin_amdgcn_readfirstlane((int)xmk[li][px + j]);
              if constexpr ((DBG & 4) != 0) {   // no fan-out: the loaded values are consumed by one accumulator
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[0][i] += (float)v[j][i] + (float)v[4 + j][i];
                mk = 0;
              }
              if (mk == 0) continue;
              f32x4 wq[BT_COLS / 4];
#pragma unroll
              for (int k = 0; k < BT_COLS / 4; ++k) wq[k] = *reinterpret_cast<const f32x4*>(&wxt[li][px + j][4 * k]);
              // the two rows are combined first (fixed order), then fanned out to the tile columns
              float vf[VEC];
#pragma unroll
              for (int i = 0; i < VEC; ++i) vf[i] = a0 * (float)v[j][i];
              if (py1 >= 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) vf[i] += a1 * (float)v[4 + j][i];
              }
#pragma unroll
              for (int t = 0; t < BT_COLS; ++t) {
                if (mk & (1u << t)) {
                  asm volatile("; col taken");  // a real wave-uniform branch (see the forward kernel)
                  const float wt = wq[t >> 2][t & 3];
#pragma unroll
                  for (int i = 0; i < VEC; ++i) acc[t][i] += wt * vf[i];
                }
              }
            }
          }
        }
      }
      __syncthreads();  // the tables are rewritten for the next chunk
    }
    __syncthreads();  // list is rebuilt for the next chunk of RoIs
  }
  if (!c_ok) return;
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int y = ty0 + wave * RPW + rr;
    if (y >= H) break;
    float* __restrict__ gmap = gfeat + (((size_t)n * H + y) * W) * C + c0;
#pragma unroll
    for (int t = 0; t < BT_COLS; ++t) {
      const int x = tx0 + t;
      if (x < W) {
#pragma unroll
        for (int i0 = 0; i0 < VEC; i0 += 4) {
          const f32x4 o = {accr[rr][t][i0], accr[rr][t][i0 + 1], accr[rr][t][i0 + 2], accr[rr][t][i0 + 3]};
          *reinterpret_cast<f32x4*>(gmap + (size_t)x * C + i0) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// NCHW (reference layout) kernels: one thread per output element / per grad element
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nchw_kernel(
    const T* __restrict__ feat, const float* __restrict__ rois, T* __restrict__ out, int C, int H, int W, int R,
    int ph, int pw, float scale, int sampling_ratio, int aligned) {
  const size_t total = (size_t)R * C * ph * pw;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int px = (int)(idx % pw);
    const int py = (int)((idx / pw) % ph);
    const int c = (int)((idx / ((size_t)pw * ph)) % C);
    const int roi = (int)(idx / ((size_t)pw * ph * C));
    const RoiGeom g = roi_geom(rois + (size_t)roi * 5, ph, pw, scale, sampling_ratio, aligned);
    const T* __restrict__ f = feat + ((size_t)g.n * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = g.y0 + (float)py * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      int yl, yh;
      float wyl, wyh;
      if (!axis_taps(y, H, yl, yh, wyl, wyh)) continue;
      for (int ix = 0; ix < g.gw; ++ix) {
        const float x = g.x0 + (float)px * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
        int xl, xh;
        float wxl, wxh;
        if (!axis_taps(x, W, xl, xh, wxl, wxh)) continue;
        acc += wyl * wxl * (float)f[yl * W + xl] + wyl * wxh * (float)f[yl * W + xh] +
               wyh * wxl * (float)f[yh * W + xl] + wyh * wxh * (float)f[yh * W + xh];
      }
    }
    out[idx] = (T)(acc * g.inv_count);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_nchw_kernel(
    const T* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat, int C, int H, int W,
    int R, int ph, int pw, float scale, int sampling_ratio, int aligned) {
  const size_t total = (size_t)R * C * ph * pw;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int px = (int)(idx % pw);
    const int py = (int)((idx / pw) % ph);
    const int c = (int)((idx / ((size_t)pw * ph)) % C);
    const int roi = (int)(idx / ((size_t)pw * ph * C));
    const RoiGeom g = roi_geom(rois + (size_t)roi * 5, ph, pw, scale, sampling_ratio, aligned);
    float* __restrict__ f = gfeat + ((size_t)g.n * C + c) * H * W;
    const float gv = (float)gout[idx] * g.inv_count;
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = g.y0 + (float)py * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      int yl, yh;
      float wyl, wyh;
      if (!axis_taps(y, H, yl, yh, wyl, wyh)) continue;
      for (int ix = 0; ix < g.gw; ++ix) {
        const float x = g.x0 + (float)px * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
        int xl, xh;
        float wxl, wxh;
        if (!axis_taps(x, W, xl, xh, wxl, wxh)) continue;
        atomicAdd(f + yl * W + xl, wyl * wxl * gv);
        atomicAdd(f + yl * W + xh, wyl * wxh * gv);
        atomicAdd(f + yh * W + xl, wyh * wxl * gv);
        atomicAdd(f + yh * W + xh, wyh * wxh * gv);
      }
    }
  }
}

int check_common(const void* a, const void* rois, const void* b, int N, int C, int H, int W, int layout, int R,
                 int ph, int pw, int dtype) {
  if (R > 0 && (!a || !b || !rois)) return COIN_EINVAL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || ph <= 0 || pw <= 0) return COIN_EINVAL;
  if (layout != COIN_NCHW && layout != COIN_NHWC) return COIN_EINVAL;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  if (layout == COIN_NHWC) {
    const int vec = dtype == COIN_F32 ? 4 : 8;
    if (C % vec) return COIN_ESHAPE;
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15)) return COIN_EALIGN;
  }
  return COIN_OK;
}

}  // namespace

// launch of the column-walk forward (pw in {7, 14}); returns false if the shape is not served by it
#ifndef BF16_SW
#define BF16_SW 3
#endif
#ifdef COIN_LAB
int g_roi_fwd_variant = 0;   // lab hook (tools/roibench.py): bit 0 = per-sample kernel for every shape; bits 2 / 3 = column-walk kernel without its loads / stores
#else
static constexpr int g_roi_fwd_variant = 0;   // the product library has no variant switch (ROI_LAB() is the constant false)
#endif

template <bool ML>
static bool launch_fwd_cols(const void* feat, const float* rois, void* out, int C, int H, int W, int R, int ph, int pw, float scale,
                            int sampling_ratio, int aligned, const RoiLevelTable& lv, const int* roi_level, int nlevels, int dtype, hipStream_t st) {
  if ((pw != 7 && pw != 14) || (g_roi_fwd_variant & 1)) return false;
  const int vec = dtype == COIN_F32 ? 4 : 8;
  const int wpr = (C / vec + 63) / 64;
  const int bpr = (ph * (pw / 7) * wpr + 3) / 4;
  const long long nblk = (long long)((R + 7) / 8) * 8 * bpr;
  if (nblk > 0x7fffffffLL) return false;
  const int grid = (int)nblk;
#define GO(T, SW) roi_align_fwd_cols_kernel<T, ML, SW><<<grid, 256, 0, st>>>((const T*)feat, rois, (T*)out, C, H, W, R, ph, pw, scale, sampling_ratio, aligned, lv, roi_level, nlevels, wpr, bpr, g_roi_fwd_variant)
  if (dtype == COIN_F32)
    GO(float, 3);
  else
    GO(bf16_t, BF16_SW);
#undef GO
  return true;
}

#ifdef COIN_LAB
extern "C" void coin_roi_align_lab_variant(int v) { g_roi_fwd_variant = v; }
#endif

extern "C" int coin_roi_align_fwd(const void* feat, int N, int C, int H, int W, int layout, const float* rois,
                                  int R, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                  void* out, int dtype, void* stream) {
  int rc = check_common(feat, rois, out, N, C, H, W, layout, R, ph, pw, dtype);
  if (rc) return rc;
  if (R == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (layout == COIN_NHWC) {
    const int grid = ((R + 7) / 8) * 8 * ph;
    const RoiLevelTable none = {};
    if (launch_fwd_cols<false>(feat, rois, out, C, H, W, R, ph, pw, spatial_scale, sampling_ratio, aligned, none, nullptr, 0, dtype, st))
      return coin_launch_status();
    if (dtype == COIN_F32)
      roi_align_fwd_nhwc_kernel<float, false><<<grid, 256, 0, st>>>((const float*)feat, rois, (float*)out, C, H, W, R, ph,
                                                                      pw, spatial_scale, sampling_ratio, aligned, none, nullptr, 0);
    else
      roi_align_fwd_nhwc_kernel<bf16_t, false><<<grid, 256, 0, st>>>((const bf16_t*)feat, rois, (bf16_t*)out, C, H, W, R,
                                                                       ph, pw, spatial_scale, sampling_ratio, aligned, none, nullptr, 0);
  } else {
    const size_t total = (size_t)R * C * ph * pw;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (dtype == COIN_F32)
      roi_align_fwd_nchw_kernel<float><<<grid, 256, 0, st>>>((const float*)feat, rois, (float*)out, C, H, W, R, ph,
                                                               pw, spatial_scale, sampling_ratio, aligned);
    else
      roi_align_fwd_nchw_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)feat, rois, (bf16_t*)out, C, H, W, R,
                                                                ph, pw, spatial_scale, sampling_ratio, aligned);
  }
  return coin_launch_status();
}

#ifdef COIN_LAB
int g_roi_bwd_cfg = 0;   // lab hook (tools/roibwd_bench.py): tile shape of the backward gather
extern "C" void coin_roi_align_lab_bwd_cfg(int v) { g_roi_bwd_cfg = v; }
extern "C" void coin_lab_set_roi_bwd_old(int v) { g_roi_bwd_cfg = v ? 10 : 0; }   // tools/ab_bench.py roi_bwd_old: round 2's backward
#else
constexpr int g_roi_bwd_cfg = 0;
#endif

template <typename T, int VEC, int NW, int COLS, int LC, int DBG = 0, int ALG = 0, int RPW = 1, int PYR = 4, int PXC = 4, bool LT = false>
static void launch_bwd_gather(const T* grad_out, const float* rois, float* grad_feat, int N, int C, int H, int W, int R, int ph, int pw, float scale,
                              int sampling_ratio, int aligned, const int* roi_level, int level, hipStream_t st) {
  constexpr int ROWS = NW * RPW;
  const int tiles_x = (W + COLS - 1) / COLS, tiles_y = (H + ROWS - 1) / ROWS;
  const int ntiles = tiles_x * tiles_y * N;
  const int nparts = (C + 64 * VEC - 1) / (64 * VEC);
  roi_align_bwd_gather_kernel<T, VEC, NW, COLS, LC, DBG, ALG, RPW, PYR, PXC, LT><<<ntiles * nparts, NW * 64, 0, st>>>(grad_out, rois, grad_feat, C, H, W, R, ph, pw, scale, sampling_ratio,
                                                                                         aligned, tiles_x, tiles_y, ntiles, nparts, roi_level, level);
}

static int roi_align_bwd_impl(const void* grad_out, int N, int C, int H, int W, int layout, const float* rois,
                              int R, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                              float* grad_feat, int dtype, void* stream, const int* roi_level, int level) {
  int rc = check_common(grad_out, rois, grad_feat, N, C, H, W, layout, R, ph, pw, dtype);
  if (rc) return rc;
  if (!grad_feat) return COIN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (R == 0 || layout == COIN_NCHW) {  // grad_feat is always fully (over)written
    if (hipMemsetAsync(grad_feat, 0, sizeof(float) * (size_t)N * C * H * W, st) != hipSuccess) return coin_launch_status();
    if (R == 0) return COIN_OK;
  }
  if (layout == COIN_NHWC && ph <= 16 && pw <= 16) {
    // atomic-free gather per map tile: writes every element of grad_feat exactly once
    const int cfg = ROI_LAB(true) ? g_roi_bwd_cfg : 0;
#define BWD_GO(T, VEC, ROWS, COLS, LC, ...)                                                                                             \
  launch_bwd_gather<T, VEC, ROWS, COLS, LC, ##__VA_ARGS__>((const T*)grad_out, rois, grad_feat, N, C, H, W, R, ph, pw, spatial_scale, sampling_ratio, \
                                            aligned, roi_level, level, st)
#ifdef COIN_LAB   // tools/roibwd_bench.py: tile shapes, the two summation orders, rounds in flight, kernel minus its parts
#define BWD_LAB_CASES(T)                                   \
    case 1: BWD_GO(T, 4, 8, 8, 32); break;                 \
    case 2: BWD_GO(T, 4, 8, 16, 16); break;                \
    case 3: BWD_GO(T, 4, 4, 16, 32); break;                \
    case 4: BWD_GO(T, 8, 8, 8, 32); break;                 \
    case 10: BWD_GO(T, 4, 4, 8, 32); break;                \
    case 11: BWD_GO(T, 4, 4, 8, 32, 1); break;             \
    case 12: BWD_GO(T, 4, 4, 8, 32, 2); break;             \
    case 14: BWD_GO(T, 4, 4, 8, 32, 4); break;             \
    case 16: BWD_GO(T, 4, 4, 8, 32, 6); break;             \
    case 20: BWD_GO(T, 4, 4, 8, 32, 0, 1); break;          \
    case 22: BWD_GO(T, 8, 4, 8, 32, 0, 1); break;          \
    case 23: BWD_GO(T, 8, 8, 8, 32, 0, 1); break;          \
    case 24: BWD_GO(T, 4, 4, 8, 16, 0, 1, 2); break;       \
    case 34: BWD_GO(T, 4, 4, 8, 32, 0, 1, 1, 2, 4); break; \
    case 37: BWD_GO(T, 8, 4, 8, 32, 0, 1, 1, 2, 7); break; \
    case 41: BWD_GO(T, 4, 4, 8, 32, 0, 1, 1, 2, 7); break; \
    case 47: BWD_GO(T, 8, 4, 8, 32, 0, 1, 1, 2, 7, true); break; \
    case 48: BWD_GO(T, 4, 4, 8, 32, 0, 1, 1, 2, 7, true); break; \
    case 49: BWD_GO(T, 4, 4, 8, 32, 0, 1, 1, 2, 4, true); break;
#else
#define BWD_LAB_CASES(T)
#endif
    // shipped: bf16 -- 16-byte lanes (512 channels per workgroup), all bin rows of a map row combined before the column fan-out, rounds of
    // 2 bin rows x 7 bin columns in flight, the tables of a list entry held one bin per lane (v_readlane instead of LDS round trips); f32 -- round 2's pairs of bin rows x 4 bin columns (the wider rounds cost it registers:
    // 0.51 vs 0.57 ms on a real step's boxes but 1.32 vs 1.23 ms on 300-800 px boxes)
    if (dtype == COIN_F32) {
      switch (cfg) {
        BWD_LAB_CASES(float)
        default: BWD_GO(float, 4, 4, 8, 32); break;
      }
    } else {
      switch (cfg) {
        BWD_LAB_CASES(bf16_t)
        default: BWD_GO(bf16_t, 8, 4, 8, 32, 0, 1, 1, 2, 7, true); break;
      }
    }
#undef BWD_LAB_CASES
#undef BWD_GO
  } else if (roi_level != nullptr) {
    return COIN_ESHAPE;   // the level filter exists in the tile-gather kernel only (channels-last, bins <= 16 x 16)
  } else if (layout == COIN_NHWC) {
    if (hipMemsetAsync(grad_feat, 0, sizeof(float) * (size_t)N * C * H * W, st) != hipSuccess) return coin_launch_status();
    if (R == 0) return COIN_OK;
    dim3 grid(R, (C + BWD_CH - 1) / BWD_CH);
    const int es = dtype == COIN_F32 ? 4 : 2;
    const int tile_bytes = ((ph * pw * BWD_CH * es) + 15) & ~15;
    const size_t lds = (size_t)tile_bytes + sizeof(float) * ((size_t)ph * H + (size_t)pw * W) + sizeof(int) * (2 * (size_t)(H + W) + 4);
    if (lds > 160 * 1024) return COIN_ESHAPE;  // feature map too large for the LDS weight tables
    if (dtype == COIN_F32) {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_nhwc_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      roi_align_bwd_nhwc_kernel<float><<<grid, 256, lds, st>>>((const float*)grad_out, rois, grad_feat, C, H, W, R, ph,
                                                                 pw, spatial_scale, sampling_ratio, aligned, tile_bytes);
    } else {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_nhwc_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      roi_align_bwd_nhwc_kernel<bf16_t><<<grid, 256, lds, st>>>((const bf16_t*)grad_out, rois, grad_feat, C, H, W, R,
                                                                  ph, pw, spatial_scale, sampling_ratio, aligned, tile_bytes);
    }
  } else {
    const size_t total = (size_t)R * C * ph * pw;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (dtype == COIN_F32)
      roi_align_bwd_nchw_kernel<float><<<grid, 256, 0, st>>>((const float*)grad_out, rois, grad_feat, C, H, W, R, ph,
                                                               pw, spatial_scale, sampling_ratio, aligned);
    else
      roi_align_bwd_nchw_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)grad_out, rois, grad_feat, C, H, W, R,
                                                                ph, pw, spatial_scale, sampling_ratio, aligned);
  }
  return coin_launch_status();
}

extern "C" int coin_roi_align_bwd(const void* grad_out, int N, int C, int H, int W, int layout, const float* rois,
                                  int R, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                  float* grad_feat, int dtype, void* stream) {
  return roi_align_bwd_impl(grad_out, N, C, H, W, layout, rois, R, ph, pw, spatial_scale, sampling_ratio, aligned, grad_feat, dtype, stream, nullptr, 0);
}

// ---- multi-level pooler (FPN extension; no counterpart in the reference, whose pooler has one level: clip_roi_heads.py:172-176)
extern "C" int coin_roi_align_fwd_levels(const coin_roi_level* levels, int nlevels, int N, int C, const float* rois, const int* roi_level,
                                         int R, int ph, int pw, int sampling_ratio, int aligned, void* out, int dtype, void* stream) {
  if (!levels || nlevels <= 0 || nlevels > COIN_ROI_MAX_LEVELS) return COIN_EINVAL;
  if (R > 0 && !roi_level) return COIN_EINVAL;
  RoiLevelTable t = {};
  for (int k = 0; k < nlevels; ++k) {
    int rc = check_common(levels[k].feat, rois, out, N, C, levels[k].H, levels[k].W, COIN_NHWC, R, ph, pw, dtype);
    if (rc) return rc;
    t.feat[k] = levels[k].feat;
    t.H[k] = levels[k].H;
    t.W[k] = levels[k].W;
    t.scale[k] = levels[k].spatial_scale;
  }
  if (R == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (launch_fwd_cols<true>(nullptr, rois, out, C, 0, 0, R, ph, pw, 0.f, sampling_ratio, aligned, t, roi_level, nlevels, dtype, st))
    return coin_launch_status();
  const int grid = ((R + 7) / 8) * 8 * ph;
  if (dtype == COIN_F32)
    roi_align_fwd_nhwc_kernel<float, true><<<grid, 256, 0, st>>>(nullptr, rois, (float*)out, C, 0, 0, R, ph, pw, 0.f, sampling_ratio, aligned, t, roi_level, nlevels);
  else
    roi_align_fwd_nhwc_kernel<bf16_t, true><<<grid, 256, 0, st>>>(nullptr, rois, (bf16_t*)out, C, 0, 0, R, ph, pw, 0.f, sampling_ratio, aligned, t, roi_level, nlevels);
  return coin_launch_status();
}

extern "C" int coin_roi_align_bwd_level(const void* grad_out, int N, int C, int H, int W, const float* rois, const int* roi_level, int level,
                                        int R, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned, float* grad_feat, int dtype,
                                        void* stream) {
  if (R > 0 && !roi_level) return COIN_EINVAL;
  if (ph > 16 || pw > 16) return COIN_ESHAPE;
  return roi_align_bwd_impl(grad_out, N, C, H, W, COIN_NHWC, rois, R, ph, pw, spatial_scale, sampling_ratio, aligned, grad_feat, dtype, stream, roi_level, level);
}
