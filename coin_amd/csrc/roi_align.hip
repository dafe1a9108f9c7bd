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
// Backward accumulates each RoI's footprint in LDS (ds_add_f32, lane = channel, conflict-free)
// and flushes it with 256-byte-contiguous global float atomics: ~(samples*4)/(footprint) fewer
// global atomics than the per-tap scheme, which is what bounds a naive backward on this chip
// (~1.3 TB/s of atomic bytes, MI355X_MICROARCH.md "Global float atomics").
//
// The NCHW kernels are the layout-compatible (reference layout) path: one thread per element.
#include "common.h"

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

// ------------------------------------------------------------------------------------------
// forward, NHWC
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_kernel(
    const T* __restrict__ feat, const float* __restrict__ rois, T* __restrict__ out, int C, int H, int W, int R,
    int ph, int pw, float scale, int sampling_ratio, int aligned) {
  constexpr int VEC = Vec16<T>::N;
  typedef typename Vec16<T>::type vec_t;
  // XCD-aware: the ph row-blocks of one RoI share blockIdx % 8.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, s = bid >> 3;
  const int roi = xcd + 8 * (s / ph);
  const int py = s % ph;
  if (roi >= R) return;
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

// ------------------------------------------------------------------------------------------
// backward, NHWC: LDS footprint accumulation
// ------------------------------------------------------------------------------------------
constexpr int BWD_LDS_FLOATS = 16384;  // 64 KiB window -> 2 blocks per CU
constexpr int BWD_CH = 64;             // channels per block (lane = channel)

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc_kernel(
    const T* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat, int C, int H, int W,
    int R, int ph, int pw, float scale, int sampling_ratio, int aligned) {
  __shared__ float acc[BWD_LDS_FLOATS];
  const int roi = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * BWD_CH + lane;
  const bool c_ok = c < C;
  const RoiGeom g = roi_geom(rois + (size_t)roi * 5, ph, pw, scale, sampling_ratio, aligned);
  if (g.gh <= 0 || g.gw <= 0) return;
  float* __restrict__ gmap = gfeat + (size_t)g.n * H * W * C;
  const T* __restrict__ go = gout + (size_t)roi * ph * pw * C;

  // horizontal footprint of the RoI (conservative; every tap is re-checked against it)
  int fx0, fx1;
  {
    float xf = g.x0 + 0.5f * g.bw / (float)g.gw;
    float xl = g.x0 + (float)(pw - 1) * g.bw + ((float)g.gw - 0.5f) * g.bw / (float)g.gw;
    if (xf > xl) { float t = xf; xf = xl; xl = t; }
    fx0 = (int)fmaxf(floorf(xf) - 1.f, 0.f);
    fx1 = (int)fminf(fmaxf(floorf(xl) + 2.f, 0.f), (float)(W - 1));
    if (fx0 > W - 1) fx0 = W - 1;
    if (fx1 < fx0) fx1 = fx0;
  }
  const int fw = fx1 - fx0 + 1;
  const int rows_cap = BWD_LDS_FLOATS / (fw * BWD_CH);  // 0 for footprints wider than 256 px: direct atomics
  int wb = 0, wrows = 0;                                 // current window [wb, wb + wrows)

  auto flush = [&]() {
    const int n = wrows * fw * BWD_CH;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int pix = i >> 6;  // BWD_CH == 64: lane == i & 63
      const float v = acc[i];
      if (c_ok && v != 0.f) {
        const int row = pix / fw, col = pix - row * fw;
        atomicAdd(gmap + ((size_t)(wb + row) * W + (fx0 + col)) * C + c, v);
      }
    }
  };
  auto clear = [&](int rows) {
    const int n = rows * fw * BWD_CH;
    for (int i = threadIdx.x; i < n; i += 256) acc[i] = 0.f;
  };

  for (int py = 0; py < ph; ++py) {
    const float ybase = g.y0 + (float)py * g.bh;
    // rows this bin row can touch (conservative)
    float yf = ybase + 0.5f * g.bh / (float)g.gh;
    float yl_ = ybase + ((float)g.gh - 0.5f) * g.bh / (float)g.gh;
    if (yf > yl_) { float t = yf; yf = yl_; yl_ = t; }
    int a = (int)fmaxf(floorf(yf) - 1.f, 0.f);
    int b = (int)fminf(fmaxf(floorf(yl_) + 2.f, 0.f), (float)(H - 1));
    if (a > H - 1) a = H - 1;
    if (b < a) b = a;
    const bool fits = (b - a + 1) <= rows_cap;
    if (fits && (wrows == 0 || b >= wb + rows_cap || a < wb)) {
      // open a new window starting at row a
      __syncthreads();
      if (wrows > 0) flush();
      __syncthreads();
      wb = a;
      wrows = rows_cap < (H - a) ? rows_cap : (H - a);
      clear(wrows);
      __syncthreads();
    }
    for (int px = wave; px < pw; px += 4) {
      float gv = 0.f;
      if (c_ok) gv = (float)go[((size_t)py * pw + px) * C + c] * g.inv_count;
      const float xbase = g.x0 + (float)px * g.bw;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = ybase + ((float)iy + 0.5f) * g.bh / (float)g.gh;
        int yl, yh;
        float wyl, wyh;
        if (!axis_taps(y, H, yl, yh, wyl, wyh)) continue;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = xbase + ((float)ix + 0.5f) * g.bw / (float)g.gw;
          int xl, xh;
          float wxl, wxh;
          if (!axis_taps(x, W, xl, xh, wxl, wxh)) continue;
          if (!c_ok) continue;
          const int ys[2] = {yl, yh};
          const int xs[2] = {xl, xh};
          const float wy[2] = {wyl, wyh};
          const float wx[2] = {wxl, wxh};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float w = wy[j] * wx[i] * gv;
              const int yy = ys[j], xx = xs[i];
              // wave-uniform test: (yy, xx) do not depend on the lane
              if (fits && yy >= wb && yy < wb + wrows && xx >= fx0 && xx <= fx1) {
                atomicAdd(&acc[((yy - wb) * fw + (xx - fx0)) * BWD_CH + lane], w);
              } else {
                atomicAdd(gmap + ((size_t)yy * W + xx) * C + c, w);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (wrows > 0) flush();
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
  if (!a || !b || (!rois && R > 0)) return COIN_EINVAL;
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

extern "C" int coin_roi_align_fwd(const void* feat, int N, int C, int H, int W, int layout, const float* rois,
                                  int R, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                  void* out, int dtype, void* stream) {
  int rc = check_common(feat, rois, out, N, C, H, W, layout, R, ph, pw, dtype);
  if (rc) return rc;
  if (R == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (layout == COIN_NHWC) {
    const int grid = ((R + 7) / 8) * 8 * ph;
    if (dtype == COIN_F32)
      roi_align_fwd_nhwc_kernel<float><<<grid, 256, 0, st>>>((const float*)feat, rois, (float*)out, C, H, W, R, ph,
                                                               pw, spatial_scale, sampling_ratio, aligned);
    else
      roi_align_fwd_nhwc_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)feat, rois, (bf16_t*)out, C, H, W, R,
                                                                ph, pw, spatial_scale, sampling_ratio, aligned);
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

extern "C" int coin_roi_align_bwd(const void* grad_out, int N, int C, int H, int W, int layout, const float* rois,
                                  int R, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                  float* grad_feat, int dtype, void* stream) {
  int rc = check_common(grad_out, rois, grad_feat, N, C, H, W, layout, R, ph, pw, dtype);
  if (rc) return rc;
  if (R == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (layout == COIN_NHWC) {
    dim3 grid(R, (C + BWD_CH - 1) / BWD_CH);
    if (dtype == COIN_F32)
      roi_align_bwd_nhwc_kernel<float><<<grid, 256, 0, st>>>((const float*)grad_out, rois, grad_feat, C, H, W, R, ph,
                                                               pw, spatial_scale, sampling_ratio, aligned);
    else
      roi_align_bwd_nhwc_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)grad_out, rois, grad_feat, C, H, W, R,
                                                                ph, pw, spatial_scale, sampling_ratio, aligned);
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
