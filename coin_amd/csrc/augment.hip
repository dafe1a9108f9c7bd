// Two-view input augmentation on the device: the image operations of COIN's DatasetMapperUnsupervised
// (coin/data/dataset_mapper.py:363-450, coin/data/detection_utils.py:22-45, coin/data/transforms/augmentation_impl.py:64-92) on uint8
// images resident in HBM, so that the 2-worker PIL pipeline of the reference does not have to feed hundreds of views per second.
//
// The reference runs these through Pillow (via torchvision / detectron2).  Every kernel reproduces Pillow's own arithmetic -- integer
// and byte work, bit-exact: 22-bit fixed-point resampling taps computed in double, C-float blends truncated to uint8, the 16.16
// luma weights, the float/double mixture of the RGB<->HSV rows, the 8.24 fixed-point extended box blur -- as restated and pinned
// against Pillow in oracle/augment.py.  Images are [H, W, 3] RGB interleaved (PIL / numpy layout); the last kernel of a chain
// writes [3, H, W] for coin_normalize_pad.  All kernels are byte streams (HBM-bound, a few MB per image).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace {

constexpr int AUG_THREADS = 256;
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(long long v) { return v < 0 ? 0 : (v > 255 ? 255 : (int)v); }

// ------------------------------------------------------------------------------------------ resize (Pillow Resample.c, bilinear)
// One output pixel per thread along the resampled axis; `stride_px_in/out` = pixels between consecutive positions of that axis,
// `line_px_in/out` = pixels between consecutive lines.  Coefficients as precompute_coeffs + normalize_coeffs_8bpc (double).
__global__ __launch_bounds__(AUG_THREADS) void resample_axis_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int in_size, int out_size,
                                                                    int lines, long long stride_in, long long line_in, long long stride_out,
                                                                    long long line_out, int flip_out, int lines_fast) {
  // lines_fast: consecutive threads take consecutive LINES (the vertical pass: neighbouring columns are neighbouring bytes)
  const int t = blockIdx.x * AUG_THREADS + threadIdx.x;
  const int xx = lines_fast ? (int)blockIdx.y : t;
  const int line = lines_fast ? t : (int)blockIdx.y;
  if (xx >= out_size || line >= lines) return;
  double scale = (double)in_size / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double v = (x + xmin - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    ww += v < 1.0 ? 1.0 - v : 0.0;
  }
  long long acc0 = 1 << (PRECISION_BITS - 1), acc1 = acc0, acc2 = acc0;
  const uint8_t* p = src + ((long long)line * line_in + (long long)xmin * stride_in) * 3;
  for (int x = 0; x < xmax; ++x) {
    double v = (x + xmin - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    double k = v < 1.0 ? 1.0 - v : 0.0;
    if (ww != 0.0) k /= ww;
    const double ks = __dmul_rn(k, (double)(1 << PRECISION_BITS));  // no FMA contraction: Pillow rounds the product first
    const long long kk = k < 0 ? (long long)(int)__dadd_rn(-0.5, ks) : (long long)(int)__dadd_rn(0.5, ks);
    acc0 += p[0] * kk;
    acc1 += p[1] * kk;
    acc2 += p[2] * kk;
    p += stride_in * 3;
  }
  const int xo = flip_out ? out_size - 1 - xx : xx;
  uint8_t* q = dst + ((long long)line * line_out + (long long)xo * stride_out) * 3;
  q[0] = (uint8_t)clip8(acc0 >> PRECISION_BITS);
  q[1] = (uint8_t)clip8(acc1 >> PRECISION_BITS);
  q[2] = (uint8_t)clip8(acc2 >> PRECISION_BITS);
}

__global__ __launch_bounds__(AUG_THREADS) void copy_flip_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int flip) {
  const long long i = (long long)blockIdx.x * AUG_THREADS + threadIdx.x;
  if (i >= (long long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long long)y * W);
  const uint8_t* p = src + i * 3;
  uint8_t* q = dst + ((long long)y * W + (flip ? W - 1 - x : x)) * 3;
  q[0] = p[0];
  q[1] = p[1];
  q[2] = p[2];
}

// ------------------------------------------------------------------------------------------ point operations
__device__ __forceinline__ int luma(int r, int g, int b) { return (int)(((unsigned)r * 19595u + (unsigned)g * 38470u + (unsigned)b * 7471u + 0x8000u) >> 16); }

// Image.blend(degenerate, img, alpha): C float arithmetic, no contraction
__device__ __forceinline__ int blend1(int deg, int v, float alpha, bool inside) {
  const float r = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(v - deg)));
  if (inside) return (int)(uint8_t)r;
  return r <= 0.f ? 0 : (r >= 255.f ? 255 : (int)(uint8_t)r);
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) {
    uh = us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)(__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc));
  else h = (float)(__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc));
  h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
  uh = clip8((long long)(int)__dmul_rn((double)h, 255.0));
  us = clip8((long long)(int)__dmul_rn((double)s, 255.0));
}

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const double hh = __ddiv_rn(__dmul_rn((double)h, 6.0), 255.0);
  const double fi = floor(hh);
  const double f = __dsub_rn(hh, fi), fs = __ddiv_rn((double)s, 255.0), dv = (double)v;
  const int p = clip8((long long)floor(__dadd_rn(__dmul_rn(dv, __dsub_rn(1.0, fs)), 0.5)));
  const int q = clip8((long long)floor(__dadd_rn(__dmul_rn(dv, __dsub_rn(1.0, __dmul_rn(fs, f))), 0.5)));
  const int t = clip8((long long)floor(__dadd_rn(__dmul_rn(dv, __dsub_rn(1.0, __dmul_rn(fs, __dsub_rn(1.0, f)))), 0.5)));
  switch (((int)fi) % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// out_chw: write [3, H*W] planes instead of interleaved pixels
__global__ __launch_bounds__(AUG_THREADS) void point_op_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long n, int op, float fparam,
                                                               int iparam, const int* __restrict__ dev_mean, int out_chw) {
  const long long i = (long long)blockIdx.x * AUG_THREADS + threadIdx.x;
  if (i >= n) return;
  int r = src[i * 3], g = src[i * 3 + 1], b = src[i * 3 + 2];
  const bool inside = fparam >= 0.f && fparam <= 1.f;
  switch (op) {
    case COIN_AUG_BRIGHTNESS:
      r = blend1(0, r, fparam, inside); g = blend1(0, g, fparam, inside); b = blend1(0, b, fparam, inside);
      break;
    case COIN_AUG_CONTRAST: {
      const int m = *dev_mean;
      r = blend1(m, r, fparam, inside); g = blend1(m, g, fparam, inside); b = blend1(m, b, fparam, inside);
      break;
    }
    case COIN_AUG_SATURATION: {
      const int l = luma(r, g, b);
      r = blend1(l, r, fparam, inside); g = blend1(l, g, fparam, inside); b = blend1(l, b, fparam, inside);
      break;
    }
    case COIN_AUG_HUE: {
      int h, s, v;
      rgb2hsv(r, g, b, h, s, v);
      hsv2rgb((h + iparam) & 255, s, v, r, g, b);
      break;
    }
    case COIN_AUG_GRAYSCALE:
      r = g = b = luma(r, g, b);
      break;
    case COIN_AUG_SOLARIZE:
      r = r < iparam ? r : 255 - r; g = g < iparam ? g : 255 - g; b = b < iparam ? b : 255 - b;
      break;
    default:  // COIN_AUG_COPY
      break;
  }
  if (out_chw) {
    dst[i] = (uint8_t)r;
    dst[n + i] = (uint8_t)g;
    dst[2 * n + i] = (uint8_t)b;
  } else {
    dst[i * 3] = (uint8_t)r;
    dst[i * 3 + 1] = (uint8_t)g;
    dst[i * 3 + 2] = (uint8_t)b;
  }
}

// sum of the luma over the image -> mean = int(sum / n + 0.5) (Python float division of exact integers), left in device memory
__global__ __launch_bounds__(AUG_THREADS) void luma_sum_kernel(const uint8_t* __restrict__ src, long long n, unsigned long long* __restrict__ sum) {
  __shared__ unsigned long long red[AUG_THREADS / 64];
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * AUG_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * AUG_THREADS)
    s += (unsigned)luma(src[i * 3], src[i * 3 + 1], src[i * 3 + 2]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < AUG_THREADS / 64; ++w) t += red[w];
    atomicAdd(sum, t);  // integer: exact in any order
  }
}

__global__ void luma_mean_kernel(const unsigned long long* __restrict__ sum, long long n, int* __restrict__ mean) {
  *mean = (int)(__dadd_rn(__ddiv_rn((double)*sum, (double)n), 0.5));
}

// ------------------------------------------------------------------------------------------ box blur (Pillow BoxBlur.c)
// One pass along an axis: out[x] = (ww * sum_{|d| <= r} in[clamp(x + d)] + fw * (in[clamp(x - r - 1)] + in[clamp(x + r + 1)]) + 2^23) >> 24,
// uint32 arithmetic (the closed form of ImagingLineBoxBlur32's running sums).
__global__ __launch_bounds__(AUG_THREADS) void box_blur_axis_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int size, int lines,
                                                                    long long stride, long long line_stride, int radius, unsigned ww, unsigned fw,
                                                                    int lines_fast) {
  const int t = blockIdx.x * AUG_THREADS + threadIdx.x;
  const int x = lines_fast ? (int)blockIdx.y : t;
  const int line = lines_fast ? t : (int)blockIdx.y;
  if (x >= size || line >= lines) return;
  const uint8_t* base = src + (long long)line * line_stride * 3;
  unsigned a0 = 0, a1 = 0, a2 = 0;
  for (int d = -radius; d <= radius; ++d) {
    int xi = x + d;
    xi = xi < 0 ? 0 : (xi > size - 1 ? size - 1 : xi);
    const uint8_t* p = base + (long long)xi * stride * 3;
    a0 += p[0];
    a1 += p[1];
    a2 += p[2];
  }
  int xl = x - radius - 1, xr = x + radius + 1;
  xl = xl < 0 ? 0 : xl;
  xr = xr > size - 1 ? size - 1 : xr;
  const uint8_t* pl = base + (long long)xl * stride * 3;
  const uint8_t* pr = base + (long long)xr * stride * 3;
  uint8_t* q = dst + ((long long)line * line_stride + (long long)x * stride) * 3;
  q[0] = (uint8_t)((a0 * ww + (unsigned)(pl[0] + pr[0]) * fw + (1u << 23)) >> 24);
  q[1] = (uint8_t)((a1 * ww + (unsigned)(pl[1] + pr[1]) * fw + (1u << 23)) >> 24);
  q[2] = (uint8_t)((a2 * ww + (unsigned)(pl[2] + pr[2]) * fw + (1u << 23)) >> 24);
}

// _gaussian_blur_radius (float arithmetic, sqrt / floor in double), evaluated on the host
float gaussian_box_radius(float radius, int passes) {
  volatile float sigma2 = radius * radius;
  sigma2 = sigma2 / (float)passes;
  volatile float L = (float)sqrt(12.0 * (double)sigma2 + 1.0);
  volatile float l = (float)floor(((double)L - 1.0) / 2.0);
  volatile float t1 = 2.0f * l;
  t1 = t1 + 1.0f;
  volatile float t2 = l + 1.0f;
  volatile float t3 = l * t2;
  volatile float t4 = 3.0f * sigma2;
  t3 = t3 - t4;
  volatile float a = t1 * t3;
  volatile float t5 = t2 * t2;
  t5 = sigma2 - t5;
  t5 = 6.0f * t5;
  a = a / t5;
  return l + a;
}

}  // namespace

extern "C" int coin_aug_resize_bilinear_u8(const uint8_t* src, int H, int W, uint8_t* dst, int out_h, int out_w, int flip_h, uint8_t* tmp,
                                           void* stream) {
  if (!src || !dst || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return COIN_EINVAL;
  if (out_w != W && out_h != H && !tmp) return COIN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (out_w == W && out_h == H) {
    copy_flip_kernel<<<(unsigned)(((long long)H * W + AUG_THREADS - 1) / AUG_THREADS), AUG_THREADS, 0, st>>>(src, dst, H, W, flip_h);
    return coin_launch_status();
  }
  const uint8_t* cur = src;
  if (out_w != W) {  // horizontal pass: lines = rows
    uint8_t* o = out_h != H ? tmp : dst;
    dim3 grid((out_w + AUG_THREADS - 1) / AUG_THREADS, H);
    resample_axis_kernel<<<grid, AUG_THREADS, 0, st>>>(cur, o, W, out_w, H, 1, W, 1, out_w, out_h != H ? 0 : flip_h, 0);
    cur = o;
  }
  if (out_h != H) {  // vertical pass: lines = columns; the flip is applied on the way out
    dim3 grid((out_w + AUG_THREADS - 1) / AUG_THREADS, out_h);
    if (!flip_h) {
      resample_axis_kernel<<<grid, AUG_THREADS, 0, st>>>(cur, dst, H, out_h, out_w, out_w, 1, out_w, 1, 0, 1);
    } else {  // column c of the source lands in column out_w - 1 - c: shift the destination base, walk the lines backwards
      resample_axis_kernel<<<grid, AUG_THREADS, 0, st>>>(cur, dst + (long long)(out_w - 1) * 3, H, out_h, out_w, out_w, 1, out_w, -1, 0, 1);
    }
  }
  return coin_launch_status();
}

extern "C" int coin_aug_point_op_u8(const uint8_t* src, uint8_t* dst, int H, int W, int op, float fparam, int iparam, void* workspace,
                                    int out_chw, void* stream) {
  if (!src || !dst || H <= 0 || W <= 0 || op < COIN_AUG_COPY || op > COIN_AUG_SOLARIZE) return COIN_EINVAL;
  if (op == COIN_AUG_CONTRAST && (!workspace || ((uintptr_t)workspace & 7))) return COIN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)H * W;
  const int* mean = nullptr;
  if (op == COIN_AUG_CONTRAST) {
    unsigned long long* sum = (unsigned long long*)workspace;
    int* m = (int*)(sum + 1);
    const hipError_t e = hipMemsetAsync(sum, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    long long g = (n + AUG_THREADS - 1) / AUG_THREADS;
    luma_sum_kernel<<<(unsigned)(g > 1024 ? 1024 : g), AUG_THREADS, 0, st>>>(src, n, sum);
    luma_mean_kernel<<<1, 1, 0, st>>>(sum, n, m);
    mean = m;
  }
  point_op_kernel<<<(unsigned)((n + AUG_THREADS - 1) / AUG_THREADS), AUG_THREADS, 0, st>>>(src, dst, n, op, fparam, iparam, mean, out_chw);
  return coin_launch_status();
}

extern "C" int coin_aug_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, int H, int W, float radius, uint8_t* tmp, void* stream) {
  if (!src || !dst || !tmp || H <= 0 || W <= 0 || !(radius >= 0.f) || src == dst) return COIN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int passes = 3;
  const float fr = gaussian_box_radius(radius, passes);
  if (fr == 0.f) {
    copy_flip_kernel<<<(unsigned)(((long long)H * W + AUG_THREADS - 1) / AUG_THREADS), AUG_THREADS, 0, st>>>(src, dst, H, W, 0);
    return coin_launch_status();
  }
  const int r = (int)fr;
  volatile float den = fr * 2.0f;
  den = den + 1.0f;
  volatile float q = (float)(1 << 24) / den;
  const unsigned ww = (unsigned)q;
  const unsigned fw = ((1u << 24) - (unsigned)(r * 2 + 1) * ww) / 2;
  // 3 horizontal passes src -> tmp -> dst -> tmp, then 3 vertical passes tmp -> dst -> tmp -> dst
  dim3 gh((W + AUG_THREADS - 1) / AUG_THREADS, H), gv((W + AUG_THREADS - 1) / AUG_THREADS, H);
  box_blur_axis_kernel<<<gh, AUG_THREADS, 0, st>>>(src, tmp, W, H, 1, W, r, ww, fw, 0);
  box_blur_axis_kernel<<<gh, AUG_THREADS, 0, st>>>(tmp, dst, W, H, 1, W, r, ww, fw, 0);
  box_blur_axis_kernel<<<gh, AUG_THREADS, 0, st>>>(dst, tmp, W, H, 1, W, r, ww, fw, 0);
  box_blur_axis_kernel<<<gv, AUG_THREADS, 0, st>>>(tmp, dst, H, W, W, 1, r, ww, fw, 1);
  box_blur_axis_kernel<<<gv, AUG_THREADS, 0, st>>>(dst, tmp, H, W, W, 1, r, ww, fw, 1);
  box_blur_axis_kernel<<<gv, AUG_THREADS, 0, st>>>(tmp, dst, H, W, W, 1, r, ww, fw, 1);
  return coin_launch_status();
}
