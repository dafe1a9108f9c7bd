// Table-driven elementwise kernels: fused SGD(momentum) over all parameter tensors in one launch,
// teacher EMA over a whole state dict in one launch, and uint8 -> normalised/padded image batches.
//
// Reference call sites: coin/solver/build.py:96-103 + coin/engine/pre_train.py:199-202 (torch.optim.SGD
// over ~170 single-tensor param groups under GradScaler), coin/modeling/meta_arch/ts_ensemble.py:39-69
// (state-dict EMA), coin/modeling/meta_arch/clip_rcnn.py:287-298 (ToTensor/Normalize/pad).
// All three are HBM-bound streams: 16-byte accesses, grid-stride, one launch instead of hundreds.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sgd_kernel(const coin_sgd_tensor* __restrict__ table, float momentum,
                                                  float inv_scale, float lr_scale, int first_step, const float* __restrict__ gate) {
  if (gate != nullptr && *gate == 0.f) return;   // device-side "skip this update" (data-parallel CKG step: no rank had merge terms)
  coin_sgd_tensor t = table[blockIdx.y];
  t.lr *= lr_scale;
  const int64_t n = t.numel;
  const int64_t n4 = ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.momentum_buf) & 15) == 0) ? (n >> 2) : 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 p = reinterpret_cast<f32x4*>(t.param)[i];
    const f32x4 g0 = reinterpret_cast<const f32x4*>(t.grad)[i];
    f32x4 m;
    if (first_step) m = (f32x4){0.f, 0.f, 0.f, 0.f};
    else m = reinterpret_cast<f32x4*>(t.momentum_buf)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = g0[j] * inv_scale + t.weight_decay * p[j];
      m[j] = first_step ? g : momentum * m[j] + g;
      p[j] -= t.lr * m[j];
    }
    reinterpret_cast<f32x4*>(t.momentum_buf)[i] = m;
    reinterpret_cast<f32x4*>(t.param)[i] = p;
    if (t.bf16_shadow) {
      bf16x4 s;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = (bf16_t)p[j];
      reinterpret_cast<bf16x4*>(t.bf16_shadow)[i] = s;
    }
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float p = t.param[i];
    const float g = t.grad[i] * inv_scale + t.weight_decay * p;
    const float m = first_step ? g : momentum * t.momentum_buf[i] + g;
    p -= t.lr * m;
    t.momentum_buf[i] = m;
    t.param[i] = p;
    if (t.bf16_shadow) reinterpret_cast<bf16_t*>(t.bf16_shadow)[i] = (bf16_t)p;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(const coin_ema_tensor* __restrict__ table, float keep) {
  const coin_ema_tensor t = table[blockIdx.y];
  const int64_t n = t.numel;
  const int64_t n4 = ((((uintptr_t)t.teacher | (uintptr_t)t.student) & 15) == 0) ? (n >> 2) : 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const float w = 1.0f - keep;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 a = reinterpret_cast<f32x4*>(t.teacher)[i];
    const f32x4 s = reinterpret_cast<const f32x4*>(t.student)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = s[j] * w + a[j] * keep;
    reinterpret_cast<f32x4*>(t.teacher)[i] = a;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    t.teacher[i] = t.student[i] * w + t.teacher[i] * keep;
}

template <typename T>
__global__ __launch_bounds__(256) void normalize_pad_kernel(const uint8_t* __restrict__ img, int h, int w, float m0,
                                                            float m1, float m2, float s0, float s1, float s2,
                                                            T* __restrict__ out, int Hp, int Wp, int layout) {
  const int64_t total = (int64_t)Hp * Wp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / Wp), x = (int)(i - (int64_t)y * Wp);
    float v[3] = {0.f, 0.f, 0.f};
    if (y < h && x < w) {
      const size_t o = (size_t)y * w + x;
      const size_t plane = (size_t)h * w;
      // ToTensor: u8 / 255 ; Normalize: (v - mean) / std     (same operation order as torchvision)
      v[0] = ((float)img[o] / 255.0f - m0) / s0;
      v[1] = ((float)img[plane + o] / 255.0f - m1) / s1;
      v[2] = ((float)img[2 * plane + o] / 255.0f - m2) / s2;
    }
    if (layout == COIN_NCHW) {
      out[i] = (T)v[0];
      out[total + i] = (T)v[1];
      out[2 * total + i] = (T)v[2];
    } else {
      out[i * 3 + 0] = (T)v[0];
      out[i * 3 + 1] = (T)v[1];
      out[i * 3 + 2] = (T)v[2];
    }
  }
}

}  // namespace

extern "C" int coin_abi_version(void) { return COIN_ABI_VERSION; }
extern "C" const char* coin_build_arch(void) { return "gfx950"; }
extern "C" int coin_clear_last_error(void) { return (int)hipGetLastError(); }

extern "C" int coin_sgd_step(const coin_sgd_tensor* table, int num_tensors, int64_t max_numel, float momentum,
                             float inv_loss_scale, float lr_scale, int first_step, const float* gate, void* stream) {
  if (num_tensors < 0 || max_numel < 0 || (num_tensors > 0 && !table)) return COIN_EINVAL;
  if (num_tensors == 0 || max_numel == 0) return COIN_OK;
  if (num_tensors > 65535) return COIN_ESHAPE;
  int64_t gx = (max_numel + 1023) / 1024;
  if (gx > 512) gx = 512;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, (unsigned)num_tensors);
  sgd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(table, momentum, inv_loss_scale, lr_scale, first_step, gate);
  return coin_launch_status();
}

extern "C" int coin_ema_update(const coin_ema_tensor* table, int num_tensors, int64_t max_numel, float keep,
                               void* stream) {
  if (num_tensors < 0 || max_numel < 0 || (num_tensors > 0 && !table)) return COIN_EINVAL;
  if (num_tensors == 0 || max_numel == 0) return COIN_OK;
  if (num_tensors > 65535) return COIN_ESHAPE;
  int64_t gx = (max_numel + 1023) / 1024;
  if (gx > 512) gx = 512;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, (unsigned)num_tensors);
  ema_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(table, keep);
  return coin_launch_status();
}

extern "C" int coin_normalize_pad(const uint8_t* img, int h, int w, const float mean[3], const float std_[3], void* out,
                                  int n, int Hp, int Wp, int layout, int dtype, void* stream) {
  if (!img || !out || !mean || !std_ || h <= 0 || w <= 0 || n < 0 || Hp < h || Wp < w) return COIN_EINVAL;
  if (layout != COIN_NCHW && layout != COIN_NHWC) return COIN_EINVAL;
  if (dtype != COIN_F32 && dtype != COIN_BF16) return COIN_EINVAL;
  const int64_t total = (int64_t)Hp * Wp;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t st = (hipStream_t)stream;
  const size_t img_elems = (size_t)3 * Hp * Wp;
  if (dtype == COIN_F32)
    normalize_pad_kernel<float><<<grid, 256, 0, st>>>(img, h, w, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2],
                                                       (float*)out + (size_t)n * img_elems, Hp, Wp, layout);
  else
    normalize_pad_kernel<bf16_t><<<grid, 256, 0, st>>>(img, h, w, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2],
                                                        (bf16_t*)out + (size_t)n * img_elems, Hp, Wp, layout);
  return coin_launch_status();
}
