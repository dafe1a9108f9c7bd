// Fused loss kernels (forward value + gradient for a unit upstream gradient in one launch).
//
// These tensors are tiny (R x (K+1) with R <= a few thousand, K+1 <= 21): the cost is launch count
// and host syncs, not bytes.  Each loss is ONE single-workgroup launch (1024 threads): rows are
// independent, one lane owns one row, the row reductions over C stay in registers and the loss sum
// is reduced with wavefront shuffles then across the 16 waves through LDS.  One workgroup also
// makes the summation order fixed (bitwise reproducible losses).
//
// Reference call sites: coin/utils/losses.py:13-34 (MILCrossEntropy), coin/modeling/roi_heads/
// fast_rcnn.py:526-545 (KLDivLoss 'mean'), :601-646 (box_reg_loss), :351 (text-align L1),
// coin/modeling/proposal_generator/rpn.py:300-340 (RPN BCE / L1 / KL).
#include "common.h"

namespace {

constexpr int LOSS_THREADS = 1024;
constexpr int MAX_C = 64;

__global__ __launch_bounds__(LOSS_THREADS) void mil_ce_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ target, const int64_t* __restrict__ labels,
    const float* __restrict__ weights, int R, int C, int avg_positives, int reduction_mean, float* __restrict__ loss,
    float* __restrict__ grad_x) {
  __shared__ float red[16];
  const float gscale = reduction_mean ? 1.0f / (float)(R > 0 ? R : 1) : 1.0f;
  float local = 0.f;
  for (int r = threadIdx.x; r < R; r += LOSS_THREADS) {
    const float* __restrict__ xr = x + (size_t)r * ldx;
    // NO max-subtraction: reference quirk (losses.py:15-18)
    float se = 0.f, S = 0.f, T = 0.f;
    const int lab = labels ? (int)labels[r] : -1;
    for (int c = 0; c < C; ++c) {
      const float e = expf(xr[c]);
      const float t = labels ? (c == lab ? 1.f : 0.f) : target[(size_t)r * C + c];
      se += e;
      S += t * e;
      T += t;
    }
    S = S / se;  // sum_c t_c p_c
    const float w = weights ? weights[r] : 1.f;
    const float l = -logf(avg_positives ? S / (T + 1e-6f) : S) * w;
    local += l;
    if (grad_x) {
      for (int c = 0; c < C; ++c) {
        const float p = expf(xr[c]) / se;
        const float t = labels ? (c == lab ? 1.f : 0.f) : target[(size_t)r * C + c];
        grad_x[(size_t)r * C + c] = w * (p - t * p / S) * gscale;
      }
    }
  }
  const float tot = block_reduce_sum(local, red);
  if (threadIdx.x == 0) *loss = tot * gscale;
}

// MILFocalLoss (coin/utils/losses.py:36-73): alpha_r = sum_c t*alpha_c / (T + 1e-6);  P = sum_c t*p / (avg ? T + 1e-6 : 1);
// l_r = -alpha_r (1 - P)^gamma log P [* w_r];  loss = mean_r l_r (R == 0 -> NaN, as torch's mean of an empty tensor) or the sum.
__global__ __launch_bounds__(LOSS_THREADS) void mil_focal_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ target, const int64_t* __restrict__ labels,
    const float* __restrict__ weights, const float* __restrict__ alpha, float gamma, int R, int C, int avg_positives,
    int reduction_mean, float* __restrict__ loss, float* __restrict__ grad_x) {
  __shared__ float red[16];
  const float gscale = reduction_mean ? 1.0f / (float)R : 1.0f;
  float local = 0.f;
  for (int r = threadIdx.x; r < R; r += LOSS_THREADS) {
    const float* __restrict__ xr = x + (size_t)r * ldx;
    float se = 0.f, S = 0.f, T = 0.f, A = 0.f;
    const int lab = labels ? (int)labels[r] : -1;
    for (int c = 0; c < C; ++c) {
      const float e = expf(xr[c]);  // no max-subtraction (losses.py:54-57)
      const float t = labels ? (c == lab ? 1.f : 0.f) : target[(size_t)r * C + c];
      se += e;
      S += t * e;
      T += t;
      A += t * alpha[c];
    }
    S = S / se;
    const float k = avg_positives ? 1.0f / (T + 1e-6f) : 1.0f;
    const float a = A / (T + 1e-6f);
    const float P = S * k;
    const float om = 1.0f - P;
    const float pw = powf(om, gamma);
    const float w = weights ? weights[r] : 1.f;
    local += -a * pw * logf(P) * w;
    if (grad_x) {
      // dl/dP = -a [ -gamma (1-P)^(gamma-1) log P + (1-P)^gamma / P ];  dP/dx_c = k p_c (t_c - S)
      const float dldp = -a * (-gamma * powf(om, gamma - 1.0f) * logf(P) + pw / P) * w;
      for (int c = 0; c < C; ++c) {
        const float p = expf(xr[c]) / se;
        const float t = labels ? (c == lab ? 1.f : 0.f) : target[(size_t)r * C + c];
        grad_x[(size_t)r * C + c] = dldp * k * p * (t - S) * gscale;
      }
    }
  }
  const float tot = block_reduce_sum(local, red);
  if (threadIdx.x == 0) *loss = tot * gscale;
}

// mode 0: logits -> softmax; mode 1: probabilities; mode 2: binary sigmoid
__global__ __launch_bounds__(LOSS_THREADS) void kl_div_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ q, int ldq, const uint8_t* __restrict__ mask,
    int R, int C, int mode, float eps, float* __restrict__ loss, float* __restrict__ grad_x) {
  __shared__ float red[16];
  __shared__ float s_cnt;
  // pass 1: number of selected rows
  float cnt = 0.f;
  for (int r = threadIdx.x; r < R; r += LOSS_THREADS) cnt += (!mask || mask[r]) ? 1.f : 0.f;
  cnt = block_reduce_sum(cnt, red);
  if (threadIdx.x == 0) s_cnt = cnt;
  __syncthreads();
  const float nrow = s_cnt;
  const int cols = mode == 2 ? 2 : C;
  const float inv = nrow > 0.f ? 1.0f / (nrow * (float)cols) : 0.f;
  float local = 0.f;
  for (int r = threadIdx.x; r < R; r += LOSS_THREADS) {
    const bool on = !mask || mask[r];
    if (mode == 2) {
      float g = 0.f;
      if (on) {
        const float xv = x[r];
        const float p1 = 1.0f / (1.0f + expf(-xv));
        const float p0 = 1.0f - p1;
        const float q1 = q[r], q0 = 1.0f - q1;
        if (q1 > 0.f) local += q1 * (logf(q1) - logf(p1 + eps));
        if (q0 > 0.f) local += q0 * (logf(q0) - logf(p0 + eps));
        g = (-q1 / (p1 + eps) + q0 / (p0 + eps)) * p1 * p0 * inv;
      }
      if (grad_x) grad_x[r] = g;
      continue;
    }
    const float* __restrict__ xr = x + (size_t)r * ldx;
    const float* __restrict__ qr = q + (size_t)r * ldq;
    if (!on) {
      if (grad_x)
        for (int c = 0; c < C; ++c) grad_x[(size_t)r * C + c] = 0.f;
      continue;
    }
    if (mode == 0) {
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) m = fmaxf(m, xr[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(xr[c] - m);
      float su = 0.f;
      for (int c = 0; c < C; ++c) {
        const float p = expf(xr[c] - m) / se;
        const float qq = qr[c];
        if (qq > 0.f) local += qq * (logf(qq) - logf(p + eps));
        su += qq * p / (p + eps);
      }
      if (grad_x)
        for (int c = 0; c < C; ++c) {
          const float p = expf(xr[c] - m) / se;
          const float u = qr[c] * p / (p + eps);
          grad_x[(size_t)r * C + c] = (p * su - u) * inv;
        }
    } else {
      for (int c = 0; c < C; ++c) {
        const float p = xr[c];
        const float qq = qr[c];
        if (qq > 0.f) local += qq * (logf(qq) - logf(p + eps));
        if (grad_x) grad_x[(size_t)r * C + c] = -qq / (p + eps) * inv;
      }
    }
  }
  const float tot = block_reduce_sum(local, red);
  if (threadIdx.x == 0) *loss = tot * inv;
}

__device__ __forceinline__ void box_deltas(const float* __restrict__ s, const float* __restrict__ t, float wx, float wy,
                                           float ww, float wh, float (&d)[4]) {
  const float sw = s[2] - s[0], sh = s[3] - s[1];
  const float sx = s[0] + 0.5f * sw, sy = s[1] + 0.5f * sh;
  const float tw = t[2] - t[0], th = t[3] - t[1];
  const float tx = t[0] + 0.5f * tw, ty = t[1] + 0.5f * th;
  d[0] = wx * (tx - sx) / sw;
  d[1] = wy * (ty - sy) / sh;
  d[2] = ww * logf(tw / sw);
  d[3] = wh * logf(th / sh);
}

__global__ __launch_bounds__(LOSS_THREADS) void box_reg_l1_kernel(
    const float* __restrict__ props, const float* __restrict__ gts, const float* __restrict__ pred,
    const int64_t* __restrict__ cls, int R, int nfg, float wx, float wy, float ww, float wh, float normalizer,
    float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ float red[16];
  const float inv = 1.0f / normalizer;
  float local = 0.f;
  for (int r = threadIdx.x; r < R; r += LOSS_THREADS) {
    const int64_t c = cls[r];
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (c >= 0 && c < nfg) {
      float d[4];
      box_deltas(props + (size_t)r * 4, gts + (size_t)r * 4, wx, wy, ww, wh, d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = pred[(size_t)r * 4 + j] - d[j];
        local += fabsf(e);
        g[j] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * inv;
      }
    }
    if (grad) {
#pragma unroll
      for (int j = 0; j < 4; ++j) grad[(size_t)r * 4 + j] = g[j];
    }
  }
  const float tot = block_reduce_sum(local, red);
  if (threadIdx.x == 0) *loss = tot * inv;
}

__global__ __launch_bounds__(LOSS_THREADS) void l1_mean_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               int64_t n, float* __restrict__ loss,
                                                               float* __restrict__ grad) {
  __shared__ float red[16];
  const float inv = 1.0f / (float)(n > 0 ? n : 1);
  float local = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += LOSS_THREADS) {
    const float e = a[i] - b[i];
    local += fabsf(e);
    if (grad) grad[i] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * inv;
  }
  const float tot = block_reduce_sum(local, red);
  if (threadIdx.x == 0) *loss = tot * inv;
}

// Multi-block (A_total ~ 250k): per-block partial sums part[block][2]; rpn_losses_sum_kernel adds them in block order (no float
// atomics: the two totals are bit-reproducible).
__global__ __launch_bounds__(256) void rpn_losses_kernel(
    const float* __restrict__ logits, const int8_t* __restrict__ labels, const float* __restrict__ deltas,
    const float* __restrict__ anchors, const float* __restrict__ gts, int64_t A_total, int64_t A_img, int min_label,
    float* __restrict__ part, float* __restrict__ g_logits,
    float* __restrict__ g_deltas) {
  __shared__ float red[16];
  float lc = 0.f, ll = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A_total; i += (int64_t)gridDim.x * 256) {
    const int lab = labels[i];
    float gl = 0.f;
    if (lab >= min_label) {
      const float xv = logits[i], y = (float)lab;
      lc += fmaxf(xv, 0.f) - xv * y + log1pf(expf(-fabsf(xv)));
      gl = 1.0f / (1.0f + expf(-xv)) - y;
    }
    if (g_logits) g_logits[i] = gl;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (lab == 1) {
      float d[4];
      box_deltas(anchors + (size_t)(i % A_img) * 4, gts + (size_t)i * 4, 1.f, 1.f, 1.f, 1.f, d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = deltas[(size_t)i * 4 + j] - d[j];
        ll += fabsf(e);
        g[j] = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
      }
    }
    if (g_deltas) {
      *reinterpret_cast<f32x4*>(g_deltas + (size_t)i * 4) = (f32x4){g[0], g[1], g[2], g[3]};
    }
  }
  const float tc = block_reduce_sum(lc, red);
  const float tl = block_reduce_sum(ll, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = tc;
    part[2 * blockIdx.x + 1] = tl;
  }
}

// The partials are added in block order by ONE thread per total (the order is the contract: bit-reproducible) -- but they are brought
// into LDS by the whole workgroup first: read one by one from global memory by the adding thread, ~1 000 dependent loads took 88 us
// on the main stream of every step (round 6, tools/chain_gap.py); from LDS the same chain is ~2 us.
__global__ __launch_bounds__(256) void rpn_losses_sum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ loss_cls,
                                                             float* __restrict__ loss_loc) {
  __shared__ float sh[2 * COIN_RPN_LOSS_MAX_BLOCKS];
  for (int i = threadIdx.x; i < 2 * nblk; i += 256) sh[i] = part[i];
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int b = 0; b < nblk; ++b) t += sh[2 * b + threadIdx.x];
    *(threadIdx.x == 0 ? loss_cls : loss_loc) = t;
  }
}

}  // namespace

extern "C" int coin_mil_ce_fwd_bwd(const float* x, int ldx, const float* target, const int64_t* labels,
                                   const float* weights, int R, int C, int avg_positives, int reduction_mean,
                                   float* loss, float* grad_x, void* stream) {
  if (!loss || R < 0 || C <= 0 || C > MAX_C) return C > MAX_C ? COIN_ESHAPE : COIN_EINVAL;
  if (R > 0 && (!x || ldx < C)) return COIN_EINVAL;
  if (R > 0 && ((target == nullptr) == (labels == nullptr))) return COIN_EINVAL;
  mil_ce_kernel<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(x, ldx, target, labels, weights, R, C, avg_positives,
                                                           reduction_mean, loss, grad_x);
  return coin_launch_status();
}

extern "C" int coin_mil_focal_fwd_bwd(const float* x, int ldx, const float* target, const int64_t* labels,
                                      const float* weights, const float* alpha, float gamma, int R, int C, int avg_positives,
                                      int reduction_mean, float* loss, float* grad_x, void* stream) {
  if (!loss || !alpha || R < 0 || C <= 0 || C > MAX_C) return C > MAX_C ? COIN_ESHAPE : COIN_EINVAL;
  if (R > 0 && (!x || ldx < C)) return COIN_EINVAL;
  if (R > 0 && ((target == nullptr) == (labels == nullptr))) return COIN_EINVAL;
  mil_focal_kernel<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(x, ldx, target, labels, weights, alpha, gamma, R, C, avg_positives, reduction_mean, loss, grad_x);
  return coin_launch_status();
}

extern "C" int coin_kl_div_fwd_bwd(const float* x, int ldx, const float* q, int ldq, const uint8_t* row_mask, int R,
                                   int C, int mode, float eps, float* loss, float* grad_x, void* stream) {
  if (!loss || R < 0 || mode < 0 || mode > 2) return COIN_EINVAL;
  if (mode != 2 && (C <= 0 || C > MAX_C)) return C > MAX_C ? COIN_ESHAPE : COIN_EINVAL;
  if (R > 0 && (!x || !q)) return COIN_EINVAL;
  if (R > 0 && mode != 2 && (ldx < C || ldq < C)) return COIN_EINVAL;
  kl_div_kernel<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(x, ldx, q, ldq, row_mask, R, C, mode, eps, loss, grad_x);
  return coin_launch_status();
}

extern "C" int coin_box_reg_l1_fwd_bwd(const float* proposals, const float* gt_boxes, const float* pred_deltas,
                                       const int64_t* gt_classes, int R, int num_fg_classes, float wx, float wy,
                                       float ww, float wh, float normalizer, float* loss, float* grad_deltas,
                                       void* stream) {
  if (!loss || R < 0 || normalizer <= 0.f) return COIN_EINVAL;
  if (R > 0 && (!proposals || !gt_boxes || !pred_deltas || !gt_classes)) return COIN_EINVAL;
  box_reg_l1_kernel<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(proposals, gt_boxes, pred_deltas, gt_classes, R,
                                                               num_fg_classes, wx, wy, ww, wh, normalizer, loss,
                                                               grad_deltas);
  return coin_launch_status();
}

extern "C" int coin_l1_mean_fwd_bwd(const float* a, const float* b, int64_t n, float* loss, float* grad_a, void* stream) {
  if (!loss || n < 0 || (n > 0 && (!a || !b))) return COIN_EINVAL;
  l1_mean_kernel<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(a, b, n, loss, grad_a);
  return coin_launch_status();
}

extern "C" int coin_rpn_losses_fwd_bwd(const float* logits, const int8_t* labels, const float* deltas,
                                       const float* anchors, const float* matched_gt, int64_t A_total,
                                       int64_t A_per_image, int min_label, float* loss_cls, float* loss_loc,
                                       float* grad_logits, float* grad_deltas, void* workspace, void* stream) {
  if (!loss_cls || !loss_loc || A_total < 0 || A_per_image <= 0) return COIN_EINVAL;
  if (A_total > 0 && (!logits || !labels || !deltas || !anchors || !matched_gt || !workspace)) return COIN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int grid = (int)((A_total + 255) / 256);
  if (grid > COIN_RPN_LOSS_MAX_BLOCKS) grid = COIN_RPN_LOSS_MAX_BLOCKS;
  if (grid > 0)
    rpn_losses_kernel<<<grid, 256, 0, st>>>(logits, labels, deltas, anchors, matched_gt, A_total, A_per_image, min_label,
                                            (float*)workspace, grad_logits, grad_deltas);
  rpn_losses_sum_kernel<<<1, 256, 0, st>>>((const float*)workspace, grid, loss_cls, loss_loc);
  return coin_launch_status();
}
