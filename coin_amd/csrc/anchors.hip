// Anchor / proposal labelling on the device: detectron2's Matcher over pairwise IoU, and the fixed-size random fg/bg
// subsampling, each as ONE launch sequence for a whole batch of images.
//
// Replaces, per image and per step, the chain  pairwise_iou (min / max / sub / clamp / mul / where / div over a [G, A] matrix,
// A = 62 250 anchors at 800x1333) -> max over G -> threshold bands -> max over A (low-quality matches) -> == -> any -> where
// of DualTeacherRPN.label_and_sample_anchors (coin/modeling/proposal_generator/rpn.py:120-254, detectron2 Matcher:
// oracle/d2.py) and the  rand -> argsort -> scatter -> rank compare  of the sync-free sampler: ~60 launches of tiny kernels and
// a full 62 250-key sort per image become three launches per batch.
//
// Integer / index work: results are bit-exact with the composed torch ops.  IoU is evaluated with individually rounded fp32
// operations in pairwise_iou's order (no FMA contraction: __fmul_rn / __fadd_rn / __fsub_rn / __fdiv_rn), the arg-max over the
// ground-truth boxes takes the lowest index among equal maxima (torch.max), and the "low-quality" rule compares each IoU with
// the per-box maximum for equality exactly as `(iou == best_per_gt).any(0)` does -- including its quirk that a box overlapping
// no anchor at all (best = 0) marks every anchor with IoU 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

constexpr int AM_THREADS = 256;
constexpr int AM_MAX_IMAGES = 64;
constexpr int AM_MAX_GT = 512;   // ground-truth boxes per image held in LDS

struct AmOffsets {
  int v[AM_MAX_IMAGES + 1];  // image i owns rows [v[i], v[i+1]) of the concatenated ground-truth box list
};

__device__ __forceinline__ float box_area(float4 b) { return __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y)); }

// pairwise_iou (detectron2 structures/boxes.py; coin_amd/structures.py:72-78): a = ground-truth box, b = anchor
__device__ __forceinline__ float iou_rn(float4 a, float area_a, float4 b, float area_b) {
  float w = __fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x));
  float h = __fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y));
  w = w > 0.f ? w : 0.f;
  h = h > 0.f ? h : 0.f;
  const float inter = __fmul_rn(w, h);
  return inter > 0.f ? __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter)) : 0.f;
}

// pass 1: per anchor the best ground-truth box (value, lowest index), the band label, and the per-box maximum over the anchors
__global__ __launch_bounds__(AM_THREADS) void anchor_match_kernel(const float4* __restrict__ gt, AmOffsets off, const float4* __restrict__ anchors,
                                                                  size_t astride /* 0: one shared box set; A: a set per image */, int A, float lo, float hi, int label_lo, int label_mid, int label_hi,
                                                                  int empty_label, int64_t* __restrict__ matched, int8_t* __restrict__ labels,
                                                                  float4* __restrict__ matched_boxes, unsigned* __restrict__ gt_best) {
  __shared__ float4 sbox[AM_MAX_GT];
  __shared__ float sarea[AM_MAX_GT];
  __shared__ unsigned sbest[AM_MAX_GT];
  const int img = blockIdx.y;
  const int g0 = off.v[img], G = off.v[img + 1] - g0;
  for (int g = threadIdx.x; g < G; g += AM_THREADS) {
    const float4 b = gt[g0 + g];
    sbox[g] = b;
    sarea[g] = box_area(b);
    sbest[g] = 0u;
  }
  __syncthreads();
  const int a = blockIdx.x * AM_THREADS + threadIdx.x;
  if (a < A) {
    const size_t o = (size_t)img * A + a;
    if (G == 0) {  // Matcher on an empty IoU matrix: index 0, the lowest band's label (the callers' own rule for such images: empty_label)
      matched[o] = 0;
      labels[o] = (int8_t)empty_label;
      if (matched_boxes) matched_boxes[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float4 b = anchors[(size_t)img * astride + a];
      const float area_b = box_area(b);
      float best = -1.f;
      int arg = 0;
      for (int g = 0; g < G; ++g) {
        const float v = iou_rn(sbox[g], sarea[g], b, area_b);
        if (v > best) {
          best = v;
          arg = g;
        }
        const unsigned u = __float_as_uint(v);  // IoU >= 0: unsigned order == float order
        if (u > sbest[g]) atomicMax(&sbest[g], u);
      }
      int lab = 1;  // Matcher initialises with ones; every finite value falls into one band
      if (best < lo) lab = label_lo;
      else if (best < hi) lab = label_mid;
      else if (best >= hi) lab = label_hi;
      matched[o] = arg;
      labels[o] = (int8_t)lab;
      if (matched_boxes) matched_boxes[o] = sbox[arg];
    }
  }
  __syncthreads();
  if (gt_best)
    for (int g = threadIdx.x; g < G; g += AM_THREADS)
      if (sbest[g]) atomicMax(&gt_best[g0 + g], sbest[g]);
}

// pass 2 (allow_low_quality_matches): label 1 where the anchor's IoU with some box equals that box's maximum over all anchors
__global__ __launch_bounds__(AM_THREADS) void anchor_low_quality_kernel(const float4* __restrict__ gt, AmOffsets off, const float4* __restrict__ anchors,
                                                                        size_t astride, int A, const unsigned* __restrict__ gt_best, int8_t* __restrict__ labels) {
  __shared__ float4 sbox[AM_MAX_GT];
  __shared__ float sarea[AM_MAX_GT];
  __shared__ float sbest[AM_MAX_GT];
  const int img = blockIdx.y;
  const int g0 = off.v[img], G = off.v[img + 1] - g0;
  if (G == 0) return;
  for (int g = threadIdx.x; g < G; g += AM_THREADS) {
    const float4 b = gt[g0 + g];
    sbox[g] = b;
    sarea[g] = box_area(b);
    sbest[g] = __uint_as_float(gt_best[g0 + g]);
  }
  __syncthreads();
  const int a = blockIdx.x * AM_THREADS + threadIdx.x;
  if (a >= A) return;
  const float4 b = anchors[(size_t)img * astride + a];
  const float area_b = box_area(b);
  bool hit = false;
  for (int g = 0; g < G; ++g) hit |= iou_rn(sbox[g], sarea[g], b, area_b) == sbest[g];
  if (hit) labels[(size_t)img * A + a] = 1;
}

// ------------------------------------------------------------------------------------------ subsampling
// One workgroup per image.  cls[m]: -1 ignore, bg_label negative, anything else positive.  Chosen = the k_pos positives and k_neg
// negatives with the smallest (key, index), k_pos = min(#pos, pos_cap), k_neg = min(#neg, num_samples - k_pos): the subsets
// `rank < k` of an ascending stable sort by key selects (coin_amd/box_ops.py:sample_masks), found by a 3-pass radix select on the
// 24-bit keys instead of sorting all M keys.  out[m] = 1 (chosen positive) / 0 (chosen negative) / -1.
constexpr int SS_THREADS = 1024;
constexpr int SS_TIE_CAP = 2048;   // elements of the threshold bin that are ranked exactly (more only with thousands of equal keys: index order)

__device__ int block_sum(int v, int* scratch) {  // all threads get the sum
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  int s = 0;
  for (int w = 0; w < SS_THREADS / 64; ++w) s += scratch[w];
  return s;
}

// exclusive prefix of v over the block in thread order
__device__ int block_excl_scan(int v, int* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += scratch[w];
  return base + inc - v;
}

// The radix select runs on k24 = floor(key * 2^24): uniform 24-bit integers give evenly filled radix bins, and each wave counts into
// its own LDS histogram (a shared histogram of raw float bits serialised ~60 000 atomics on the few exponent bins: 0.79 ms per
// launch, measured).  k24 is monotone in the key but not injective -- the CPU generator's torch.rand draws multiples of 2^-24, the
// device generator's are finer below 0.5 -- so the elements that share the threshold's 24-bit bin are ranked by (raw float bits,
// index) afterwards: the selection is exactly the ascending stable sort by the float key (round-2 ADVICE).
__device__ __forceinline__ unsigned key24(float k) {
  const float v = k * 16777216.0f;
  return v <= 0.f ? 0u : (v >= 16777215.0f ? 16777215u : (unsigned)v);
}

template <typename CLS>
__global__ __launch_bounds__(SS_THREADS) void sample_labels_kernel(const CLS* __restrict__ cls, const float* __restrict__ keys, int M, int bg_label,
                                                                   int num_samples, int pos_cap, int8_t* __restrict__ out) {
  __shared__ int whist[SS_THREADS / 64][256];
  __shared__ int hist[256];
  __shared__ int scratch[SS_THREADS / 64];
  __shared__ unsigned s_prefix;
  __shared__ int s_k;
  __shared__ int s_tidx[SS_TIE_CAP];
  __shared__ unsigned s_tbits[SS_TIE_CAP];
  const int img = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const CLS* c = cls + (size_t)img * M;
  const float* ky = keys + (size_t)img * M;
  int8_t* o = out + (size_t)img * M;
  // thread t visits m = t, t + 1024, ...: every wave access is one contiguous run (the first version gave each thread a contiguous
  // chunk, i.e. 64 different cache lines per wave instruction, and spent 0.5 ms in the texture addresser on 4 CUs)
  int np = 0, nn = 0;
  for (int m = threadIdx.x; m < M; m += SS_THREADS) {
    const int v = (int)c[m];
    np += (v != -1 && v != bg_label);
    nn += (v == bg_label);
    o[m] = -1;
  }
  const int cnt_pos = block_sum(np, scratch), cnt_neg = block_sum(nn, scratch);
  const int k_pos = min(cnt_pos, pos_cap);
  const int k_neg = min(cnt_neg, num_samples - k_pos);
  for (int side = 0; side < 2; ++side) {
    const int k = side == 0 ? k_pos : k_neg, cnt = side == 0 ? cnt_pos : cnt_neg;
    const int8_t mark = side == 0 ? 1 : 0;
    auto member = [&](int m) {
      const int v = (int)c[m];
      return side == 0 ? (v != -1 && v != bg_label) : (v == bg_label);
    };
    if (k <= 0) continue;           // uniform across the block
    if (k >= cnt) {                 // everything of this class is taken
      for (int m = threadIdx.x; m < M; m += SS_THREADS)
        if (member(m)) o[m] = mark;
      continue;
    }
    // radix select: the k-th smallest key (1-based) among the members, most significant byte first
    unsigned prefix = 0, mask = 0;
    int kk = k;
    for (int shift = 16; shift >= 0; shift -= 8) {
      __syncthreads();
      for (int i = lane; i < 256; i += 64) whist[wave][i] = 0;
      __syncthreads();
      for (int m = threadIdx.x; m < M; m += SS_THREADS) {
        if (!member(m)) continue;
        const unsigned u = key24(ky[m]);
        if ((u & mask) == prefix) atomicAdd(&whist[wave][(u >> shift) & 255u], 1);
      }
      __syncthreads();
      if (threadIdx.x < 256) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < SS_THREADS / 64; ++w) t += whist[w][threadIdx.x];
        hist[threadIdx.x] = t;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int acc = 0, d = 0;
        for (; d < 255; ++d) {
          if (acc + hist[d] >= kk) break;
          acc += hist[d];
        }
        s_prefix = prefix | ((unsigned)d << shift);
        s_k = kk - acc;
      }
      __syncthreads();
      prefix = s_prefix;
      kk = s_k;
      mask |= 255u << shift;
    }
    // members with key < T are all taken; of those with key == T the first kk in index order
    const unsigned T = prefix;
    int ties = 0;
    for (int m = threadIdx.x; m < M; m += SS_THREADS)
      if (member(m) && key24(ky[m]) == T) ++ties;
    const int n_ties = block_sum(ties, scratch);
    if (n_ties <= kk) {             // the usual case (one element in the threshold bin): no ranking needed
      for (int m = threadIdx.x; m < M; m += SS_THREADS)
        if (member(m) && key24(ky[m]) <= T) o[m] = mark;
    } else if (n_ties <= SS_TIE_CAP) {
      // several elements in the threshold bin: gather them (index order), rank by (raw key bits, index) -- non-negative floats order
      // as unsigned integers -- and take the first kk
      int base = 0;
      for (int m0 = 0; m0 < M; m0 += SS_THREADS) {
        const int m = m0 + threadIdx.x;
        const bool mem = m < M && member(m);
        const unsigned u = mem ? key24(ky[m]) : 0xffffffffu;
        const int tie = mem && u == T;
        const int pos = base + block_excl_scan(tie, scratch);
        if (tie) {
          s_tidx[pos] = m;
          s_tbits[pos] = __float_as_uint(ky[m]);
        }
        if (mem && u < T) o[m] = mark;
        base += block_sum(tie, scratch);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < n_ties; i += SS_THREADS) {
        const unsigned bi = s_tbits[i];
        int r = 0;
        for (int j = 0; j < n_ties; ++j) {
          const unsigned bj = s_tbits[j];
          r += (bj < bi) || (bj == bi && j < i);
        }
        if (r < kk) o[s_tidx[i]] = mark;
      }
    } else {                        // thousands of keys in the threshold bin (equal keys): rank them in index order, 1024 indices per round
      int base = 0;
      for (int m0 = 0; m0 < M; m0 += SS_THREADS) {
        const int m = m0 + threadIdx.x;
        const bool mem = m < M && member(m);
        const unsigned u = mem ? key24(ky[m]) : 0xffffffffu;
        const int tie = mem && u == T;
        const int rank = base + block_excl_scan(tie, scratch);
        if (mem && (u < T || (tie && rank < kk))) o[m] = mark;
        base += block_sum(tie, scratch);
      }
    }
  }
}

}  // namespace

extern "C" int coin_anchor_match(const float* gt_boxes, const int* gt_offsets_host, int num_images, const float* anchors, int A,
                                 int anchors_per_image, float lo, float hi, int label_lo, int label_mid, int label_hi, int empty_label,
                                 int allow_low_quality, int64_t* matched, int8_t* labels, float* matched_boxes, void* workspace, void* stream) {
  if (!gt_offsets_host || !anchors || !matched || !labels || num_images < 0 || A < 0) return COIN_EINVAL;
  if (num_images > AM_MAX_IMAGES) return COIN_ESHAPE;
  if (num_images == 0 || A == 0) return COIN_OK;
  AmOffsets off;
  for (int i = 0; i <= num_images; ++i) {
    off.v[i] = gt_offsets_host[i];
    if (i && (off.v[i] < off.v[i - 1] || off.v[i] - off.v[i - 1] > AM_MAX_GT)) return COIN_ESHAPE;
  }
  const int total = off.v[num_images];
  if (off.v[0] != 0 || (total > 0 && !gt_boxes) || (allow_low_quality && total > 0 && !workspace)) return COIN_EINVAL;
  if (((uintptr_t)gt_boxes & 15) || ((uintptr_t)anchors & 15) || ((uintptr_t)matched_boxes & 15)) return COIN_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  unsigned* best = allow_low_quality ? (unsigned*)workspace : nullptr;
  if (best && total > 0) {
    const hipError_t e = hipMemsetAsync(best, 0, sizeof(unsigned) * (size_t)total, st);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid((A + AM_THREADS - 1) / AM_THREADS, num_images);
  const size_t astride = anchors_per_image ? (size_t)A : 0;
  anchor_match_kernel<<<grid, AM_THREADS, 0, st>>>((const float4*)gt_boxes, off, (const float4*)anchors, astride, A, lo, hi, label_lo, label_mid, label_hi,
                                                   empty_label, matched, labels, (float4*)matched_boxes, best);
  if (best && total > 0)
    anchor_low_quality_kernel<<<grid, AM_THREADS, 0, st>>>((const float4*)gt_boxes, off, (const float4*)anchors, astride, A, best, labels);
  return coin_launch_status();
}

extern "C" int coin_sample_labels(const void* cls, int cls_is_int64, const float* keys, int num_images, int M, int bg_label, int num_samples,
                                  int pos_cap, int8_t* out, void* stream) {
  if (!cls || !keys || !out || num_images < 0 || M < 0 || num_samples < 0 || pos_cap < 0 || pos_cap > num_samples) return COIN_EINVAL;
  if (num_images == 0 || M == 0) return COIN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (cls_is_int64)
    sample_labels_kernel<int64_t><<<num_images, SS_THREADS, 0, st>>>((const int64_t*)cls, keys, M, bg_label, num_samples, pos_cap, out);
  else
    sample_labels_kernel<int8_t><<<num_images, SS_THREADS, 0, st>>>((const int8_t*)cls, keys, M, bg_label, num_samples, pos_cap, out);
  return coin_launch_status();
}
