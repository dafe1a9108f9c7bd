// Batched greedy NMS for gfx950 (replaces torchvision.ops.nms under detectron2 find_top_rpn_proposals /
// batched_nms: coin/modeling/proposal_generator/rpn.py:113-115, coin/modeling/roi_heads/fast_rcnn.py:164).
//
// Boxes arrive sorted by descending score.  Kernel 1 builds the upper-triangular suppression bit
// matrix (one 64-bit word per box x 64-box column block).  Kernel 2 resolves it with one workgroup per
// image, two-level: inside a 64-box block the greedy order is resolved on a single 64-bit word per lane
// with wave shuffles (no memory traffic); the rows of the block's boxes are then OR-ed into the running
// "removed" bitmap with independent, coalesced loads.  No host round trip (torchvision resolves the
// matrix on the CPU).
#include "common.h"

namespace {

constexpr int NMS_MAX_WORDS = 256;  // one removed-bitmap word per thread of the scan block: up to 16384 boxes per image

// IoU(a, b) = inter / (area_a + area_b - inter) > thr  (torchvision's nms test), decided without the division wherever that is safe: q = RN(inter / u) lies within half an ulp of inter / u and
// p = RN(thr * u) within half an ulp of thr * u, so inter > p (1 + 2^-21) implies q > thr and inter < p (1 - 2^-21) implies q < thr; only
// the pairs in between (and u = 0: 0 / 0 = NaN, not greater) take the exact quotient.  Same decisions as the division, bit for bit
// (tests/test_kernels_gpu.py holds the kept sets to the oracle's); the fp32 division is ~10 of the ~25 instructions of a pair.
__device__ __forceinline__ bool iou_above(const f32x4 a, const float area_a, const f32x4 b, const float thr) {
  const float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  const float ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
  const float ab = (b[2] - b[0]) * (b[3] - b[1]);
  const float u = area_a + ab - inter;
  const float p = thr * u;
  if (inter > p * 1.00000048f) return true;
  if (inter < p * 0.99999952f) return false;
  return inter / u > thr;
}

// grid: (col_blocks, col_blocks, B); block: 64
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                      int n_max, float thr, unsigned long long* __restrict__ mask,
                                                      int col_blocks) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int n = counts[b];
  const int row = rb * 64 + threadIdx.x;
  __shared__ f32x4 cbox[64];
  const f32x4* __restrict__ bx = reinterpret_cast<const f32x4*>(boxes) + (size_t)b * n_max;
  const int col = cb * 64 + threadIdx.x;
  if (col < n) cbox[threadIdx.x] = bx[col];
  __syncthreads();
  if (row >= n) return;
  const f32x4 me = bx[row];
  const float area_me = (me[2] - me[0]) * (me[3] - me[1]);
  unsigned long long bits = 0ull;
  const int lim = (n - cb * 64) < 64 ? (n - cb * 64) : 64;
  const int start = (rb == cb) ? threadIdx.x + 1 : 0;
  for (int j = start; j < lim; ++j)
    if (iou_above(me, area_me, cbox[j], thr)) bits |= 1ull << j;
  mask[((size_t)b * n_max + row) * col_blocks + cb] = bits;
}

// grid: B; block: 256.  Thread t owns 64-bit word t of the running "removed" bitmap (n_max <= 256*64 = 16384).
// Per 64-box block: wave 0 resolves the greedy order inside the block on the diagonal word with wave shuffles (registers
// only); then every thread ORs the mask rows of the block's boxes into its word.  The 64 row reads of a block do not depend
// on the block's outcome (they are masked by the kept bits afterwards), so the rows -- and the diagonal word -- of block
// b+1 are requested BEFORE block b is resolved and are in registers by the time they are needed: the serial chain per
// block is two barriers, the 64-step shuffle loop and 64 ANDs/ORs, not a memory round trip.  One workgroup per image runs
// alone on its CU, so the ~260 VGPRs of the double-buffered rows cost nothing.
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                       const int* __restrict__ counts, int n_max, int col_blocks,
                                                       int max_keep, int* __restrict__ keep, int* __restrict__ num_keep) {
  __shared__ unsigned long long s_rw, s_kept;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = counts[b];
  const unsigned long long* __restrict__ m = mask + (size_t)b * n_max * col_blocks;
  int* __restrict__ kout = keep + (size_t)b * n_max;
  unsigned long long remv = 0ull;
  int nk = 0;
  const int nblk = (n + 63) >> 6;
  const bool owner = t < col_blocks;

  unsigned long long cur[64], nxt[64];
  unsigned long long diag_cur = 0ull, diag_nxt = 0ull;
  auto fetch = [&](int blk, unsigned long long (&rows)[64], unsigned long long& diag) {
    if (blk < nblk && owner && t > blk) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        int row = blk * 64 + j;
        row = row < n_max ? row : n_max - 1;
        rows[j] = m[(size_t)row * col_blocks + t];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 64; ++j) rows[j] = 0ull;
    }
    const int drow = blk * 64 + lane;
    diag = (blk < nblk && wave == 0 && drow < n) ? m[(size_t)drow * col_blocks + blk] : 0ull;
  };
  fetch(0, nxt, diag_nxt);
  for (int blk = 0; blk < nblk && nk < max_keep; ++blk) {
#pragma unroll
    for (int j = 0; j < 64; ++j) cur[j] = nxt[j];
    diag_cur = diag_nxt;
    fetch(blk + 1, nxt, diag_nxt);  // in flight while this block is resolved
    if (t == blk) s_rw = remv;
    __syncthreads();
    if (wave == 0) {
      // The greedy order inside the block on the SCALAR unit (round 6): the running word and the kept bits are wave-uniform, so the 64
      // dependent steps are s_bitcmp / s_cbranch / s_or on SGPRs, and a diagonal word is fetched (two v_readlane) only for a box that is
      // kept.  Before: 64-bit vector arithmetic per step and two ds_bpermute per box -- ~2 500 cycles of a ~7 000-cycle block.
      const unsigned long long rw0 = s_rw;
      unsigned long long rw = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rw0 >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rw0);
      const int dlo = (int)(unsigned)diag_cur, dhi = (int)(unsigned)(diag_cur >> 32);
      const int row = blk * 64 + lane;
      unsigned long long kept = 0ull;
      const int lim = (n - blk * 64) < 64 ? (n - blk * 64) : 64;
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        if (j < lim && !((rw >> j) & 1ull)) {
          kept |= 1ull << j;
          rw |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, j) << 32) | (unsigned)__builtin_amdgcn_readlane(dlo, j);
        }
      }
      if ((kept >> lane) & 1ull) {
        const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
        if (pos < max_keep) kout[pos] = row;
      }
      if (lane == 0) s_kept = kept;
    }
    __syncthreads();
    const unsigned long long kept = s_kept;
    unsigned long long acc = 0ull;
#pragma unroll
    for (int j = 0; j < 64; ++j) acc |= cur[j] & (0ull - ((kept >> j) & 1ull));
    remv |= acc;
    nk += __popcll(kept);
  }
  if (t == 0) num_keep[b] = nk < max_keep ? nk : max_keep;
}

}  // namespace

extern "C" size_t coin_nms_workspace_bytes(int B, int n_max) {
  const size_t cb = (size_t)(n_max + 63) / 64;
  return (size_t)B * (size_t)n_max * cb * sizeof(unsigned long long);
}

extern "C" int coin_nms_batched(const float* boxes, const int* counts, int B, int n_max, float iou_threshold,
                                int max_keep, void* workspace, int* keep, int* num_keep, void* stream) {
  if (B < 0 || n_max < 0 || max_keep < 0) return COIN_EINVAL;
  if (B == 0 || n_max == 0) return COIN_OK;
  if (!boxes || !counts || !workspace || !keep || !num_keep) return COIN_EINVAL;
  if (n_max > 64 * NMS_MAX_WORDS) return COIN_ESHAPE;
  if ((uintptr_t)boxes & 15) return COIN_EALIGN;
  const int cb = (n_max + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cb, cb, B);
  nms_mask_kernel<<<grid, 64, 0, st>>>(boxes, counts, n_max, iou_threshold, (unsigned long long*)workspace, cb);
  int rc = coin_launch_status();
  if (rc) return rc;
  nms_scan_kernel<<<B, 256, 0, st>>>((const unsigned long long*)workspace, counts, n_max, cb, max_keep, keep, num_keep);
  return coin_launch_status();
}
