// Window attention forward for the Swin-T student of the FPN extension (BASELINE.json configs[4] "MFMA window-attention"; no
// counterpart in /root/reference: SURVEY finding 2, coin_amd/modeling/swin.py).
//
//   out[b][i][h*32 + d] = sum_j softmax_j( scale * q[b][i][h] . k[b][j][h] + bias[h][i][j] + mask[b % nW][i][j] ) * v[b][j][h][d]
//
// for windows of T <= 64 tokens (7 x 7 = 49) and head dimension 32.  One wave per (window, head):
//   * S^T = K Q^T on MFMA 16x16x32 with K = head dimension = ONE k-step: both operands are 16-byte row reads straight from the qkv
//     tensor (no LDS); the accumulator of tile (jt, it) then holds, per lane, 4 consecutive KEYS of one QUERY (lane & 15);
//   * softmax over the keys: 16 values per lane and query tile + two cross-lane steps (lanes l, l^16, l^32 hold the other keys);
//   * O^T = V^T P^T: the probabilities, converted to bf16 in registers, ARE the B operand (the contraction index is permuted the same
//     way on both operands); V^T fragments come from a [64][32] LDS image of V through ds_read_b64_tr_b16;
//   * the accumulator of O^T holds 4 consecutive head channels of one query per lane -> 8-byte stores.
// Rows >= T of q / k / v are read as zeros, bias columns >= T must be <= -1e30 (the host pads bias to [heads][64][64]).
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int WT = 64;   // padded tokens
constexpr int HD = 32;   // head dimension

__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                               const float* __restrict__ mask, bf16_t* __restrict__ out, int nwin, int nW,
                                                               int heads, int T, float scale) {
  __shared__ __attribute__((aligned(16))) char vlds[4][WT * HD * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  const bool live = item < nwin * heads;   // (kept in the barrier-free code below: a dead wave only skips its loads and stores)
  const int b = live ? item / heads : 0, h = live ? item - b * heads : 0;
  const int fr = lane & 15, fq = lane >> 4;
  const size_t tok_stride = (size_t)3 * heads * HD;
  const bf16_t* base = qkv + (size_t)b * T * tok_stride + (size_t)h * HD;

  // ---- V -> LDS image [64 tokens][32 channels] (rows >= T zero): lane = token
  {
    char* vt = vlds[wave];
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 v = z;
      if (live && lane < T) v = *reinterpret_cast<const bf16x8*>(base + (size_t)lane * tok_stride + 2 * heads * HD + c * 8);
      *reinterpret_cast<bf16x8*>(vt + lane * (HD * 2) + c * 16) = v;
    }
  }
  // ---- Q / K fragments: row = tile * 16 + fr, channels 8 fq .. 8 fq + 7
  bf16x8 qf[4], kf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = t * 16 + fr;
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
    qf[t] = kf[t] = z;
    if (live && row < T) {
      qf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * tok_stride + fq * 8);
      kf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * tok_stride + heads * HD + fq * 8);
    }
  }
  // ---- S^T tiles: st[jt][it][r] = score of key jt*16 + fq*4 + r, query it*16 + fr
  f32x4 st[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it) st[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt], qf[it], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  const float* brow = bias + (size_t)h * WT * WT;
  const float* mrow = mask ? mask + (size_t)(b % nW) * WT * WT : nullptr;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int qi = it * 16 + fr;
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(brow + qi * WT + jt * 16 + fq * 4);
      f32x4 s = st[jt][it] * scale + bb;
      if (mrow) s += *reinterpret_cast<const f32x4*>(mrow + qi * WT + jt * 16 + fq * 4);
      st[jt][it] = s;
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[jt][it][r] - mx);
        st[jt][it][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) st[jt][it] *= inv;
  }
  __syncthreads();  // V image complete (every wave wrote its own image; the barrier also orders this wave's LDS writes before its reads)
  // ---- O^T = V^T P^T, keys in two steps of 32: k index 8 fq + e  <->  key (2 s + (e >> 2)) * 16 + fq * 4 + (e & 3)
  f32x4 ot[2][4];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it) ot[dt][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int q4 = (lane >> 2) & 3, p4 = lane & 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 vf[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const char* a0 = vlds[wave] + ((2 * s) * 16 + fq * 4 + q4) * (HD * 2) + (dt * 16 + p4 * 4) * 2;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * HD * 2));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      vf[dt] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      bf16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (bf16_t)st[2 * s][it][r];
        pf[4 + r] = (bf16_t)st[2 * s + 1][it][r];
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ot[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf, ot[dt][it], 0, 0, 0);
    }
  }
  // ---- ot[dt][it][r] = out[query it*16 + fr][channel dt*16 + fq*4 + r]
  if (live) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int qi = it * 16 + fr;
      if (qi < T) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)ot[dt][it][r];
          *reinterpret_cast<bf16x4*>(out + ((size_t)b * T + qi) * heads * HD + h * HD + dt * 16 + fq * 4) = o;
        }
      }
    }
  }
}

}  // namespace

extern "C" int coin_window_attn_fwd(const void* qkv, const float* bias, const float* mask, void* out, int num_windows, int windows_per_image,
                                    int heads, int tokens, int head_dim, float scale, void* stream) {
  if (!qkv || !bias || !out || num_windows < 0 || heads <= 0 || windows_per_image <= 0) return COIN_EINVAL;
  if (head_dim != HD || tokens <= 0 || tokens > WT) return COIN_ESHAPE;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)mask & 15) || ((uintptr_t)out & 7)) return COIN_EALIGN;
  if (num_windows == 0) return COIN_OK;
  const long long items = (long long)num_windows * heads;
  window_attn_fwd_kernel<<<(unsigned)((items + 3) / 4), 256, 0, (hipStream_t)stream>>>((const bf16_t*)qkv, bias, mask, (bf16_t*)out, num_windows,
                                                                                         windows_per_image, heads, tokens, scale);
  return coin_launch_status();
}
