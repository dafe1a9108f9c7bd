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


// ---------------------------------------------------------------------------------------------------------------- backward
// d qkv and d bias of the same attention, one wave per workgroup: grid = (window chunks, heads); a wave walks the windows of its chunk
// for ONE head and keeps that head's bias gradient in registers (fp32), so the reduction over the thousands of windows that share a
// bias table needs one 16 KiB partial per wave (summed in chunk order by window_attn_dbias_kernel: no atomics, bit-reproducible).
// Per (window, head), with P recomputed exactly as the forward does (S^T = K Q^T tile layout: lane = query, registers = keys):
//   dP^T = V dO^T           same MFMA form as S^T, operands read straight from global rows
//   dS^T = P^T o (dP^T - rowsum(P o dP))      in registers; rowsum over the keys = 16 values per lane + lanes l^16, l^32
//   dV^T = dO^T P           contraction over the QUERIES: both operands through transposed LDS reads (dO image [64][32], P image
//                           [query][key] written from the registers as bf16)
//   dQ^T = K^T dS^T         the forward's PV form: dS^T registers ARE the B operand (permuted key index), K^T by transposed reads
//   dK^T = Q^T dS           as dV^T with (dO, P) -> (Q, dS)
// 80 MFMA 16x16x32 per (window, head); P and dS enter the products as bf16 like P does in the forward.
constexpr int WB_IMG = WT * HD * 2;        // [64 tokens][32 channels] bf16
constexpr int WB_PIMG = WT * WT * 2;       // [64 queries][64 keys] bf16

__device__ __forceinline__ bf16x8 wa_tr_frag(const char* img, int pitch, int row0, int col0, int fq, int q4, int p4) {
  // k index 8 fq + e  <->  row row0 + (e >> 2) * 16 + fq * 4 + (e & 3); lane (lane & 15) receives column col0 + (lane & 15)
  const char* a0 = img + (row0 + fq * 4 + q4) * pitch + (col0 + p4 * 4) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * pitch));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(64) void window_attn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                              const float* __restrict__ mask, const bf16_t* __restrict__ dout,
                                                              bf16_t* __restrict__ dqkv, float* __restrict__ part, int nwin, int nW, int heads,
                                                              int T, float scale, int wpc) {
  __shared__ __attribute__((aligned(16))) char lds[3 * WB_IMG + WB_PIMG];
  char* const kimg = lds;
  char* const qimg = lds + WB_IMG;
  char* const oimg = lds + 2 * WB_IMG;
  char* const pimg = lds + 3 * WB_IMG;
  const int lane = threadIdx.x;
  const int h = blockIdx.y, chunk = blockIdx.x;
  const int fr = lane & 15, fq = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const size_t tok_stride = (size_t)3 * heads * HD;
  const float* brow = bias + (size_t)h * WT * WT;
  f32x4 dsacc[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it) dsacc[jt][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 zero8;
#pragma unroll
  for (int i = 0; i < 8; ++i) zero8[i] = (bf16_t)0.f;

  const int b_end = (chunk + 1) * wpc < nwin ? (chunk + 1) * wpc : nwin;
  for (int b = chunk * wpc; b < b_end; ++b) {
    const bf16_t* base = qkv + (size_t)b * T * tok_stride + (size_t)h * HD;
    const bf16_t* dob = dout + (size_t)b * T * heads * HD + (size_t)h * HD;
    bf16_t* gb = dqkv + (size_t)b * T * tok_stride + (size_t)h * HD;
    // ---- K, Q, dO images: lane = token (rows >= T zero)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 kq = zero8, qq = zero8, oo = zero8;
      if (lane < T) {
        qq = *reinterpret_cast<const bf16x8*>(base + (size_t)lane * tok_stride + c * 8);
        kq = *reinterpret_cast<const bf16x8*>(base + (size_t)lane * tok_stride + heads * HD + c * 8);
        oo = *reinterpret_cast<const bf16x8*>(dob + (size_t)lane * heads * HD + c * 8);
      }
      *reinterpret_cast<bf16x8*>(kimg + lane * (HD * 2) + c * 16) = kq;
      *reinterpret_cast<bf16x8*>(qimg + lane * (HD * 2) + c * 16) = qq;
      *reinterpret_cast<bf16x8*>(oimg + lane * (HD * 2) + c * 16) = oo;
    }
    // ---- S^T = K Q^T and dP^T = V dO^T from global rows
    f32x4 st[4][4], dpt[4][4];
    {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 16 + fr;
        af[t] = bfr[t] = zero8;
        if (row < T) {
          bfr[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * tok_stride + fq * 8);
          af[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * tok_stride + heads * HD + fq * 8);
        }
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) st[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[jt], bfr[it], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 16 + fr;
        af[t] = bfr[t] = zero8;
        if (row < T) {
          af[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * tok_stride + 2 * heads * HD + fq * 8);
          bfr[t] = *reinterpret_cast<const bf16x8*>(dob + (size_t)row * heads * HD + fq * 8);
        }
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) dpt[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[jt], bfr[it], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // ---- softmax (the forward's arithmetic), dS^T = P^T (dP^T - sum_j P dP), bias gradient
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
          const float pp = __expf(st[jt][it][r] - mx);
          st[jt][it][r] = pp;
          sum += pp;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      float rs = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        st[jt][it] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) rs += st[jt][it][r] * dpt[jt][it][r];
      }
      rs += __shfl_xor(rs, 16, 64);
      rs += __shfl_xor(rs, 32, 64);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dpt[jt][it][r] = st[jt][it][r] * (dpt[jt][it][r] - rs);
        dsacc[jt][it] += dpt[jt][it];
      }
    }
    // ---- P image [query][key] (bf16)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)st[jt][it][r];
        *reinterpret_cast<bf16x4*>(pimg + (it * 16 + fr) * (WT * 2) + (jt * 16 + fq * 4) * 2) = o;
      }
    __syncthreads();
    // ---- dV^T[d][key] = sum over queries dO[q][d] P[q][key]
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[dt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) a[dt] = wa_tr_frag(oimg, HD * 2, 2 * s * 16, dt * 16, fq, q4, p4);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const bf16x8 bb = wa_tr_frag(pimg, WT * 2, 2 * s * 16, jt * 16, fq, q4, p4);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) acc[dt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dt], bb, acc[dt][jt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int kj = jt * 16 + fr;
        if (kj < T) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[dt][jt][r];
            *reinterpret_cast<bf16x4*>(gb + (size_t)kj * tok_stride + 2 * heads * HD + dt * 16 + fq * 4) = o;
          }
        }
      }
    }
    __syncthreads();   // the P image has been read: it becomes the dS image
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)dpt[jt][it][r];
        *reinterpret_cast<bf16x4*>(pimg + (it * 16 + fr) * (WT * 2) + (jt * 16 + fq * 4) * 2) = o;
      }
    __syncthreads();
    // ---- dQ^T[d][query] = scale * sum over keys K[key][d] dS[query][key]   (dS^T registers = B operand, as P in the forward)
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[dt][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) a[dt] = wa_tr_frag(kimg, HD * 2, 2 * s * 16, dt * 16, fq, q4, p4);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          bf16x8 pf;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pf[r] = (bf16_t)dpt[2 * s][it][r];
            pf[4 + r] = (bf16_t)dpt[2 * s + 1][it][r];
          }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dt], pf, acc[dt][it], 0, 0, 0);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int qi = it * 16 + fr;
        if (qi < T) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[dt][it][r] * scale);
            *reinterpret_cast<bf16x4*>(gb + (size_t)qi * tok_stride + dt * 16 + fq * 4) = o;
          }
        }
      }
    }
    // ---- dK^T[d][key] = scale * sum over queries Q[q][d] dS[q][key]
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[dt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) a[dt] = wa_tr_frag(qimg, HD * 2, 2 * s * 16, dt * 16, fq, q4, p4);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const bf16x8 bb = wa_tr_frag(pimg, WT * 2, 2 * s * 16, jt * 16, fq, q4, p4);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) acc[dt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dt], bb, acc[dt][jt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int kj = jt * 16 + fr;
        if (kj < T) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[dt][jt][r] * scale);
            *reinterpret_cast<bf16x4*>(gb + (size_t)kj * tok_stride + heads * HD + dt * 16 + fq * 4) = o;
          }
        }
      }
    }
    __syncthreads();   // the images are rewritten for the next window
  }
  // ---- this wave's bias-gradient partial, register order: float4 index (jt * 4 + it) * 64 + lane
  f32x4* __restrict__ pp = reinterpret_cast<f32x4*>(part) + ((size_t)h * gridDim.x + chunk) * (16 * 64) + lane;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it) pp[(jt * 4 + it) * 64] = dsacc[jt][it];
}

// dbias[h][i][j] (i, j < T) = sum over chunks of the partials, in chunk order.  Thread = one float4 slot of the register order.
__global__ __launch_bounds__(256) void window_attn_dbias_kernel(const float* __restrict__ part, float* __restrict__ dbias, int nchunks, int heads, int T) {
  const int slot = blockIdx.x * 256 + threadIdx.x;   // [heads][16 tiles][64 lanes]
  if (slot >= heads * 1024) return;
  const int h = slot >> 10, rest = slot & 1023, tile = rest >> 6, lane = rest & 63;
  const int jt = tile >> 2, it = tile & 3, fr = lane & 15, fq = lane >> 4;
  const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(part) + (size_t)h * nchunks * 1024 + rest;
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nchunks; ++c) a += src[(size_t)c * 1024];
  const int qi = it * 16 + fr;
  if (qi >= T) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kj = jt * 16 + fq * 4 + r;
    if (kj < T) dbias[((size_t)h * T + qi) * T + kj] = a[r];
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

static int wa_bwd_chunks(int num_windows, int heads) {
  int c = 1024 / (heads > 0 ? heads : 1);
  if (c < 1) c = 1;
  return num_windows < c ? num_windows : c;
}

extern "C" size_t coin_window_attn_bwd_workspace_bytes(int num_windows, int heads) {
  if (num_windows <= 0 || heads <= 0) return 0;
  return (size_t)wa_bwd_chunks(num_windows, heads) * heads * 64 * 64 * sizeof(float);
}

extern "C" int coin_window_attn_bwd(const void* qkv, const float* bias, const float* mask, const void* dout, void* dqkv, float* dbias,
                                    void* workspace, int num_windows, int windows_per_image, int heads, int tokens, int head_dim, float scale,
                                    void* stream) {
  if (!qkv || !bias || !dout || !dqkv || !dbias || !workspace || num_windows < 0 || heads <= 0 || windows_per_image <= 0) return COIN_EINVAL;
  if (head_dim != HD || tokens <= 0 || tokens > WT || heads > 65535) return COIN_ESHAPE;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)mask & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 7) || ((uintptr_t)workspace & 15))
    return COIN_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (num_windows == 0) {
    if (hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)heads * tokens * tokens, st) != hipSuccess) return coin_launch_status();
    return COIN_OK;
  }
  const int chunks = wa_bwd_chunks(num_windows, heads), wpc = (num_windows + chunks - 1) / chunks;
  const int used = (num_windows + wpc - 1) / wpc;   // chunks that own at least one window
  window_attn_bwd_kernel<<<dim3(used, heads), 64, 0, st>>>((const bf16_t*)qkv, bias, mask, (const bf16_t*)dout, (bf16_t*)dqkv, (float*)workspace,
                                                          num_windows, windows_per_image, heads, tokens, scale, wpc);
  window_attn_dbias_kernel<<<(heads * 1024 + 255) / 256, 256, 0, st>>>((const float*)workspace, dbias, used, heads, tokens);
  return coin_launch_status();
}
