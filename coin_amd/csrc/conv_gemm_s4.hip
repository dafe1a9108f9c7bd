// 128 x 128 x 32 bf16 GEMM core for the backbone's small maps (gfx950): the same contraction, operands and epilogues as the persistent
// 256 x 256 core of conv_gemm_p8.hip
//   C[M, N] = A[M, K] . B[N, K]^T   (1x1 = NHWC GEMM, 3x3 / pad 1 = implicit GEMM; coin/modeling/utils.py:77-90,219-241 at the res3 /
//   res4 resolutions: M = 16 600 ... 66 400 pixels, N = 128 ... 1024), bf16 rows + BatchNorm statistics partials + residual add,
// for launches whose 256 x 256 tiling leaves the chip idle: [16 600 x 256] is 65 such tiles for 256 CUs, [16 600 x 1024 x 256] is 260
// tiles of FOUR K-tiles each -- one workgroup per CU (160 KiB of LDS) then pays its pipeline fill, its epilogue and its store drain
// with nothing beside it (round 5: 190-370 TFLOP/s on these shapes, 0.08-0.15 of the peak, 4.4 ms of a 33.7 ms step).
//
// Structure: 4 waves (2 x 2), each 64 x 64 = 4 x 4 `mfma_f32_16x16x32_bf16` accumulators (64 VGPRs); K-tile 32: an operand tile is
// 128 rows x 64 B = 8 KiB, a stage (A | B) 16 KiB, NST stages; the 128 x 256 B epilogue image re-uses the ring.  With two stages a
// workgroup needs 32 KiB of LDS and <= 128 VGPRs: FOUR workgroups per CU (four waves per SIMD), each with its own barrier and its own
// `vmcnt` -- one workgroup's DMA latency, epilogue and store drain run under the others' MFMAs, which is the two-workgroup structure
// DESIGN.md named since round 3, reached through the tile size instead of through a second ring.  The grid is one workgroup per tile
// (x K pieces), dispatched by the hardware: no persistent walk, a CU takes the next tile as soon as one of its four slots is free.
// Operands go HBM / L2 -> LDS by range-checked buffer LDS-DMA (16 B per lane; one instruction = 16 rows x 64 B; rows beyond the
// operand and 3x3 taps outside the image arrive as zeros); 16-byte chunk c of row r sits at chunk c ^ ((-(r >> 2)) & 3) (applied on
// the SOURCE address): conflict-free `ds_read_b128` fragment reads (the round-2 256 x 256 x 32 kernel's image).
// K order = the persistent kernel's (3x3: 64-channel chunk major, tap minor; 32-wide MFMA steps in ascending k), same MFMA
// instruction, same operand order: an output computed whole by this kernel has the SAME BITS as the persistent kernel's.
// Statistics partials are per 128-ROW tile here (pivot = the tile's first row): coin_conv_gemm_stats_tile_rows tells the caller,
// coin_conv_gemm_stats_finalize takes the tile height.
// A tile can be cut along K into `split` pieces (fp32 partials through the caller's workspace in thread-private order, summed in piece
// order by conv_gemm_s4_tail_kernel: bit-reproducible) -- measured without gain on the step's shapes, so the default is whole tiles.
#include <stdlib.h>
#include "common.h"
#include "conv_gemm_p8.h"
#include "conv_gemm_dev.h"

#ifdef COIN_LAB   // lab only (tools/gemm_lab): bit 0 = every DMA reads out of range (zeros, no memory traffic), bit 1 = no MFMA, bit 2 = no epilogue
#define S4_DBG(p, bits) ((p).dbg & (bits))
#else
#define S4_DBG(p, bits) 0
#endif

namespace {

constexpr int SM = 128, SN = 128, SK = 32;
constexpr int S_TILE = 128 * SK * 2;   // 8 KiB: 128 rows x 64 B
constexpr int S_STAGE = 2 * S_TILE;    // A | B
constexpr int S_IMG = 128 * 256;       // epilogue image: 128 rows x 256 B

struct S4Args {
  const bf16_t* A; int lda;
  const bf16_t* B; int ldb;
  bf16_t* C; int ldc;
  const bf16_t* R; int ldr;
  int rp_h, rp_w; unsigned rp_magic_hw, rp_magic_w;   // pooled residual, as P8Args
  int M, N, K, H, W, Cin;
  float* stats; long long stats_rows;
  int tiles_m, tiles_n, split;
  float* slab;
  unsigned a_bytes, b_bytes;
  int dbg;
};

__device__ __forceinline__ bf16x8 s4_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4));
}

// Epilogue of one 128 x 128 tile: accumulators -> bf16 image in LDS -> whole 256-byte rows (+ residual, + statistics of the stored
// values).  Workgroup-wide (256 threads, aligned); `img` = 32 KiB of LDS that nobody reads any more.
template <bool STATS>
__device__ __forceinline__ void s4_epilogue(const S4Args& p, f32x4 (&acc)[4][4], int tm, int tn, char* img, int lane, int wave) {
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
  const int m0 = tm * SM, n0 = tn * SN;
  const int chunk = threadIdx.x & 15, rsub = threadIdx.x >> 4;   // 16 chunks of 8 columns per row; rows rsub, rsub + 16, ...
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wr * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = wc * 8 + j * 2 + (fq >> 1);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[i][j][r];
      *reinterpret_cast<bf16x4*>(img + row * 256 + ((ch ^ (row & 15)) << 4) + (fq & 1) * 8) = o;
    }
  }
  P8_LDS_SYNC();
  const int gcol = n0 + chunk * 8;
  const bool col_ok = gcol < p.N;
  f32x2 s1[4], s2[4], piv[4];
  if (STATS) {
    const bf16x8 pv = *reinterpret_cast<const bf16x8*>(img + (chunk << 4));  // row 0 of the tile: every thread's pivot
    p8_pairs(pv, piv);
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[i] = s2[i] = f32x2{0.f, 0.f};
  }
  const bool has_r = p.R != nullptr;
#pragma unroll
  for (int half = 0; half < 2; ++half) {   // four rows per pass: R rows requested first, the image reads back to back, then add / store / statistics
    bf16x8 v[4], rr[4];
    float rscale[4] = {1.f, 1.f, 1.f, 1.f};
    if (has_r && col_ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int grow = m0 + (half * 4 + q) * 16 + rsub;
        grow = grow < p.M ? grow : p.M - 1;
        size_t rrow = (size_t)grow;
        if (p.rp_w) rrow = p8_pooled_row(grow, p.rp_h, p.rp_w, p.rp_magic_hw, p.rp_magic_w, rscale[q]);
        rr[q] = *reinterpret_cast<const bf16x8*>(p.R + rrow * p.ldr + gcol);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (half * 4 + q) * 16 + rsub;
      v[q] = *reinterpret_cast<const bf16x8*>(img + row * 256 + ((chunk ^ (row & 15)) << 4));
    }
    if (has_r && col_ok) {  // C = bf16(bf16(A.B^T) + R): what two separate launches would store
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x2 a[4], b[4];
        p8_pairs(v[q], a);
        p8_pairs(rr[q], b);
        const f32x2 sc = {rscale[q], rscale[q]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 t = __builtin_elementwise_fma(sc, b[i], a[i]);
          v[q][2 * i] = (bf16_t)t.x;
          v[q][2 * i + 1] = (bf16_t)t.y;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int grow = m0 + (half * 4 + q) * 16 + rsub;
      if (grow < p.M && col_ok) *reinterpret_cast<bf16x8*>(p.C + (size_t)grow * p.ldc + gcol) = v[q];
      if (STATS && (long long)grow < p.stats_rows) {
        f32x2 f[4];
        p8_pairs(v[q], f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 d = f[i] - piv[i];
          s1[i] += d;
          s2[i] = __builtin_elementwise_fma(d, d, s2[i]);
        }
      }
    }
  }
  if (STATS) {
    // threads with equal `chunk`: lanes l, l ^ 16, l ^ 32, l ^ 48 of a wave (vector-ALU swaps), then the 4 waves through the image, fixed order
    float t1[8], t2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      t1[i] = (i & 1) ? s1[i >> 1].y : s1[i >> 1].x;
      t2[i] = (i & 1) ? s2[i >> 1].y : s2[i >> 1].x;
    }
    p8_rows_sum(t1);
    p8_rows_sum(t2);
    P8_LDS_SYNC();   // the image has been consumed
    float* red = reinterpret_cast<float*>(img);  // [4 waves][16 chunks][16] + [128] pivots
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red[(wave * 16 + lane) * 16 + i] = t1[i];
        red[(wave * 16 + lane) * 16 + 8 + i] = t2[i];
      }
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[4 * 16 * 16 + lane * 8 + i] = (i & 1) ? piv[i >> 1].y : piv[i >> 1].x;
      }
    }
    P8_LDS_SYNC();
    if (threadIdx.x < 128 && n0 + (int)threadIdx.x < p.N) {
      const int c = threadIdx.x, ch = c >> 3, ci = c & 7;
      float r1[4], r2[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        r1[w] = red[(w * 16 + ch) * 16 + ci];
        r2[w] = red[(w * 16 + ch) * 16 + 8 + ci];
      }
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a1 += r1[w];
        a2 += r2[w];
      }
      float* __restrict__ part = p.stats + (size_t)tm * 3 * p.N + n0 + c;
      part[0] = red[4 * 16 * 16 + c];
      part[p.N] = a1;
      part[2 * (size_t)p.N] = a2;
    }
  }
}

// `s_waitcnt vmcnt(PER * ahead)`: all but the PER * ahead youngest DMA instructions of this wave have landed (the count is an immediate;
// PER = a wave's DMA instructions per stage: 4 or 8)
template <int PER>
__device__ __forceinline__ void s4_wait_ahead(int ahead) {
  static_assert(PER == 4 || PER == 8, "");
  if (PER == 4) {
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
  } else {
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
  }
}

__device__ __forceinline__ bf16x8 s4_frag128(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// NST = ring depth, BK = K-tile (32: 64-byte operand rows, stage 16 KiB; 64: 128-byte rows, stage 32 KiB).
// BK 32 / NST 2: 32 KiB of LDS -> four workgroups per CU; BK 64 / NST 2: 64 KiB -> two.
template <bool GATHER3, bool STATS, int NST, int BK>
__global__ __launch_bounds__(256, (NST * BK <= 64) ? 4 : ((NST * BK <= 96) ? 3 : ((NST * BK <= 160) ? 2 : 1))) void conv_gemm_s4_kernel(const S4Args p) {
  constexpr int TILE = 128 * BK * 2, STAGE = 2 * TILE;   // operand tile / stage bytes
  constexpr int E = BK / 16;                              // DMA instructions per operand, wave and stage (1 KiB each)
  constexpr int RPI = 1024 / (BK * 2);                    // rows per DMA instruction: 16 (64-byte rows) / 8 (128-byte rows)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order: workgroups b, b + 8, ... share an XCD (one L2) and take neighbouring items -- the pieces of a tile, then the tiles
  // of a row panel (same A rows) -- of a contiguous eighth of the list
  const int nwg = gridDim.x;
  int item;
  {
    const int q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    item = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tile = item / p.split, piece = item - tile * p.split;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int nk = p.K / BK;
  const int kb = (int)((long long)piece * nk / p.split), ke = (int)((long long)(piece + 1) * nk / p.split);

  // ---- this lane's E A rows and E B rows of a stage (one DMA instruction = RPI rows), source chunk swizzled:
  // 64-byte rows: chunk c of row r at c ^ ((-(r >> 2)) & 3);  128-byte rows: at c ^ (r & 7)
  const int rl = BK == 32 ? lane >> 2 : lane >> 3;
  const int sc = (BK == 32 ? ((lane & 3) ^ ((0 - (rl >> 2)) & 3)) : ((lane & 7) ^ rl)) * 8;
  unsigned a_off[E], b_off[E], taps[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int rloc = (wave * E + e) * RPI + rl;
    int row = tm * SM + rloc;
    row = row < p.M ? row : p.M - 1;
    taps[e] = 0;
    if (GATHER3) {
      const int hw = p.H * p.W;
      const int nb = row / hw, rem = row - nb * hw;
      const int oy = rem / p.W, ox = rem - oy * p.W;
      unsigned m = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = oy + t / 3 - 1, xx = ox + t % 3 - 1;
        m |= (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W ? 1u : 0u) << t;
      }
      taps[e] = m;
      a_off[e] = (unsigned)(((size_t)row * p.Cin + sc) * 2);
    } else {
      a_off[e] = (unsigned)(((size_t)row * p.lda + sc) * 2);
    }
    const int brow = tn * SN + rloc;
    b_off[e] = brow < p.N ? (unsigned)(((size_t)brow * p.ldb + sc) * 2) : P8_OOB;
  }

  auto stage = [&](int kt, int slot) {
    char* da = lds + slot * STAGE + wave * (E * 1024);
    char* db = da + TILE;
    int koff = kt * BK;
    if (GATHER3) {
      const int kt64 = BK == 32 ? kt >> 1 : kt, chunk = kt64 / 9, tap = kt64 - chunk * 9;
      const int cofs = chunk * 64 + (BK == 32 ? (kt & 1) * 32 : 0);
      const int shift = (((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.Cin + cofs) * 2;
#pragma unroll
      for (int e = 0; e < E; ++e) blds16(p.A, p.a_bytes, (((taps[e] >> tap) & 1u) && !S4_DBG(p, 1)) ? a_off[e] + (unsigned)shift : P8_OOB, 0, da + e * 1024);
      koff = tap * p.Cin + cofs;
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) blds16(p.A, p.a_bytes, S4_DBG(p, 1) ? P8_OOB : a_off[e], kt * (BK * 2), da + e * 1024);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) blds16(p.B, p.b_bytes, S4_DBG(p, 1) ? P8_OOB : b_off[e], koff * 2, db + e * 1024);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (kb + s < ke) stage(kb + s, s);
  const int fr = lane & 15, fq = lane >> 4;
  int slot = 0;
  for (int kt = kb; kt < ke; ++kt) {
    // stage kt has landed once at most the loads of the `ahead` later stages are outstanding; the barrier publishes every wave's DMA
    // writes and says that the slot re-staged below (read in iteration kt - 1) is no longer being read
    const int left = ke - 1 - kt;
    s4_wait_ahead<2 * E>(left < NST - 2 ? left : NST - 2);
    P8_BAR();
    if (kt + NST - 1 < ke) {
      int ns = slot + NST - 1;
      ns = ns >= NST ? ns - NST : ns;
      stage(kt + NST - 1, ns);
    }
    const char* la = lds + slot * STAGE;
    const char* lb = la + TILE;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = BK == 32 ? s4_frag(lb, wc * 64 + j * 16 + fr, fq) : s4_frag128(lb, wc * 64 + j * 16 + fr, ks * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = BK == 32 ? s4_frag(la, wr * 64 + i * 16 + fr, fq) : s4_frag128(la, wr * 64 + i * 16 + fr, ks * 4 + fq);
      __builtin_amdgcn_s_setprio(1);
      if (!S4_DBG(p, 2)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // swapped: lane = row m, regs = 4 columns n
      } else {
        acc[0][0][0] += (float)af[0][0] + (float)af[1][0] + (float)af[2][0] + (float)af[3][0] + (float)bfr[0][0] + (float)bfr[1][0] + (float)bfr[2][0] + (float)bfr[3][0];
      }
      __builtin_amdgcn_s_setprio(0);
    }
    slot = slot + 1 == NST ? 0 : slot + 1;
  }
  P8_LDS_SYNC();   // every wave is done with the ring: it becomes the epilogue image

  if (S4_DBG(p, 4)) {
    if (acc[0][0][0] == 12345.678f) p.C[0] = (bf16_t)acc[1][1][1];
  } else if (p.split == 1) {
    s4_epilogue<STATS>(p, acc, tm, tn, lds, lane, wave);
  } else {
    // K piece: fp32 accumulators in thread-private order (float4 index q * 256 + thread): coalesced 16-byte stores
    f32x4* __restrict__ sl = reinterpret_cast<f32x4*>(p.slab) + ((size_t)tile * p.split + piece) * (16 * 256) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sl[(i * 4 + j) * 256] = acc[i][j];
  }
}

// Split-K tail: the pieces of one tile summed in piece order (thread-private order, as they were stored) -> the epilogue.  grid = tiles.
template <bool STATS>
__global__ __launch_bounds__(256) void conv_gemm_s4_tail_kernel(const S4Args p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const f32x4* __restrict__ sl = reinterpret_cast<const f32x4*>(p.slab) + (size_t)tile * p.split * (16 * 256) + threadIdx.x;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = sl[(i * 4 + j) * 256];
  for (int s = 1; s < p.split; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += sl[(size_t)s * (16 * 256) + (i * 4 + j) * 256];
  }
  s4_epilogue<STATS>(p, acc, tm, tn, lds, lane, wave);
}

}  // namespace

#ifdef COIN_LAB
int coin_s4_split = -1;   // lab hook: -1 = default policy, >= 1 = this many K pieces per tile
int coin_s4_stages = 0;   // lab hook: 0 = default, 2 / 3 / 5 / 8 = ring depth at K-tile 32; 12 / 13 / 14 = depth 2 / 3 / 4 at K-tile 64
int coin_s4_maxwg = 0;    // lab hook: 0 = default, 1..4 = workgroups per CU (by the size of the LDS request)
int coin_s4_debug = 0;    // lab hook: S4Args::dbg
#else
static constexpr int coin_s4_split = -1, coin_s4_stages = 0, coin_s4_maxwg = 0, coin_s4_debug = 0;
#endif

static int s4_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

bool coin_s4_nt_ok(int M, int N, int K, int mode, int Cin, int lda, int ldb) {
  if (M <= 0 || N <= 0 || N % 8 || K % 64 || K < 64) return false;
  if ((size_t)M * (mode == 1 ? Cin : lda) * 2 >= 0x7f000000ull || (size_t)N * ldb * 2 >= 0x7f000000ull) return false;  // 32-bit buffer offsets
  if (mode == 1 && (Cin % 64 || K != 9 * Cin)) return false;
  return true;
}

// K pieces per tile: never by default.  Measured (tools/gemm_lab sbench, profiles/r6_lab_s4.log): layer3's 3x3 [16 600 x 256 x 2304],
// 260 tiles: whole tiles 33 us, 2 pieces 40 us, 4 pieces 33 us -- what a piece saves in loop time goes into the fp32 round trip of its
// 64 KiB partial tile and the second launch.  The mechanism stays for the lab (coin_s4_split) and for callers with a workspace.
static int s4_split(int ntiles, int nk) {
  if (coin_s4_split >= 1) return coin_s4_split < nk ? coin_s4_split : nk;
  return 1;
}

// Dispatch rule, from tools/gemm_lab sbench on the step's shapes (profiles/r6_lab_s4.log; us, persistent 256 x 256 kernel -> this one):
//   N < 256 (served by the 256 x 128 kernel before): [66 800 x 128 x 1152] 45 -> 29, [266 400 x 128 x 1152] 114 -> 94;
//   fewer 256-tiles than half the CUs: [16 600 x 256 x 2304] 48 -> 33, [16 600 x 256 x 1024] 26 -> 19.6;
//   fewer than three rounds of 256-tiles and K <= 1024: [16 600 x 1024 x 256] 23 -> 17, [66 800 x 512 x 128] 26 -> 19.6,
//     [66 800 x 256 x 512] 35 -> 27, [16 600 x 1024 x 512] 33 -> 26.5;
//   K <= 256 below eight rounds: [266 400 x 256 x 128] 42 -> 34.
// The persistent kernel keeps the long-K launches even with few tiles ([66 800 x 256 x 2304]: 86 against 103 us here, the RPN head's
// [16 600 x 1024 x 9216]: 284 against 403) and everything at res5 size: a 128 x 128 tile moves twice the operand bytes per flop through
// the LDS-DMA path, which saturates near 25 B/clk/CU of real traffic (40 B/clk/CU with every offset out of range, `nomem` in the lab).
bool coin_s4_nt_wanted(int M, int N, int K) {
  const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const int cus = s4_cus();
  return N < 256 || 2 * t256 < cus || (t256 < 3LL * cus && K <= 1024) || (K <= 256 && t256 < 8LL * cus);
}

size_t coin_s4_nt_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K % 64) return 0;
  const int ntiles = ((M + SM - 1) / SM) * ((N + SN - 1) / SN);
  const int s = s4_split(ntiles, K / SK);
  return s > 1 ? (size_t)ntiles * s * 65536 : 0;
}

int coin_s4_nt_launch(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc, const void* R, int ldr,
                      int M, int N, int K, float* stats, long long stats_rows, void* workspace, size_t workspace_bytes, hipStream_t st, int rp_h, int rp_w) {
  S4Args a;
  a.A = (const bf16_t*)A; a.lda = lda;
  a.B = (const bf16_t*)B; a.ldb = ldb;
  a.C = (bf16_t*)C; a.ldc = ldc;
  a.R = (const bf16_t*)R; a.ldr = ldr;
  a.rp_h = R ? rp_h : 0; a.rp_w = R ? rp_w : 0;
  a.rp_magic_hw = a.rp_w ? (unsigned)((0x100000000ull + (unsigned)(rp_h * rp_w) - 1) / (unsigned)(rp_h * rp_w)) : 0;
  a.rp_magic_w = a.rp_w ? (unsigned)((0x100000000ull + (unsigned)rp_w - 1) / (unsigned)rp_w) : 0;
  a.M = M; a.N = N; a.K = K; a.H = H; a.W = W; a.Cin = Cin;
  a.stats = stats; a.stats_rows = stats_rows;
  a.tiles_m = (M + SM - 1) / SM; a.tiles_n = (N + SN - 1) / SN;
  a.a_bytes = (unsigned)((size_t)M * (mode == 1 ? Cin : lda) * 2);
  a.b_bytes = (unsigned)((size_t)N * ldb * 2);
  const int ntiles = a.tiles_m * a.tiles_n;
  a.split = s4_split(ntiles, K / SK);
  if (a.split > 1 && (workspace == nullptr || (size_t)ntiles * a.split * 65536 > workspace_bytes)) a.split = 1;
  a.slab = (float*)workspace;
  a.dbg = coin_s4_debug;
  const int grid = ntiles * a.split;
  // lab builds can choose the ring (coin_s4_stages = 10 * BK/32 + NST ... see below); default: K-tile 32, two stages
  int nst = 2, bk = 32;
  if (coin_s4_stages) { nst = coin_s4_stages % 10; bk = coin_s4_stages >= 10 ? 64 : 32; }
  int lds = nst * bk * 512;
  if (coin_s4_maxwg >= 1) {
    const int want = (160 * 1024 / coin_s4_maxwg) & ~1023;
    lds = want > lds ? want : lds;
  }
#define S4_ONE(G3, ST, NS, KK)                                                                                                               \
  do {                                                                                                                                       \
    static bool attr_set = false;                                                                                                            \
    if (!attr_set) {                                                                                                                         \
      (void)hipFuncSetAttribute((const void*)conv_gemm_s4_kernel<G3, ST, NS, KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      attr_set = true;                                                                                                                       \
    }                                                                                                                                        \
    conv_gemm_s4_kernel<G3, ST, NS, KK><<<grid, 256, lds, st>>>(a);                                                                          \
  } while (0)
#ifdef COIN_LAB
#define S4_LAUNCH(G3, ST)                                  \
  do {                                                     \
    if (bk == 64 && nst == 4) S4_ONE(G3, ST, 4, 64);       \
    else if (bk == 64 && nst == 3) S4_ONE(G3, ST, 3, 64);  \
    else if (bk == 64) S4_ONE(G3, ST, 2, 64);              \
    else if (nst == 8) S4_ONE(G3, ST, 8, 32);              \
    else if (nst == 5) S4_ONE(G3, ST, 5, 32);              \
    else if (nst == 3) S4_ONE(G3, ST, 3, 32);              \
    else S4_ONE(G3, ST, 2, 32);                            \
  } while (0)
#else
#define S4_LAUNCH(G3, ST) S4_ONE(G3, ST, 2, 32)
#endif
  const bool st_main = stats != nullptr && a.split == 1;
  if (mode == 1) {
    if (st_main) S4_LAUNCH(true, true); else S4_LAUNCH(true, false);
  } else {
    if (st_main) S4_LAUNCH(false, true); else S4_LAUNCH(false, false);
  }
#undef S4_LAUNCH
  if (a.split > 1) {
    if (stats) conv_gemm_s4_tail_kernel<true><<<ntiles, 256, S_IMG, st>>>(a);
    else conv_gemm_s4_tail_kernel<false><<<ntiles, 256, S_IMG, st>>>(a);
  }
  return coin_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------- TN (weight gradient)
// dW[co][k] = sum over pixels m of gy[m][co] * xcol[m][k] for the same small maps: output tile 128 co x 128 k (one tap: Cin % 128 == 0),
// K-tile = 32 pixels, 4 waves x (64 x 64), two stages of (G | X) = 2 x 8 KiB: 32 KiB of LDS, up to four workgroups per CU.
// Why a second tile size here: the persistent kernel cuts the pixels over ALL its 256 workgroups and every workgroup writes a whole
// 256 x 256 fp32 partial tile -- 64 MB of partials (written, then read by the reduction) for a [1024 x 256] gradient whose operands
// are 42 MB; with 128 x 128 tiles and `wpc` workgroups per CU the partials are wpc x 16 MB, and a layer with 16 or 36 such tiles cuts the
// pixels 16 or 7 times instead of 64.
// Both operands are pixel-major: a K-tile is [32 pixels][128 channels] = 256-byte rows (one DMA instruction = 4 rows) and the MFMA
// fragments (8 consecutive pixels of one channel per lane) come through ds_read_b64_tr_b16.  16-byte chunk c of row r sits at
// chunk c ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3)): the 8 rows a 32-lane half reads per transposed read (rows q, q + 8: 32 bytes each)
// fall into 8 different 32-byte bank groups.
// Work item = (pixel piece s, tile): pieces are dealt to the XCDs in contiguous ranges, so the workgroups of one XCD walk the same pixels
// at the same time and share the gy / x rows through its L2.  A piece stores its fp32 accumulators in thread-private order;
// conv_wgrad_s4_reduce_kernel sums the pieces of a tile in a fixed order (four interleaved chains joined in chain order: bit-reproducible).
namespace {

struct TsArgs {
  const bf16_t* GY;
  const bf16_t* X;
  float* slab;
  int M, Cout, Cin, Ktot, H, W;
  int tiles_k, ntiles, nkt, S;
  unsigned magic_w;   // ceil(2^32 / W)
  int hw, r32;        // H * W, 32 % hw
  unsigned g_bytes, x_bytes, x_bias;
};

constexpr unsigned TS_OOB = 0x80000000u;
constexpr int TS_TILE = 32 * 256;       // 8 KiB: 32 pixels x 128 channels
constexpr int TS_STAGE = 2 * TS_TILE;   // G | X

__device__ __forceinline__ int ts_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

typedef short ts16x4 __attribute__((ext_vector_type(4)));
typedef short ts16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive pixels (rows r .. r + 3 and r + 4 .. r + 7: 1024 bytes on) of one channel per lane.  Inline asm: the intrinsic makes
// hipcc drain every LDS-DMA in flight (see conv_gemm_p8.hip tn_frag); the consumers sit behind an explicit `s_waitcnt lgkmcnt(0)`.
__device__ __forceinline__ bf16x8 ts_frag(unsigned adr) {
  ts16x4 a, b;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(adr) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(b) : "v"(adr) : "memory");
  const ts16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool GATHER3>
__global__ __launch_bounds__(256, 4) void conv_wgrad_s4_kernel(const TsArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nwg = gridDim.x;
  int item;
  {
    const int q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    item = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int s = item / p.ntiles, tile = item - s * p.ntiles;
  const int tco = tile / p.tiles_k, tk = tile - tco * p.tiles_k;
  const int co0 = tco * 128, k0 = tk * 128;
  const int kb = (int)((long long)s * p.nkt / p.S), ke = (int)((long long)(s + 1) * p.nkt / p.S);
  int dy = 0, dx = 0, cib = k0;
  if (GATHER3) {
    const int tap = k0 / p.Cin;
    cib = k0 - tap * p.Cin;
    dy = tap / 3 - 1;
    dx = tap % 3 - 1;
  }
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.GY), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.X) - p.x_bias), 0, p.x_bytes, 0x00020000);

  // this lane's two rows (pixels) of a K-tile per operand: instruction (wave * 2 + e) covers rows 4 * (wave * 2 + e) .. + 3
  unsigned go[2], xo[2];
  int rem[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = (wave * 2 + e) * 4 + (lane >> 4);
    const int lc = (lane & 15) ^ ts_swz(row);
    const long long m = (long long)kb * 32 + row;
    go[e] = (unsigned)((m * p.Cout + co0 + lc * 8) * 2);
    xo[e] = (unsigned)(((m + dy * p.W + dx) * p.Cin + cib + lc * 8) * 2 + p.x_bias);   // >= 0: the bias covers the largest negative shift
    rem[e] = GATHER3 ? (int)(m % p.hw) : 0;
  }
  auto stage = [&](int slot) {   // stages the K-tile the offsets point at, then advances them by one K-tile
    char* dg = lds + slot * TS_STAGE + wave * 2048;
    char* dxp = dg + TS_TILE;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(dg + e * 1024), 16, go[e], 0, 0, 0);
      unsigned off = xo[e];
      if (GATHER3) {
        const int oy = (int)__umulhi((unsigned)rem[e], p.magic_w);
        const int ox = rem[e] - oy * p.W;
        off = ((unsigned)(oy + dy) < (unsigned)p.H && (unsigned)(ox + dx) < (unsigned)p.W) ? off : TS_OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dxp + e * 1024), 16, off, 0, 0, 0);
      go[e] += 64u * p.Cout;
      xo[e] += 64u * p.Cin;
      if (GATHER3) {
        const int r = rem[e] + p.r32;
        rem[e] = r >= p.hw ? r - p.hw : r;
      }
    }
  };

  // transposed-read addresses inside a stage: lane group g4 = lane >> 4 takes pixels 8 g4 .. 8 g4 + 7; per read the lane supplies the
  // address of row 8 g4 + q4 (+ 4 for the second read), columns 4 p4 .. 4 p4 + 3 of the fragment's 16 channels
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int trow = 8 * g4 + q4, tsw = ts_swz(trow);
  const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)lds;
  unsigned g_adr[4], x_adr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    g_adr[i] = lds0 + trow * 256 + (((wr * 8 + i * 2 + (p4 >> 1)) ^ tsw) << 4) + (p4 & 1) * 8;
    x_adr[i] = lds0 + TS_TILE + trow * 256 + (((wc * 8 + i * 2 + (p4 >> 1)) ^ tsw) << 4) + (p4 & 1) * 8;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (kb < ke) stage(0);
  unsigned sofs = 0;
  for (int kt = kb; kt < ke; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_BAR();
    if (kt + 1 < ke) stage(sofs ? 0 : 1);
    bf16x8 gf[4], xf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = ts_frag(x_adr[j] + sofs);
#pragma unroll
    for (int i = 0; i < 4; ++i) gf[i] = ts_frag(g_adr[i] + sofs);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P8_SCHED();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[j], gf[i], acc[i][j], 0, 0, 0);   // lane = co, regs = 4 consecutive k
    __builtin_amdgcn_s_setprio(0);
    P8_SCHED();
    sofs ^= (unsigned)TS_STAGE;
  }
  // fp32 partial tile in thread-private order (float4 index q * 256 + thread): coalesced 16-byte stores
  f32x4* __restrict__ sl = reinterpret_cast<f32x4*>(p.slab) + ((size_t)s * p.ntiles + tile) * (16 * 256) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[(i * 4 + j) * 256] = acc[i][j];
}

// dW tile = sum of its S partial tiles.  One workgroup per (tile, accumulator q = i * 4 + j, quarter of the 256 threads' slots): its four
// waves take the pieces w, w + 4, ... (8 loads in flight per lane), the four chains are joined in wave order through LDS -- a fixed order.
__global__ __launch_bounds__(256) void conv_wgrad_s4_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Ktot, int tiles_k,
                                                                    int ntiles, int S) {
  __shared__ f32x4 part[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quarter = blockIdx.x & 3, q = (blockIdx.x >> 2) & 15, tile = blockIdx.x >> 6;
  const int tid = quarter * 64 + lane;   // the main kernel's thread whose accumulator q this lane sums
  const f32x4* __restrict__ base = reinterpret_cast<const f32x4*>(slab) + (size_t)tile * (16 * 256) + q * 256 + tid;
  const size_t pstride = (size_t)ntiles * (16 * 256);
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s0 = wave; s0 < S; s0 += 32) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + 4 * u;
      v[u] = s < S ? base[(size_t)s * pstride] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  if (wave) part[wave - 1][lane] = a;
  __syncthreads();
  if (wave == 0) {
    a += part[0][lane];
    a += part[1][lane];
    a += part[2][lane];
    const int mw = tid >> 6, ml = tid & 63;   // the main kernel's wave / lane
    const int wr = mw >> 1, wc = mw & 1, fr = ml & 15, fq = ml >> 4, i = q >> 2, j = q & 3;
    const int tco = tile / tiles_k, tk = tile - tco * tiles_k;
    const int row = tco * 128 + wr * 64 + i * 16 + fr, col = tk * 128 + wc * 64 + j * 16 + fq * 4;
    *reinterpret_cast<f32x4*>(dw + (size_t)row * Ktot + col) = a;
  }
}

}  // namespace

#ifdef COIN_LAB
int coin_s4_tn_wpc = 0;   // lab hook: workgroups per CU the pixel split aims at (0 = default)
#else
static constexpr int coin_s4_tn_wpc = 0;
#endif

bool coin_s4_tn_ok(int M, int Cout, int Cin, int Ktot, int mode) {
  if (M <= 0 || Cout % 128 || Cin % 128) return false;
  if (((size_t)M + 64 + 2048) * (size_t)(Cout > Cin ? Cout : Cin) * 2 >= 0x7f000000ull) return false;  // 32-bit buffer offsets
  return mode == 0 ? Ktot == Cin : Ktot == 9 * Cin;
}

static void ts_plan(int M, int Cout, int Ktot, TsArgs& a) {
  a.tiles_k = Ktot / 128;
  a.ntiles = (Cout / 128) * a.tiles_k;
  a.nkt = (M + 31) / 32;
  const int wpc = coin_s4_tn_wpc > 0 ? coin_s4_tn_wpc : 2;
  int S = (wpc * s4_cus() + a.ntiles - 1) / a.ntiles;
  const int max_s = (a.nkt + 3) / 4;   // at least four K-tiles (128 pixels) per piece
  S = S > max_s ? max_s : S;
  a.S = S < 1 ? 1 : S;
}

// Dispatch rule (tools/gemm_lab wbench, profiles/r6_lab_s4.log): the persistent kernel's fixed cost is its partials -- 256 workgroups x
// 256 KiB = 64 MB written and read back whatever the shape -- so it wins only where the operands are several times that: res5
// ([100 352 x 2048 x 512]: 214 against 281 us here; operands 514 MB).  Below ~160 MB of operands this kernel wins everywhere measured:
// layer3 [16 600 x 256 x 1024] 40 -> 22.5 us, layer2 [66 800 x 128 x 1152] 63 -> 35, [266 400 x 128 x 1152] 178 -> 98, the RPN head's
// [16 600 x 1024 x 9216] 420 -> 336, the box head's [2048 x 1024 x 2048] 35 -> 25.
bool coin_s4_tn_wanted(int M, int Cout, int Cin) { return (long long)M * (Cout + Cin) * 2 <= 160LL << 20; }

size_t coin_s4_tn_workspace_bytes(int M, int Cout, int Ktot) {
  if (M <= 0 || Cout % 128 || Ktot % 128) return 0;
  TsArgs a;
  ts_plan(M, Cout, Ktot, a);
  return (size_t)a.ntiles * a.S * 65536;
}

int coin_s4_tn_launch(const void* GY, const void* X, int mode, int H, int W, int Cin, int M, int Cout, int Ktot, float* dW, void* workspace, hipStream_t st) {
  TsArgs a;
  ts_plan(M, Cout, Ktot, a);
  a.GY = (const bf16_t*)GY; a.X = (const bf16_t*)X; a.slab = (float*)workspace;
  a.M = M; a.Cout = Cout; a.Cin = Cin; a.Ktot = Ktot; a.H = mode ? H : 1; a.W = mode ? W : 1;
  a.magic_w = (unsigned)((0x100000000ull + (unsigned)a.W - 1) / (unsigned)a.W);
  a.hw = a.H * a.W; a.r32 = 32 % a.hw;
  a.x_bias = mode ? (unsigned)(W + 1) * Cin * 2 : 0;
  a.g_bytes = (unsigned)((size_t)M * Cout * 2);
  a.x_bytes = (unsigned)((size_t)M * Cin * 2 + a.x_bias);
  const int grid = a.ntiles * a.S;
  if (mode == 1) conv_wgrad_s4_kernel<true><<<grid, 256, 2 * TS_STAGE, st>>>(a);
  else conv_wgrad_s4_kernel<false><<<grid, 256, 2 * TS_STAGE, st>>>(a);
  conv_wgrad_s4_reduce_kernel<<<a.ntiles * 64, 256, 0, st>>>(a.slab, dW, Cout, Ktot, a.tiles_k, a.ntiles, a.S);
  return coin_launch_status();
}
