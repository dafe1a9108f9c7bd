// Persistent 256 x 256 x 64 bf16 GEMM cores for the res5 convolutions on the RoI tiles (gfx950):
//   p8 "NT":  C[M, N] = A[M, K] . B[N, K]^T   forward + data gradient of coin/modeling/utils.py:77-90,184-186 (1x1 = NHWC GEMM,
//             3x3 / pad 1 = implicit GEMM), bf16 output rows + BatchNorm statistics partials + residual add in the epilogue;
//   p8 "TN":  dW[Cout, Ktot] = gy[M, Cout]^T . xcol[M, Ktot]   the weight gradient of the same convolutions (fp32 output).
// Entry points: coin_conv_gemm_bf16 / coin_conv_wgrad_bf16 in conv_gemm.hip dispatch here.
//
// Schedule (both): 8 waves = 2 groups of 4 (G0 = waves 0-3, G1 = waves 4-7; one wave of each group per SIMD).  A K-tile is 64 deep
// and consists of four 16 KiB half-tiles (A-lo, A-hi, B-lo, B-hi: 128 rows of the operand each); LDS holds two K-tiles (128 KiB)
// plus a 32 KiB epilogue image.  Every wave owns a 128 x 64 piece of the output made of four 64 x 32 quadrants, one from each
// (A half, B half) pair, so that every wave reads every half-tile exactly once per K-tile:
//   phase 1: read b0 (4 x ds_read_b128) + a0 (8)   -> 16 MFMA  acc[0][0] += a0 . b0
//   phase 2: read b1 (4)                           -> 16 MFMA  acc[0][1] += a0 . b1
//   phase 3: read a1 (8)                           -> 16 MFMA  acc[1][1] += a1 . b1
//   phase 4: -                                     -> 16 MFMA  acc[1][0] += a1 . b0
// Each phase is  {reads; ONE half-tile of LDS-DMA (2 x global_load_lds_dwordx4 per lane); s_barrier; MFMAs; s_barrier}, and G1 runs
// one barrier behind G0: while one group's MFMAs occupy the matrix pipes, the other group's LDS reads and DMA issue run beside them.
// The DMA stream is never drained inside the loop: phase p of K-tile g issues  p=1: A-hi(g+1), p=2: B-lo(g+2), p=3: A-lo(g+2),
// p=4: B-hi(g+2); each half-tile is waited for in the phase before its first read with `s_waitcnt vmcnt(10)` (five half-tiles = 80 KiB
// stay in flight; a load has at least five phases to land.  Round 3 measurement: with one `vmcnt(6)` per K-tile the wait came three
// phases after the issue and the 1x1 layers lost a quarter of their rate to memory latency).
// LDS hazards: a half-tile is re-staged two phases after the phase that read it (one phase after for B-lo, whose reads are retired by
// an `lgkmcnt(8)` ahead of phase 1's first barrier); a staged half-tile is first read in the phase after the barrier that follows
// its wait.  The kernels are persistent (one workgroup per CU) and the K-tile sequence runs across output tiles: the first K-tile of
// the next tile has landed when the epilogue of the current one starts.
#include <stdlib.h>
#include "common.h"
#include "conv_gemm_p8.h"
#include "conv_gemm_dev.h"

// Lab switches (tools/gemm_lab: zero-page DMA, epilogue off, s_memtime stamps, split-K / stagger overrides) exist only in objects
// compiled with -DCOIN_LAB (tools/build_lab.sh); in the product library every P8_DBG() below is the constant 0 and no switch is linked.
#ifdef COIN_LAB
#define P8_DBG(p, bits) ((p).dbg & (bits))
// (coin_p8_debug bit 6): per workgroup [sum of main-loop cycles, sum of epilogue cycles, tiles]
__device__ long long coin_p8_stamp_buf[1024 * 4];
#else
#define P8_DBG(p, bits) 0
#endif

namespace {

constexpr int PM = 256, PN = 256, PK = 64;
constexpr int P_HALF = 128 * PK * 2;   // 16 KiB
constexpr int P_KT = 4 * P_HALF;       // A-lo | A-hi | B-lo | B-hi
constexpr int P_IMG = 2 * P_KT;        // epilogue image: 128 rows x 256 B
constexpr int P_LDS = P_IMG + 128 * 256;

// XCD-aware order of the persistent walk: logical item `it` (block b = it % G in round it / G) -> position in the tile list such
// that the blocks of one XCD (equal b % 8) work on neighbouring positions (which share operand panels) in every round.
__device__ __forceinline__ int p8_remap(int it, int G, int ntiles) {
  const int r = it / G, b = it - r * G;
  const int left = ntiles - r * G, S = left < G ? left : G;
  const int x = b & 7, i = b >> 3, q = S >> 3, rr = S & 7;
  return r * G + (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
}

struct P8Args {
  const bf16_t* A; int lda;
  const bf16_t* B; int ldb;
  bf16_t* C; int ldc;
  const bf16_t* R; int ldr;
  // rp_w > 0: R is the gradient of a 2x2 average pool of the OUTPUT grid (rows = pixels of a [*, rp_h, rp_w] map): output pixel (n, h, w)
  // adds 0.25 * R[n][h / 2][w / 2] (nothing where h / 2 or w / 2 falls off the floor-pooled map) -- avg-pool backward folded into the add
  int rp_h, rp_w; unsigned rp_magic_hw, rp_magic_w;
  int M, N, K, H, W, Cin;
  float* stats; long long stats_rows;
  int tiles_m, tiles_n;
  // Split-K tail: tiles [0, whole_tiles) are computed whole (whole_tiles is a multiple of the grid when split > 1); each of the
  // remaining `rem` tiles is cut along K into `split` pieces so that the last, partly filled round occupies every CU for 1/split of a
  // tile's time (3136 tiles on 256 CUs are 12.25 rounds: 13 are paid without this).  A piece stores its fp32 accumulators to
  // slab[(tile - whole_tiles) * split + piece] and conv_gemm_p8_tail_kernel sums a tile's pieces in piece order and runs the epilogue.
  float* slab; int whole_tiles, rem, split;
  unsigned a_bytes, b_bytes;  // operand sizes for the range-checked DMA
  int stagger_first, stagger_phases, stagger_ticks;   // see the kernel entry (0 phases: off)
  int dbg;  // lab only: bit 0 = every A / B DMA reads the zero page, bit 2 = no epilogue, bit 4 = non-temporal output stores
};

// Work items of workgroup b (grid G): its whole tiles b, b + G, ... (n_whole of them) first, then at most one split-K piece.
// Everything here is wave-uniform scalar arithmetic.
struct P8Items {
  int b, G, nk, n_whole, n_items, piece, whole_tiles, rem, split;
  __device__ __forceinline__ int tile(int idx) const { return idx < n_whole ? p8_remap(b + idx * G, G, whole_tiles) : whole_tiles + b % rem; }
  __device__ __forceinline__ int kb(int idx) const { return idx < n_whole ? 0 : piece * nk / split; }
  __device__ __forceinline__ int ke(int idx) const { return idx < n_whole ? nk : (piece + 1) * nk / split; }
  __device__ __forceinline__ bool whole(int idx) const { return idx < n_whole; }
};

__device__ __forceinline__ P8Items p8_items(const P8Args& p, int b, int G, int nk) {
  P8Items w;
  w.b = b; w.G = G; w.nk = nk; w.whole_tiles = p.whole_tiles; w.rem = p.rem > 0 ? p.rem : 1; w.split = p.split > 0 ? p.split : 1;
  w.n_whole = b < p.whole_tiles ? (p.whole_tiles - 1 - b) / G + 1 : 0;
  const bool has_piece = p.split > 1 && b < p.rem * p.split;
  w.piece = has_piece ? b / w.rem : 0;
  w.n_items = w.n_whole + (has_piece ? 1 : 0);
  return w;
}

// ---------------------------------------------------------------------------------------------------------------- NT kernel
// LDS image of a half-tile: 128 rows x 128 B; one DMA instruction writes 8 rows; 16-byte chunk c of row r sits at chunk c ^ (r & 7)
// (applied on the SOURCE address; the fragment reads apply the same XOR): conflict-free ds_read_b128 for the 16x16x32 operands.

template <bool GATHER3>
struct NtCursor {
  unsigned a[4];  // [A-lo e0, A-lo e1, A-hi e0, A-hi e1]: byte offset of this lane's source row (+ swizzled chunk); 3x3: the centre pixel
  unsigned b[4];
  unsigned taps[4];
  int kt, ke, idx, buf;
};

// N % 256 == 128 (layer2's 128-channel convolutions): the last column tile has no upper B half.  Its DMA offsets lie beyond the operand
// (the range check returns zeros, no memory traffic), its two MFMA phases are skipped and the epilogue stores the lower 128 columns only.
__device__ __forceinline__ bool p8_has_bhi(const P8Args& p, int tile) {
  const int tn = tile % p.tiles_n;
  return tn * PN + 128 < p.N;
}

template <bool GATHER3>
__device__ __forceinline__ void nt_set_tile(NtCursor<GATHER3>& c, const P8Args& p, int tile, int wave, int lane) {
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int rl = lane >> 3, sc = ((lane & 7) ^ rl) * 8;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rloc = h * 128 + (wave * 2 + e) * 8 + rl;
      int row = tm * PM + rloc;
      row = row < p.M ? row : p.M - 1;
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
        c.taps[h * 2 + e] = m;
        c.a[h * 2 + e] = (unsigned)(((size_t)row * p.Cin + sc) * 2);
      } else {
        c.taps[h * 2 + e] = 0;
        c.a[h * 2 + e] = (unsigned)(((size_t)row * p.lda + sc) * 2);
      }
      c.b[h * 2 + e] = (unsigned)(((size_t)(tn * PN + rloc) * p.ldb + sc) * 2);  // rows >= N (N % 256 == 128): beyond b_bytes -> zeros
    }
}

// which: 0 = B-lo, 1 = A-lo, 2 = B-hi, 3 = A-hi of the cursor's K-tile.  Range-checked buffer LDS-DMA: 32-bit per-lane offsets, the
// K offset of the 1x1 case and of B rides in the scalar offset; 3x3 taps outside the image get an out-of-range offset -> zeros.
template <bool GATHER3, int WHICH>
__device__ __forceinline__ void nt_stage(const NtCursor<GATHER3>& c, const P8Args& p, char* lds, int wave) {
  constexpr bool IS_A = (WHICH & 1) != 0;
  constexpr int H = WHICH >> 1;
  char* dst = lds + c.buf * P_KT + (IS_A ? 0 : 2 * P_HALF) + H * P_HALF + wave * 2048;
  if (IS_A) {
    if (GATHER3) {
      const int chunk = c.kt / 9, tap = c.kt - chunk * 9;
      const int shift = (((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.Cin + chunk * PK) * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        unsigned off = ((c.taps[H * 2 + e] >> tap) & 1u) ? c.a[H * 2 + e] + (unsigned)shift : P8_OOB;
        if (P8_DBG(p, 1)) off = P8_OOB;
        blds16(p.A, p.a_bytes, off, 0, dst + e * 1024);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        blds16(p.A, p.a_bytes, (P8_DBG(p, 1)) ? P8_OOB : c.a[H * 2 + e], c.kt * (PK * 2), dst + e * 1024);
    }
  } else {
    int koff = c.kt * PK;
    if (GATHER3) {
      const int chunk = c.kt / 9, tap = c.kt - chunk * 9;
      koff = tap * p.Cin + chunk * PK;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
      blds16(p.B, p.b_bytes, (P8_DBG(p, 1)) ? P8_OOB : c.b[H * 2 + e], koff * 2, dst + e * 1024);
  }
}

// Epilogue of one output tile: four 128 x 128 quadrants through the 32 KiB image -> whole 256-byte rows (+ residual, + statistics of
// the stored values).  Workgroup-wide (all 512 threads, aligned); `img` = 32 KiB of LDS.
template <bool STATS>
__device__ __forceinline__ void p8_epilogue(const P8Args& p, f32x4 (&acc)[2][2][4][2], int tile, char* img, int lane, int wave) {
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * PM, n0 = tn * PN;
  const int chunk = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  if (P8_DBG(p, 4)) return;  // lab only: main loop without the epilogue
#pragma unroll
  for (int bh = 0; bh < 2; ++bh) {
    if (bh == 1 && n0 + 128 >= p.N) break;   // workgroup-uniform: a half-width last column tile
    // statistics in pairs of channels: v_pk_add_f32 / v_pk_fma_f32 (a wave64 vector instruction holds its SIMD for 4 cycles; the
    // unpacked form spent ~4 000 SIMD cycles per tile here: 7-15 % of a 1x1 convolution's tile)
    f32x2 s1[4], s2[4], piv[4];
#pragma unroll
    for (int ah = 0; ah < 2; ++ah) {
      // (the previous pass's readers are past their barrier below)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wr * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ch = wc * 4 + j * 2 + (fq >> 1);
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[ah][bh][i][j][r];
          *reinterpret_cast<bf16x4*>(img + row * 256 + ((ch ^ (row & 15)) << 4) + (fq & 1) * 8) = o;
        }
      }
      P8_LDS_SYNC();
      const int gcol = n0 + bh * 128 + chunk * 8;
      if (STATS && ah == 0) {
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(img + (chunk << 4));  // row 0 of the tile: every thread's pivot
        p8_pairs(pv, piv);
#pragma unroll
        for (int i = 0; i < 4; ++i) s1[i] = s2[i] = f32x2{0.f, 0.f};
      }
      // the four rows of this thread: R rows requested first (they queue behind nothing but the previous pass's stores), then the four
      // image reads back to back, then add / store / statistics -- one LDS latency and one memory latency per pass instead of four
      bf16x8 v[4], rr[4];
      const bool has_r = p.R != nullptr;
      float rscale[4] = {1.f, 1.f, 1.f, 1.f};
      if (has_r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int grow = m0 + ah * 128 + q * 32 + rsub;
          grow = grow < p.M ? grow : p.M - 1;
          size_t rrow = (size_t)grow;
          if (p.rp_w) rrow = p8_pooled_row(grow, p.rp_h, p.rp_w, p.rp_magic_hw, p.rp_magic_w, rscale[q]);   // wave-uniform: pooled residual
          rr[q] = *reinterpret_cast<const bf16x8*>(p.R + rrow * p.ldr + gcol);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 32 + rsub;
        v[q] = *reinterpret_cast<const bf16x8*>(img + row * 256 + ((chunk ^ (row & 15)) << 4));
      }
      if (has_r) {  // C = bf16(bf16(A.B^T) + R): what two separate launches would store
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 a[4], b[4];
          p8_pairs(v[q], a);
          p8_pairs(rr[q], b);
          const f32x2 sc = {rscale[q], rscale[q]};   // 1 (plain), 0.25 / 0 (pooled): exact scalings, so bf16(0.25 r) need not be formed first
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
        const int grow = m0 + ah * 128 + q * 32 + rsub;
        if (grow < p.M) {
          if (P8_DBG(p, 16)) __builtin_nontemporal_store(v[q], reinterpret_cast<bf16x8*>(p.C + (size_t)grow * p.ldc + gcol));
          else *reinterpret_cast<bf16x8*>(p.C + (size_t)grow * p.ldc + gcol) = v[q];
        }
        if (STATS && (long long)grow < p.stats_rows && !(P8_DBG(p, 256))) {
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
      P8_LDS_SYNC();
    }
    if (STATS && !(P8_DBG(p, 512))) {
      // threads with equal `chunk`: lanes l, l ^ 16, l ^ 32 of a wave, then the 8 waves through the (free) image, fixed order
      // lanes l, l ^ 32, then l ^ 16, with v_permlane32_swap / v_permlane16_swap (vector ALU).  As 64 `ds_bpermute`s per wave and
      // tile these sums cost ~4 000 cycles of the CU's LDS crossbar per tile (lab stamps: statistics epilogue 14 800 cycles against
      // 7 000 without; 9 200 with this reduction switched off).
      float t1[8], t2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t1[i] = (i & 1) ? s1[i >> 1].y : s1[i >> 1].x;
        t2[i] = (i & 1) ? s2[i >> 1].y : s2[i >> 1].x;
      }
      p8_rows_sum(t1);
      p8_rows_sum(t2);
      float* red = reinterpret_cast<float*>(img);  // [8 waves][16 chunks][16]
      if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[(wave * 16 + lane) * 16 + i] = t1[i];
          red[(wave * 16 + lane) * 16 + 8 + i] = t2[i];
        }
        if (wave == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) red[8 * 16 * 16 + lane * 8 + i] = (i & 1) ? piv[i >> 1].y : piv[i >> 1].x;
        }
      }
      P8_LDS_SYNC();
      if (threadIdx.x < 128) {
        const int c = threadIdx.x, ch = c >> 3, ci = c & 7;
        // all 16 reads first, then the sums in wave order (interleaved, every add waited for its own LDS round trip: 8 in a row)
        float r1[8], r2[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          r1[w] = red[(w * 16 + ch) * 16 + ci];
          r2[w] = red[(w * 16 + ch) * 16 + 8 + ci];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          a1 += r1[w];
          a2 += r2[w];
        }
        float* __restrict__ part = p.stats + (size_t)tm * 3 * p.N + n0 + bh * 128 + c;
        part[0] = red[8 * 16 * 16 + c];
        part[p.N] = a1;
        part[2 * (size_t)p.N] = a2;
      }
      P8_LDS_SYNC();
    }
  }
}

template <bool GATHER3, bool STATS>
__global__ __launch_bounds__(512) void conv_gemm_p8_kernel(const P8Args p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int G = gridDim.x, nk = p.K / PK;
  const P8Items items = p8_items(p, blockIdx.x, G, nk);
  if (items.n_items == 0) return;
  // De-phasing hook (lab: coin_p8_stagger; off by default -- measured no effect, the CUs do not run in lockstep): workgroups that own
  // one tile less than the others may start late, spread over `stagger_phases` offsets inside one tile time.
  if (p.stagger_phases > 1 && (int)blockIdx.x >= p.stagger_first) {
    const int ph = ((int)blockIdx.x >> 3) % p.stagger_phases;   // blockIdx.x & 7 = XCD: every XCD spreads its own CUs
    const long long wait = (long long)p.stagger_ticks * ph;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
  }
  int total_kt = items.n_whole * nk;
  if (items.n_items > items.n_whole) total_kt += items.ke(items.n_whole) - items.kb(items.n_whole);   // (>= 2: every item holds at least two K-tiles)
  int item_idx = 0;

  NtCursor<GATHER3> cur;
  cur.idx = 0;
  cur.kt = items.kb(0);
  cur.ke = items.ke(0);
  cur.buf = 0;
  nt_set_tile<GATHER3>(cur, p, items.tile(0), wave, lane);
  auto advance = [&]() {
    cur.buf ^= 1;
    if (++cur.kt == cur.ke) {
      if (++cur.idx < items.n_items) {
        cur.kt = items.kb(cur.idx);
        cur.ke = items.ke(cur.idx);
        nt_set_tile<GATHER3>(cur, p, items.tile(cur.idx), wave, lane);
      }
    }
  };

  // ---- prologue: K-tile 0 (four half-tiles) and three half-tiles of K-tile 1
  nt_stage<GATHER3, 0>(cur, p, lds, wave);
  nt_stage<GATHER3, 1>(cur, p, lds, wave);
  nt_stage<GATHER3, 2>(cur, p, lds, wave);
  nt_stage<GATHER3, 3>(cur, p, lds, wave);
  advance();
  nt_stage<GATHER3, 0>(cur, p, lds, wave);
  nt_stage<GATHER3, 1>(cur, p, lds, wave);
  nt_stage<GATHER3, 2>(cur, p, lds, wave);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // B-lo(0), A-lo(0) have landed; B-hi(0), A-hi(0) are waited for in phases 1, 2
  P8_BAR();

  // ---- fragment read offsets inside a K-tile buffer
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int sw = ((ks * 4 + fq) ^ (fr & 7)) << 4;
    a_off[ks] = (wr * 64 + fr) * 128 + sw;
    b_off[ks] = 2 * P_HALF + (wc * 32 + fr) * 128 + sw;
  }

  int g = 0;  // K-tile sequence number of this workgroup (buffer g & 1)
  long long st_main = 0, st_epi = 0, st_n = 0, st_t = 0;   // lab only (dbg bit 6): s_memtime split of a tile into main loop / epilogue
  if (P8_DBG(p, 64)) st_t = __builtin_amdgcn_s_memtime();
  for (;;) {
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[x][y][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool bhi = p8_has_bhi(p, items.tile(item_idx));
    if (wr == 1) P8_BAR();  // G1 runs one barrier behind G0
    for (int kt = items.kb(item_idx), kt_end = items.ke(item_idx); kt < kt_end; ++kt, ++g) {
      const char* kb = lds + (g & 1) * P_KT;
      bf16x8 af[4][2], b0[2][2], b1[2][2];
      // ------------------------------------------------ phase 1
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b0[j][ks] = *reinterpret_cast<const bf16x8*>(kb + b_off[ks] + j * 2048);
      P8_SCHED();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(kb + a_off[ks] + i * 2048);
      P8_SCHED();
      if (g + 1 < total_kt) {
        nt_stage<GATHER3, 3>(cur, p, lds, wave);  // A-hi(g+1)
        advance();                                       // cursor -> K-tile g+2
      }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the B-lo reads are back: B-lo may be re-staged in phase 2
      if (g + 1 < total_kt)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // B-hi(g) has landed (read in phase 2)
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[0][0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j][ks], af[i][ks], acc[0][0][i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      // ------------------------------------------------ phase 2
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b1[j][ks] = *reinterpret_cast<const bf16x8*>(kb + b_off[ks] + P_HALF + j * 2048);
      P8_SCHED();
      const bool more = g + 2 < total_kt;
      if (more) {
        nt_stage<GATHER3, 0>(cur, p, lds, wave);  // B-lo(g+2)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // A-hi(g) has landed (read in phase 3)
      } else if (g + 1 < total_kt) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      __builtin_amdgcn_s_setprio(1);
      if (bhi) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[0][1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j][ks], af[i][ks], acc[0][1][i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      // ------------------------------------------------ phase 3
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(kb + a_off[ks] + P_HALF + i * 2048);
      P8_SCHED();
      if (more) nt_stage<GATHER3, 1>(cur, p, lds, wave);  // A-lo(g+2)
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      __builtin_amdgcn_s_setprio(1);
      if (bhi) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[1][1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j][ks], af[i][ks], acc[1][1][i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      // ------------------------------------------------ phase 4
      if (more) {
        nt_stage<GATHER3, 2>(cur, p, lds, wave);  // B-hi(g+2)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // B-lo(g+1), A-lo(g+1) have landed (this wave's part; the barrier publishes it)
      } else if (g + 1 < total_kt) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[1][0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j][ks], af[i][ks], acc[1][0][i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_SCHED();
      P8_BAR();
      P8_SCHED();
    }
    if (wr == 0) P8_BAR();  // both groups aligned again
    if (P8_DBG(p, 64)) {
      const long long t = __builtin_amdgcn_s_memtime();
      st_main += t - st_t;
      st_t = t;
    }

    if (items.whole(item_idx)) {
      p8_epilogue<STATS>(p, acc, items.tile(item_idx), lds + P_IMG, lane, wave);
      if (P8_DBG(p, 64)) {
        if (P8_DBG(p, 128)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // bit 7: the stores' drain is charged to the epilogue
        const long long t = __builtin_amdgcn_s_memtime();
        st_epi += t - st_t;
        st_t = t;
        ++st_n;
      }
    } else {
      // split-K piece: fp32 accumulators in thread-private order (float4 index q * 512 + thread): coalesced 16-byte stores
      unsigned tid = threadIdx.x;
      asm volatile("" : "+v"(tid));  // keeps the 32 store addresses from being formed (and spilled) ahead of the K loop
      f32x4* __restrict__ sl = reinterpret_cast<f32x4*>(p.slab) + ((size_t)(blockIdx.x % items.rem) * p.split + items.piece) * (32 * 512) + tid;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) sl[(((x * 2 + y) * 4 + i) * 2 + j) * 512] = acc[x][y][i][j];
    }
    if (++item_idx >= items.n_items) break;
  }
#ifdef COIN_LAB
  if ((P8_DBG(p, 64)) && threadIdx.x == 0 && blockIdx.x < 1024) {
    coin_p8_stamp_buf[blockIdx.x * 4 + 0] = st_main;
    coin_p8_stamp_buf[blockIdx.x * 4 + 1] = st_epi;
    coin_p8_stamp_buf[blockIdx.x * 4 + 2] = st_n;
  }
#endif
}

// Split-K tail, pass 1: slab[tile][0] += slab[tile][1] + ... (piece order: bit-reproducible), every CU busy: grid = rem * 32 workgroups
// of 256 threads, two float4 per thread and piece.
__global__ __launch_bounds__(256) void conv_gemm_p8_slab_sum_kernel(float* __restrict__ slab, int split) {
  f32x4* __restrict__ sl = reinterpret_cast<f32x4*>(slab) + (size_t)(blockIdx.x >> 5) * split * (32 * 512) + (blockIdx.x & 31) * 512 + threadIdx.x;
  f32x4 a = sl[0], b = sl[256];
  for (int s = 1; s < split; ++s) {
    a += sl[(size_t)s * (32 * 512)];
    b += sl[(size_t)s * (32 * 512) + 256];
  }
  sl[0] = a;
  sl[256] = b;
}

// Split-K tail, pass 2: the summed accumulators of one tail tile (thread-private order, as the pieces stored them) -> the epilogue.
// grid = rem tiles, 512 threads.
template <bool STATS>
__global__ __launch_bounds__(512) void conv_gemm_p8_tail_kernel(const P8Args p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4* __restrict__ sl = reinterpret_cast<const f32x4*>(p.slab) + (size_t)blockIdx.x * p.split * (32 * 512) + threadIdx.x;
  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = sl[(((x * 2 + y) * 4 + i) * 2 + j) * 512];
  p8_epilogue<STATS>(p, acc, p.whole_tiles + blockIdx.x, lds, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------- TN kernel
// Weight gradient: dW[co][k] = sum over pixels m of gy[m][co] * xcol[m][k], k = tap * Cin + ci.  Output tile = 256 co x 256 k (one
// tap), K-tile = 64 pixels.  Both operands are pixel-major, so a half-tile is [64 pixels][128 channels] (256-byte rows; one DMA
// instruction = 4 rows) and the MFMA fragments (8 consecutive pixels of one channel per lane) are read with ds_read_b64_tr_b16.
// 16-byte chunk c of row r sits at chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3)): the two 4-row blocks a 32-lane half reads per
// transposed read (8 rows apart) cover all 64 banks.
// Work split (XCD-coherent): the pixels are cut into 8 segments, one per XCD (workgroups b, b + 8, ... share an XCD's L2); inside an
// XCD each of the `cpx` workgroups owns ONE output tile and walks the whole segment, all of them at the same pixels at the same
// time: the gy / x panels they stream are shared 16-fold in L2 (without that the 36 tiles of the 3x3 layer move 14.8 GB and the
// kernel is HBM-bound).  Tiles left over after the full rounds (T % cpx) are split along the pixels again so that every workgroup
// stays busy.  A workgroup keeps its accumulators for a whole item and writes one fp32 partial tile per item; tn_reduce_kernel sums
// the partials of a tile in a fixed order (no atomics: bit-reproducible).
struct TnArgs {
  const bf16_t* GY;
  const bf16_t* X;
  float* slab;
  int M, Cout, Cin, Ktot, H, W;
  int tiles_k, ntiles, nkt, cpx;
  unsigned magic_w;  // ceil(2^32 / W)
  int hw, r64, r32;                   // H * W, 64 % hw, 32 % hw
  unsigned g_bytes, x_bytes, x_bias;  // buffer sizes for the range-checked DMA; x is addressed from X - x_bias
  int dbg;           // lab only (coin_p8_debug): bit 0 = every DMA returns zeros, bit 1 = no partial-tile stores
};

// item `idx` of workgroup b -> (tile, [kb, ke)); false = no such item.  Shared by the main kernel and the reduction.
__host__ __device__ __forceinline__ bool tn_item(int b, int idx, int ntiles, int nkt, int cpx, int& tile, int& kb, int& ke) {
  const int x = b & 7, j = b >> 3;
  const int full = ntiles / cpx, r = ntiles - full * cpx;
  const int sb = (int)((long long)x * nkt / 8), se = (int)((long long)(x + 1) * nkt / 8);
  if (idx < full) {
    tile = idx * cpx + j;
    kb = sb;
    ke = se;
    return ke > kb;
  }
  if (idx > full || r == 0) return false;
  const int nsub = cpx / r;
  if (j >= r * nsub) return false;
  const int tt = j % r, sub = j / r, len = se - sb;
  tile = full * cpx + tt;
  kb = sb + (int)((long long)sub * len / nsub);
  ke = sb + (int)((long long)(sub + 1) * len / nsub);
  return ke > kb;
}

template <bool GATHER3>
struct TnCursor {
  unsigned go[2], xo[2];  // byte offsets (into gy / into x - bias) of this lane's two pixels among the K-tile's FIRST 32, incl. the chunk
  int rem[2];             // 3x3: index of those pixels inside their image
  int tile, kt, ke, item, buf, dy, dx;
};

constexpr unsigned TN_OOB = 0x80000000u;  // an offset beyond every buffer (host checks sizes < 2^31): the DMA then writes zeros

__device__ __forceinline__ bool tn_tap_ok(int rem, int dy, int dx, const TnArgs& p) {
  const int oy = (int)__umulhi((unsigned)rem, p.magic_w);
  const int ox = rem - oy * p.W;
  return (unsigned)(oy + dy) < (unsigned)p.H && (unsigned)(ox + dx) < (unsigned)p.W;
}

// LDS image of a half-tile: [32 pixels][256 channels] = 512-byte rows, one DMA instruction = 2 rows; 16-byte chunk c of row r sits at
// chunk c ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1)): the two 4-row blocks a 32-lane half reads per transposed read (8 rows apart,
// 32 bytes wide) cover all 64 banks.
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | (((row >> 3) & 1) << 1); }

template <bool GATHER3>
__device__ __forceinline__ void tn_set_pos(TnCursor<GATHER3>& c, const TnArgs& p, int tile, int kt, int wave, int lane) {
  const int tco = tile / p.tiles_k, tk = tile - tco * p.tiles_k;
  const int co0 = tco * 256, k0 = tk * 256;
  c.tile = tile;
  c.kt = kt;
  // Channel counts that are multiples of 128 only (layer2 of the backbone): the last co / k tile is half valid, and a 3x3 K-tile of
  // 256 spans TWO taps when Cin = 128.  Handled per LANE: a lane's 16-byte chunk lies in one tap (chunks 0-15 / 16-31; the swizzle only
  // touches the low four bits of the chunk index, so both of a lane's rows share the tap), chunks beyond Cout / Ktot get an offset
  // outside the buffer and arrive as zeros -- the MFMA work of the missing half is wasted, no byte of it is fetched.
  const int hi = (lane & 31) >> 4;          // chunk index bit 4
  const int kl = k0 + hi * 128;             // first k of this lane's 128-channel half of the K-tile
  int tap = 0, cib = kl;
  c.dy = c.dx = 0;
  if (GATHER3) {
    tap = kl / p.Cin;
    cib = kl - tap * p.Cin;
    c.dy = tap / 3 - 1;
    c.dx = tap % 3 - 1;
  }
  const bool g_ok = co0 + hi * 128 < p.Cout, x_ok = kl < p.Ktot;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = (wave * 2 + e) * 2 + (lane >> 5);
    const int lc = (lane & 31) ^ tn_swz(row);
    const long long m = (long long)kt * 64 + row;
    c.go[e] = g_ok ? (unsigned)((m * p.Cout + co0 + lc * 8) * 2) : TN_OOB;
    c.xo[e] = x_ok ? (unsigned)(((m + c.dy * p.W + c.dx) * p.Cin + cib + (lc & 15) * 8) * 2 + p.x_bias) : TN_OOB;  // >= 0: the bias covers the largest negative shift
    c.rem[e] = GATHER3 ? (int)(m % (p.H * p.W)) : 0;
  }
}

// WHICH: 0 = X-lo, 1 = G-lo (pixels 0-31 of the K-tile), 2 = X-hi, 3 = G-hi (pixels 32-63).  Range-checked buffer LDS-DMA: pixels
// beyond M (offset >= num_records) and taps outside the image (offset forced to TN_OOB) arrive as zeros, with no address select.
template <bool GATHER3, int WHICH>
__device__ __forceinline__ void tn_stage(const TnCursor<GATHER3>& c, const TnArgs& p, __amdgpu_buffer_rsrc_t rg, __amdgpu_buffer_rsrc_t rx, char* lds, int wave) {
  constexpr bool IS_G = (WHICH & 1) != 0;
  constexpr int HI = WHICH >> 1;
  char* dst = lds + c.buf * P_KT + WHICH * P_HALF + wave * 2048;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    unsigned off = IS_G ? c.go[e] + (unsigned)(HI * 64) * p.Cout : c.xo[e] + (unsigned)(HI * 64) * p.Cin;
    if (!IS_G && GATHER3) {
      int r = c.rem[e];
      if (HI) {
        r += p.r32;  // 32 % (H * W)
        r = r >= p.hw ? r - p.hw : r;
      }
      off = tn_tap_ok(r, c.dy, c.dx, p) ? off : TN_OOB;
    }
    if (P8_DBG(p, 1)) off = TN_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_G ? rg : rx, (__attribute__((address_space(3))) void*)(dst + e * 1024), 16, off, 0, 0, 0);
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive pixels (rows) of one channel per lane: two transposed reads, rows r .. r+3 and r+4 .. r+7 (2048 bytes on).
// Issued from inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr16_b64 INTRINSIC whenever an LDS-DMA
// is in flight (it has no memory operand to disambiguate; plain ds_read does not get that wait), which drained the whole DMA
// pipeline once per K-tile -- measured: the reason the round-2 weight-gradient kernel and the first p8 TN versions sat at 0.8 PF.
// The compiler neither counts these reads nor waits for them: every consumer sits behind TN_MFMA's `s_waitcnt lgkmcnt(0)` +
// sched_barrier(0) (guide section 5.7, form iii).
template <int IMM>
__device__ __forceinline__ bf16x8 tn_frag(unsigned adr) {
  s16x4 a, b;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(adr), "i"(IMM) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(adr), "i"(IMM + 2048) : "memory");
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// Phases of a K-tile g (buffer g & 1), one MFMA k-step (32 pixels) x one half of the wave's output channels each:
//   phase 1: read x (4 tiles) + gy (4 tiles, co half 0) of pixels 0-31 -> 16 MFMA; DMA G-hi(g+1)
//   phase 2: read gy (co half 1)                                        -> 16 MFMA; DMA X-lo(g+2); wait: pixels 32-63 of g landed
//   phase 3: the same for pixels 32-63;                                             DMA G-lo(g+2)
//   phase 4:                                                                        DMA X-hi(g+2); wait: pixels 0-31 of g+1 landed
// A wave retires its LDS reads (lgkmcnt(0)) BEFORE the phase's first barrier, so a half-tile is re-staged in the phase after its
// last read; five half-tiles (80 KiB) stay in flight (`s_waitcnt vmcnt(10)`): a load has five to six phases to land.
// The read / DMA sections run beside the other group's 16-MFMA burst, which leaves them about 30 vector-instruction issue slots:
// DMA addresses are 32-bit buffer offsets advanced by one add per K-tile, fragment addresses are toggled between the buffers by XOR.
template <bool GATHER3>
__global__ __launch_bounds__(512) void conv_wgrad_p8_kernel(const TnArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nitems = p.ntiles / p.cpx + 1;
  int total_kt = 0, first_item = -1;
  for (int i = 0; i < nitems; ++i) {
    int t, kb, ke;
    if (tn_item(blockIdx.x, i, p.ntiles, p.nkt, p.cpx, t, kb, ke)) {
      total_kt += ke - kb;
      if (first_item < 0) first_item = i;
    }
  }
  if (total_kt == 0) return;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.GY), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.X) - p.x_bias), 0, p.x_bytes, 0x00020000);

  TnCursor<GATHER3> cur;
  cur.buf = 0;
  auto open_item = [&](int from) {  // first non-empty item with index >= from
    for (int i = from; i < nitems; ++i) {
      int t, kb, ke;
      if (tn_item(blockIdx.x, i, p.ntiles, p.nkt, p.cpx, t, kb, ke)) {
        cur.item = i;
        cur.ke = ke;
        tn_set_pos<GATHER3>(cur, p, t, kb, wave, lane);
        return;
      }
    }
    cur.item = nitems;
  };
  open_item(first_item);
  auto advance = [&]() {
    cur.buf ^= 1;
    if (++cur.kt == cur.ke) {
      open_item(cur.item + 1);
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        cur.go[e] += 128u * p.Cout;
        cur.xo[e] += 128u * p.Cin;
        if (GATHER3) {
          const int r = cur.rem[e] + p.r64;  // r64 = 64 % (H * W)
          cur.rem[e] = r >= p.hw ? r - p.hw : r;
        }
      }
    }
  };

  tn_stage<GATHER3, 0>(cur, p, rg, rx, lds, wave);
  tn_stage<GATHER3, 1>(cur, p, rg, rx, lds, wave);
  tn_stage<GATHER3, 2>(cur, p, rg, rx, lds, wave);
  tn_stage<GATHER3, 3>(cur, p, rg, rx, lds, wave);
  if (total_kt > 1) {
    advance();
    tn_stage<GATHER3, 0>(cur, p, rg, rx, lds, wave);
    tn_stage<GATHER3, 1>(cur, p, rg, rx, lds, wave);
    tn_stage<GATHER3, 2>(cur, p, rg, rx, lds, wave);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // pixels 0-31 of K-tile 0 have landed
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  P8_BAR();

  // ---- transposed-read addresses inside a half-tile: lane group g4 = lane >> 4 takes pixels 8 g4 .. 8 g4 + 7; per read it supplies
  // the address of row 8 g4 + q4 (+ 4 for the second read), columns 4 p4 .. 4 p4 + 3 of the tile's 16 channels
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int trow = 8 * g4 + q4, tsw = tn_swz(trow);
  const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)lds;  // P_KT-aligned (1024-byte aligned, only LDS object)
  unsigned a_adr[4], b_adr[4];  // G tiles i (co half 0; half 1 = + 256 bytes), X tiles bh * 2 + j;  buffer 0, toggled with ^= P_KT
#pragma unroll
  for (int i = 0; i < 4; ++i) a_adr[i] = lds0 + P_HALF + trow * 512 + (((wr * 8 + i * 2 + (p4 >> 1)) ^ tsw) << 4) + (p4 & 1) * 8;
#pragma unroll
  for (int t = 0; t < 4; ++t) b_adr[t] = lds0 + trow * 512 + ((((t >> 1) * 16 + wc * 4 + (t & 1) * 2 + (p4 >> 1)) ^ tsw) << 4) + (p4 & 1) * 8;

  const int fr = lane & 15, fq = lane >> 4;
  int item_c = first_item, left_c;  // K-tiles left in the item being accumulated
  {
    int t, kb, ke;
    tn_item(blockIdx.x, item_c, p.ntiles, p.nkt, p.cpx, t, kb, ke);
    left_c = ke - kb;
  }
  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define TN_MFMA(AH)                                                                                                                     \
  do {                                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* reads retired before the barrier: see the re-staging rule above */           \
    P8_SCHED();                                                                                                                         \
    P8_BAR();                                                                                                                           \
    P8_SCHED();                                                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int t = 0; t < 4; ++t)                                         \
        acc[AH][t >> 1][i][t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t], af[i], acc[AH][t >> 1][i][t & 1], 0, 0, 0);          \
    __builtin_amdgcn_s_setprio(0);                                                                                                      \
    P8_SCHED();                                                                                                                         \
    P8_BAR();                                                                                                                           \
    P8_SCHED();                                                                                                                         \
  } while (0)

  if (wr == 1) P8_BAR();  // G1 runs one barrier behind G0
  for (int g = 0; g < total_kt; ++g) {
    const bool more1 = g + 1 < total_kt, more2 = g + 2 < total_kt;
    bf16x8 af[4], bf[4];
    // ------------------------------------------------ phase 1: pixels 0-31, co half 0
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = tn_frag<0>(b_adr[t]);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tn_frag<0>(a_adr[i]);
    P8_SCHED();
    if (more1) {
      tn_stage<GATHER3, 3>(cur, p, rg, rx, lds, wave);  // G-hi(g+1)
      advance();                                         // cursor -> K-tile g+2
    }
    TN_MFMA(0);
    // ------------------------------------------------ phase 2: pixels 0-31, co half 1
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tn_frag<256>(a_adr[i]);
    P8_SCHED();
    if (more2) {
      tn_stage<GATHER3, 0>(cur, p, rg, rx, lds, wave);  // X-lo(g+2)
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // pixels 32-63 of K-tile g have landed
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN_MFMA(1);
    // ------------------------------------------------ phase 3: pixels 32-63, co half 0
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = tn_frag<2 * P_HALF>(b_adr[t]);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tn_frag<2 * P_HALF>(a_adr[i]);
    P8_SCHED();
    if (more2) tn_stage<GATHER3, 1>(cur, p, rg, rx, lds, wave);  // G-lo(g+2)
    TN_MFMA(0);
    // ------------------------------------------------ phase 4: pixels 32-63, co half 1
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tn_frag<2 * P_HALF + 256>(a_adr[i]);
    P8_SCHED();
    if (more2) {
      tn_stage<GATHER3, 2>(cur, p, rg, rx, lds, wave);  // X-hi(g+2)
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // pixels 0-31 of K-tile g+1 have landed
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN_MFMA(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // the next K-tile sits in the other buffer
      a_adr[i] ^= P_KT;
      b_adr[i] ^= P_KT;
    }
    // ------------------------------------------------ end of an item: partial tile -> slab
    if (--left_c == 0) {
      float* __restrict__ out = p.slab + ((size_t)item_c * gridDim.x + blockIdx.x) * 65536;
      int tile_c, kb_c, ke_c;
      tn_item(blockIdx.x, item_c, p.ntiles, p.nkt, p.cpx, tile_c, kb_c, ke_c);
      const int tco_c = tile_c / p.tiles_k;
      const bool x1_ok = tco_c * 256 + 128 < p.Cout, y1_ok = (tile_c - tco_c * p.tiles_k) * 256 + 128 < p.Ktot;   // the tile's upper halves exist
      if (!(P8_DBG(p, 2)))
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int row = x * 128 + wr * 64 + i * 16 + fr, col = y * 128 + wc * 32 + j * 16 + fq * 4;
              if ((x == 0 || x1_ok) && (y == 0 || y1_ok))   // tn_reduce_kernel never reads the missing halves of an edge tile
                *reinterpret_cast<f32x4*>(out + row * 256 + col) = acc[x][y][i][j];
              acc[x][y][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
      for (++item_c; item_c < nitems; ++item_c) {
        int t, kb2, ke2;
        if (tn_item(blockIdx.x, item_c, p.ntiles, p.nkt, p.cpx, t, kb2, ke2)) {
          left_c = ke2 - kb2;
          break;
        }
      }
    }
  }
#undef TN_MFMA
  if (wr == 0) P8_BAR();
}

// dW[co][k] = sum of the partial tiles of tile (tco, tk) in a fixed order (XCD segment, then sub-range).  One workgroup per 4 rows
// of a tile (thread = 4 consecutive columns of one row); the partial list is walked 4 at a time so that 4 loads are in flight.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Ktot, int tiles_k, int ntiles,
                                                        int nkt, int cpx) {
  const int tile = blockIdx.x >> 6, row = (blockIdx.x & 63) * 4 + (threadIdx.x >> 6);
  const int tco = tile / tiles_k, tk = tile - tco * tiles_k;
  const int c4 = (threadIdx.x & 63) * 4;
  if (tco * 256 + row >= Cout || tk * 256 + c4 >= Ktot) return;   // half-valid edge tiles (channel counts that are odd multiples of 128): never stored
  const int grid = 8 * cpx, full = ntiles / cpx, r = ntiles - full * cpx;
  const bool whole = tile < full * cpx;
  const int idx = whole ? tile / cpx : full;
  const int j0 = whole ? tile % cpx : tile - full * cpx;
  const int nsub = whole ? 1 : cpx / r, jstep = whole ? 0 : r;
  const float* __restrict__ base = slab + (size_t)idx * grid * 65536 + row * 256 + c4;
  const int n = 8 * nsub;  // candidate partials, order: x-major, then sub
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < n; q0 += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u, x = q / nsub, sub = q - x * nsub;
      const int b = (j0 + sub * jstep) * 8 + x;
      int t, kb, ke;
      const bool ok = q < n && tn_item(b, idx, ntiles, nkt, cpx, t, kb, ke);
      v[u] = ok ? *reinterpret_cast<const f32x4*>(base + (size_t)b * 65536) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) a += v[u];
  }
  *reinterpret_cast<f32x4*>(dw + (size_t)(tco * 256 + row) * Ktot + tk * 256 + c4) = a;
}

// The same reduction for launches with FEW tiles (every tile then is a leftover tile cut into 8 * (cpx / r) pieces: two tiles -> 128 partials
// each).  One workgroup per ROW of a tile; its four waves take every fourth partial of the list (16 loads in flight per lane) and their
// sums are combined in wave order through LDS: a fixed order again, 4x the rows in flight and a quarter of the dependent latencies per
// thread (the 128-channel layers: the reduction took 50 us of a 76 us launch with the kernel above, rocprofv3 of tools/gemm_lab wbench).
__global__ __launch_bounds__(256) void tn_reduce_wide_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Ktot, int tiles_k, int ntiles,
                                                             int nkt, int cpx) {
  __shared__ f32x4 part[3][64];
  const int tile = blockIdx.x >> 8, row = blockIdx.x & 255;
  const int tco = tile / tiles_k, tk = tile - tco * tiles_k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = lane * 4;
  if (tco * 256 + row >= Cout) return;                      // whole workgroup: no barrier is reached
  const bool col_ok = tk * 256 + c4 < Ktot;
  const int grid = 8 * cpx, full = ntiles / cpx, r = ntiles - full * cpx;
  const bool whole = tile < full * cpx;
  const int idx = whole ? tile / cpx : full;
  const int j0 = whole ? tile % cpx : tile - full * cpx;
  const int nsub = whole ? 1 : cpx / r, jstep = whole ? 0 : r;
  const float* __restrict__ base = slab + (size_t)idx * grid * 65536 + row * 256 + c4;
  const int n = 8 * nsub;
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (col_ok) {
    for (int q0 = wave; q0 < n; q0 += 64) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int q = q0 + 4 * u, x = q / nsub, sub = q - x * nsub;
        const int b = (j0 + sub * jstep) * 8 + x;
        int t, kb, ke;
        const bool ok = q < n && tn_item(b, idx, ntiles, nkt, cpx, t, kb, ke);
        v[u] = ok ? *reinterpret_cast<const f32x4*>(base + (size_t)b * 65536) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) a += v[u];
    }
  }
  if (wave) part[wave - 1][lane] = a;
  __syncthreads();
  if (wave == 0 && col_ok) {
    a += part[0][lane];
    a += part[1][lane];
    a += part[2][lane];
    *reinterpret_cast<f32x4*>(dw + (size_t)(tco * 256 + row) * Ktot + tk * 256 + c4) = a;
  }
}

}  // namespace

static int p8_grid(int ntiles) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return ntiles < 256 ? ntiles : 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return ntiles < cus ? ntiles : cus;
}

#ifdef COIN_LAB
int coin_p8_debug = 0;  // lab hook, see TnArgs::dbg
int coin_p8_read_stamps(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(coin_p8_stamp_buf), sizeof(long long) * 4 * (size_t)n); }
int coin_p8_stagger = -1;  // lab hook: phases of the start-time stagger (-1: default, 0/1: off)
int coin_p8_splitk = -1;  // lab hook: -1 = default policy, 0 = never split the tail round, 1 = split whenever it is possible
#else
static constexpr int coin_p8_debug = 0, coin_p8_stagger = -1, coin_p8_splitk = -1;
#endif

bool coin_p8_nt_ok(int M, int N, int K, int mode, int Cin, int lda, int ldb) {
  // N % 256 == 128: a half-width last column tile (p8_has_bhi).  N = 128 itself stays on the 256 x 128 kernel of conv_gemm.hip: one column
  // tile with half of its MFMA phases empty ran layer2's 128-channel convolutions SLOWER than that kernel (tools/gemm_lab, round 5:
  // 3x3 [66 800 x 128 x 1152] 0.057 vs 0.045 ms, 1x1 K = 512 0.0265 vs 0.0234 ms -- 261 row tiles are two rounds on 256 CUs either way)
  if (M <= 0 || N % 128 || N < 256 || K % PK || K < 2 * PK) return false;
  if ((size_t)M * (mode == 1 ? Cin : lda) * 2 >= 0x7f000000ull || (size_t)N * ldb * 2 >= 0x7f000000ull) return false;  // 32-bit buffer offsets
  if (mode == 1 && (Cin % PK || K != 9 * Cin)) return false;
  return true;
}

// Split-K plan of the tail round (see P8Args): only where a tile is long enough for the combine pass (two passes over `rem * split`
// fp32 tiles of 256 KiB) to cost less than the idle CUs it removes -- K >= 2048 -- and the tail fills at most half of the CUs.
// G = the number of CUs (also when there are fewer tiles than CUs: then every tile is a leftover tile and is cut along K).
static void p8_nt_plan(int ntiles, int nk, int G, bool have_ws, int force, int& whole, int& rem, int& split) {
  whole = ntiles; rem = 0; split = 1;
  const int r = ntiles % G;
  if (!have_ws || force == 0 || r == 0 || 2 * r > G || (nk < 32 && force != 1)) return;   // default policy: K >= 2048
  int s = G / r;
  if (s > 8) s = 8;
  while (s > 1 && nk / s < 2) --s;
  if (s < 2) return;
  whole = ntiles - r; rem = r; split = s;
}

size_t coin_p8_nt_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N % 128 || N < 256 || K % PK) return 0;
  const int ntiles = ((M + PM - 1) / PM) * ((N + PN - 1) / PN), G = p8_grid(1 << 30);
  int whole, rem, split;
  p8_nt_plan(ntiles, K / PK, G, true, 1, whole, rem, split);   // upper bound: as if forced on
  return (size_t)rem * split * 65536 * sizeof(float);
}

int coin_p8_nt_launch(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc, const void* R, int ldr,
                      int M, int N, int K, float* stats, long long stats_rows, void* workspace, size_t workspace_bytes, hipStream_t st, int rp_h, int rp_w) {
  P8Args a;
  a.A = (const bf16_t*)A; a.lda = lda;
  a.B = (const bf16_t*)B; a.ldb = ldb;
  a.C = (bf16_t*)C; a.ldc = ldc;
  a.R = (const bf16_t*)R; a.ldr = ldr;
  a.rp_h = R ? rp_h : 0; a.rp_w = R ? rp_w : 0;
  a.rp_magic_hw = a.rp_w ? (unsigned)((0x100000000ull + (unsigned)(rp_h * rp_w) - 1) / (unsigned)(rp_h * rp_w)) : 0;
  a.rp_magic_w = a.rp_w ? (unsigned)((0x100000000ull + (unsigned)rp_w - 1) / (unsigned)rp_w) : 0;
  a.M = M; a.N = N; a.K = K; a.H = H; a.W = W; a.Cin = Cin;
  a.stats = stats; a.stats_rows = stats_rows;
  a.tiles_m = (M + PM - 1) / PM; a.tiles_n = (N + PN - 1) / PN;
  a.dbg = coin_p8_debug;
  a.stagger_first = a.stagger_phases = a.stagger_ticks = 0;
  a.a_bytes = (unsigned)((size_t)M * (mode == 1 ? Cin : lda) * 2);
  a.b_bytes = (unsigned)((size_t)N * ldb * 2);
  const int ntiles = a.tiles_m * a.tiles_n;
  const int cus = p8_grid(1 << 30);
  int grid = ntiles < cus ? ntiles : cus;
  const int force = coin_p8_splitk;   // -1 = the default policy (lab builds can override it)
  p8_nt_plan(ntiles, K / PK, cus, workspace != nullptr, force, a.whole_tiles, a.rem, a.split);
  if ((size_t)a.rem * a.split * 65536 * sizeof(float) > workspace_bytes) { a.whole_tiles = ntiles; a.rem = 0; a.split = 1; }
  if (a.split > 1 && a.whole_tiles == 0) grid = a.rem * a.split;   // fewer tiles than CUs: only pieces (<= cus of them)
  a.slab = (float*)workspace;
  {
    const int phases = coin_p8_stagger >= 0 ? coin_p8_stagger : 0;   // off: no effect measured (lab13)
    const int r = ntiles % grid;
    if (phases > 1 && a.split == 1 && ntiles > grid && r != 0) {
      const int tile_ticks = (K / PK) * 150 + 300;   // 10 ns ticks: ~1.5 us per K-tile of 64 at ~1 PFLOP/s + the epilogue
      a.stagger_first = r;
      a.stagger_phases = phases;
      a.stagger_ticks = tile_ticks / phases;
    }
  }
#define P8_LAUNCH(G3, ST)                                                                                                          \
  do {                                                                                                                             \
    static bool attr_set = false;                                                                                                  \
    if (!attr_set) {                                                                                                               \
      (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<G3, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);      \
      attr_set = true;                                                                                                             \
    }                                                                                                                              \
    conv_gemm_p8_kernel<G3, ST><<<grid, 512, P_LDS, st>>>(a);                                                                      \
  } while (0)
  if (mode == 1) {
    if (stats) P8_LAUNCH(true, true); else P8_LAUNCH(true, false);
  } else {
    if (stats) P8_LAUNCH(false, true); else P8_LAUNCH(false, false);
  }
#undef P8_LAUNCH
  if (a.split > 1) {
    conv_gemm_p8_slab_sum_kernel<<<a.rem * 32, 256, 0, st>>>(a.slab, a.split);
    if (stats) conv_gemm_p8_tail_kernel<true><<<a.rem, 512, 128 * 256, st>>>(a);
    else conv_gemm_p8_tail_kernel<false><<<a.rem, 512, 128 * 256, st>>>(a);
  }
  return coin_launch_status();
}

static void tn_plan(int M, int Cout, int Ktot, TnArgs& a, int& grid) {
  a.tiles_k = (Ktot + 255) / 256;
  a.ntiles = ((Cout + 255) / 256) * a.tiles_k;
  a.nkt = (M + 63) / 64;  // pixels beyond M read a zero page
  a.cpx = p8_grid(1 << 30) / 8;
  if (a.cpx < 1) a.cpx = 1;
  grid = 8 * a.cpx;
}

bool coin_p8_tn_ok(int M, int Cout, int Cin, int Ktot, int mode) {
  if (M <= 0 || Cout % 128 || Cin % 128) return false;
  if (((size_t)M + 64 + 256) * (size_t)(Cout > Cin ? Cout : Cin) * 2 >= 0x7f000000ull) return false;  // 32-bit buffer offsets, TN_OOB beyond them
  return mode == 0 ? Ktot == Cin : Ktot == 9 * Cin;
}

size_t coin_p8_tn_workspace_bytes(int M, int Cout, int Ktot) {
  TnArgs a;
  int grid;
  tn_plan(M, Cout, Ktot, a, grid);
  return (size_t)grid * (a.ntiles / a.cpx + 1) * 65536 * sizeof(float);
}

int coin_p8_tn_launch(const void* GY, const void* X, int mode, int H, int W, int Cin, int M, int Cout, int Ktot, float* dW, void* workspace,
                      hipStream_t st) {
  TnArgs a;
  int grid;
  tn_plan(M, Cout, Ktot, a, grid);
  a.GY = (const bf16_t*)GY; a.X = (const bf16_t*)X; a.slab = (float*)workspace;
  a.M = M; a.Cout = Cout; a.Cin = Cin; a.Ktot = Ktot; a.H = mode ? H : 1; a.W = mode ? W : 1;
  a.magic_w = (unsigned)((0x100000000ull + (unsigned)a.W - 1) / (unsigned)a.W);
  a.hw = a.H * a.W; a.r64 = 64 % a.hw; a.r32 = 32 % a.hw;
  a.x_bias = mode ? (unsigned)(W + 1) * Cin * 2 : 0;
  a.g_bytes = (unsigned)((size_t)M * Cout * 2);
  a.x_bytes = (unsigned)((size_t)M * Cin * 2 + a.x_bias);
  a.dbg = coin_p8_debug;
  if (mode == 1) {
    static bool set3 = false;
    if (!set3) { (void)hipFuncSetAttribute((const void*)conv_wgrad_p8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_IMG); set3 = true; }
    conv_wgrad_p8_kernel<true><<<grid, 512, P_IMG, st>>>(a);
  } else {
    static bool set1 = false;
    if (!set1) { (void)hipFuncSetAttribute((const void*)conv_wgrad_p8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, P_IMG); set1 = true; }
    conv_wgrad_p8_kernel<false><<<grid, 512, P_IMG, st>>>(a);
  }
  if (a.ntiles < a.cpx && a.cpx / a.ntiles >= 2)   // every tile is a leftover tile cut into >= 16 pieces
    tn_reduce_wide_kernel<<<a.ntiles * 256, 256, 0, st>>>(a.slab, dW, Cout, Ktot, a.tiles_k, a.ntiles, a.nkt, a.cpx);
  else
    tn_reduce_kernel<<<a.ntiles * 64, 256, 0, st>>>(a.slab, dW, Cout, Ktot, a.tiles_k, a.ntiles, a.nkt, a.cpx);
  return coin_launch_status();
}
