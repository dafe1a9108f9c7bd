// Micro-benchmark (development tool): how fast does ONE workgroup (512 threads = a whole CU at this kernel's register budget) retire a
// 128 KiB burst of 16-byte stores, with G workgroups doing the same at the same time?   hipcc --offload-arch=gfx950 -O3 store_burst.hip
// Every workgroup writes `bursts` tiles of 256 rows x 512 B (row pitch `pitch` bytes: the GEMM epilogue's pattern) and stamps
// s_memtime before the burst and after `s_waitcnt vmcnt(0)`.  Output: median cycles per burst and GB/s per CU / aggregate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void burst_kernel(char* out, size_t pitch, int bursts, int tiles_per_row, long long* stamps, int aux_nt, int gap_ticks, int stagger) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const u32x4 v = {(unsigned)tid, (unsigned)b, 3u, 4u};
  long long t_acc = 0;
  auto nap = [](long long ticks) { const long long t = __builtin_amdgcn_s_memrealtime(); while ((long long)__builtin_amdgcn_s_memrealtime() - t < ticks) __builtin_amdgcn_s_sleep(8); };
  if (stagger) nap((long long)gap_ticks * ((b >> 3) % stagger) / stagger);   // start offsets spread over one (gap + burst) period
  for (int k = 0; k < bursts; ++k) {
    const int tile = b + k * gridDim.x;
    const int tm = tile / tiles_per_row, tn = tile % tiles_per_row;
    char* base = out + (size_t)tm * 256 * pitch + (size_t)tn * 512;
    if (gap_ticks) nap(gap_ticks);   // stands for the tile's main loop (no memory traffic); 10 ns ticks
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      // thread -> (row = q * 16 + tid / 32, 16-byte chunk tid % 32): 16 rows x 512 B per wave-instruction group
      char* p = base + (size_t)(q * 16 + (tid >> 5)) * pitch + (tid & 31) * 16;
      if (aux_nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
      else *reinterpret_cast<u32x4*>(p) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    t_acc += __builtin_amdgcn_s_memtime() - t0;
  }
  if (tid == 0) stamps[b] = t_acc;
}

int main(int argc, char** argv) {
  const int bursts = 64;
  const size_t pitch = argc > 1 ? atoi(argv[1]) : 4096;   // bytes per output row (N * 2)
  const int tiles_per_row = (int)(pitch / 512);
  char* out;
  long long* st;
  const size_t bytes = (size_t)256 * bursts * 256 * 512 + (1 << 20);
  hipMalloc(&out, bytes);
  hipMalloc(&st, 256 * sizeof(long long));
  for (int gap : {0, 1200})                // 0: back-to-back bursts; 1200 ticks = 12 us of "main loop" between two bursts
    for (int stagger : {0, 4, 16})
      for (int G : {1, 64, 256}) {
        if (gap == 0 && stagger) continue;
        burst_kernel<<<G, 512>>>(out, pitch, bursts, tiles_per_row, st, 0, gap, stagger);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        burst_kernel<<<G, 512>>>(out, pitch, bursts, tiles_per_row, st, 0, gap, stagger);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(G);
        hipMemcpy(h.data(), st, G * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("gap=%4d stagger=%2d G=%3d  median cycles/burst %.0f (min %.0f max %.0f)  kernel %.3f ms (%.2f us per period)\n", gap, stagger, G,
               (double)h[G / 2] / bursts, (double)h[0] / bursts, (double)h[G - 1] / bursts, ms, ms * 1e3 / bursts);
      }
  return 0;
}
