// Which lane -> address patterns of ds_read_b128 are conflict-free on gfx950?  One workgroup per CU, 8 waves, each wave issues
// ITER x 8 independent ds_read_b128 with a given pattern; cycles per instruction per CU from s_memtime.  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_b128_probe.hip -o tools/probes/lds_b128_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pattern -> 16-byte unit index inside a 16-KB region, for lane (li = lane & 15, g = lane >> 4), read j
__device__ int unit_of(int pat, int lane, int j) {
  const int li = lane & 15, g = lane >> 4;
  switch (pat) {
    case 0: return j * 64 + lane;                                        // linear: unit = lane (W fragments)
    case 1: return (j * 16 + li) * 8 + ((g) ^ (li & 7));                 // 128-B rows, chunk ^ (row & 7)  (A image today, kk = 0)
    case 2: return (j * 16 + li) * 8 + ((4 + g) ^ (li & 7));             // same, kk = 1
    case 3: return (j * 16 + li) * 4 + g;                                // 64-B rows, no swizzle
    case 4: return (j * 16 + li) * 4 + (g ^ (li >> 2));                  // 64-B rows, chunk ^ (row >> 2)
    case 5: return (j * 16 + li) * 4 + (g ^ (li & 3));                   // 64-B rows, chunk ^ (row & 3)
    case 6: return (j * 16 + li) * 4 + (g ^ ((li >> 1) & 3));            // 64-B rows, chunk ^ ((row >> 1) & 3)
    case 7: return (j * 16 + li) * 8 + g;                                // 128-B rows, no swizzle
    case 8: return ((j * 16 + li) * 16) & 1023;                          // 256-B stride: every lane of a phase on the same banks
    case 9: return j * 64 + li * 4 + g;                                  // 64-B rows == linear in (li, g) order (fragment = 1 KB contiguous)
    default: return lane;
  }
}

__global__ __launch_bounds__(512) void probe(int pat, int iters, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024 * 4; i += 512) lds[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  int u[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) u[j] = unit_of(pat, lane, j) & 4095;
  u32x4 acc = (u32x4){0u, 0u, 0u, 0u};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds[u[j]];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
    asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned long long* d_out; unsigned* d_sink;
  hipMalloc(&d_out, 256 * 8); hipMalloc(&d_sink, 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  const char* names[] = {"linear unit=lane", "128B rows chunk^(row&7) kk0", "128B rows chunk^(row&7) kk1", "64B rows no swizzle", "64B rows chunk^(row>>2)",
                         "64B rows chunk^(row&3)", "64B rows chunk^((row>>1)&3)", "128B rows no swizzle", "256B stride (worst)", "64B rows fragment-contiguous"};
  for (int pat = 0; pat < 10; ++pat) {
    for (int rep = 0; rep < 2; ++rep) {
      probe<<<256, 512, 65536, 0>>>(pat, iters, d_out, d_sink);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d_out, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9: report relative numbers
    printf("{\"pattern\": %d, \"name\": \"%s\", \"ticks_per_block\": %.0f, \"ticks_per_wave_instr_x1000\": %.3f}\n", pat, names[pat], s / 256,
           s / 256 / (iters * 8.0) * 1000.0);
  }
  return 0;
}
