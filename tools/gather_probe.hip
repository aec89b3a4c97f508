// tools/gather_probe.hip -- what does MI355X sustain for dependent-free random reads of 64-byte granules?
// Variants: (0) one lane per granule, 4 x dwordx4 (k_find v1 pattern); (1) quad of lanes per granule, one
// dwordx4 each; (2) one lane per granule, only the first 16 B; tables of several sizes (Infinity-Cache resident
// and not).  Prints GB/s of granule bytes (64 B x granules touched).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* tab, u64 ngran, int iters, u32* sink) {
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  u64 seed = mix(gid * 0x9E3779B97F4A7C15ULL + 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      u64 g = seed % ngran; seed = mix(seed + it);
      const uint4* q = tab + g * 4;
      uint4 a = q[0], b = q[1], c = q[2], d = q[3];
      acc += a.x ^ b.y ^ c.z ^ d.w;
    } else if (MODE == 1) {
      u64 qs = mix((gid >> 2) * 0x9E3779B97F4A7C15ULL + 7 + (u64)it * 1315423911ULL);
      u64 g = qs % ngran;
      uint4 a = tab[g * 4 + (gid & 3)];
      acc += a.x ^ a.w;
    } else if (MODE == 2) {
      u64 g = seed % ngran; seed = mix(seed + it);
      uint4 a = tab[g * 4];
      acc += a.x ^ a.w;
    } else {  // MODE 3: two granules per lane per iteration (both rank positions), 8 loads in flight
      u64 g = seed % ngran; seed = mix(seed + it);
      u64 g2 = seed % ngran; seed = mix(seed + it);
      const uint4* q = tab + g * 4; const uint4* r = tab + g2 * 4;
      uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = r[0], f = r[1], h = r[2], k = r[3];
      acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ h.z ^ k.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  size_t sizes[] = {160ull << 20, 1ull << 30, 8ull << 30};
  int iters = 64;
  u32* sink; hipMalloc(&sink, 64);
  for (size_t sz : sizes) {
    uint4* tab; if (hipMalloc(&tab, sz) != hipSuccess) { printf("alloc fail\n"); return 1; }
    hipMemset(tab, 1, sz);
    u64 ngran = sz / 64;
    for (int mode = 0; mode < 4; ++mode) {
      for (int wg : {2048, 8192}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wg), dim3(256), 0, 0, tab, ngran, iters, sink);
          if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wg), dim3(256), 0, 0, tab, ngran, iters, sink);
          if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(wg), dim3(256), 0, 0, tab, ngran, iters, sink);
          if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(wg), dim3(256), 0, 0, tab, ngran, iters, sink);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        double lanes = (double)wg * 256;
        double gran = mode == 1 ? lanes / 4 * iters : (mode == 3 ? lanes * iters * 2 : lanes * iters);
        printf("table %5zu MB mode %d wg %5d: %8.3f ms  %8.1f Mgran/s  %7.1f GB/s (64B granules)\n", sz >> 20, mode, wg, ms,
               gran / ms / 1e3, gran * 64 / ms / 1e6);
      }
    }
    hipFree(tab);
  }
  return 0;
}
