// tools/gather_probe3.hip -- does a cache-policy bit (gfx950: sc0 / sc1 / nt) or a wider line change what the chip
// sustains for random line reads?  Quad-of-lanes pattern (one 16-byte load per lane, four lanes = one 64-byte
// granule), and an octet pattern (eight lanes = one aligned 128-byte line).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
#define LOADV(bits)                                                                       \
  asm volatile("global_load_dwordx4 %0, %1, off " bits "\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory")
template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* tab, u64 nlines, int iters, u32* sink) {
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  if (MODE == 7 || MODE == 8) {  // one lane per line: 8 (128-byte line) or 4 (64-byte line) loads from the same lane
    u64 seed = mix(gid * 0x9E3779B97F4A7C15ULL + 1);
    for (int it = 0; it < iters; ++it) {
      u64 g = seed % nlines; seed = mix(seed + it);
      const uint4* q = tab + g * (MODE == 7 ? 8 : 4);
      uint4 a = q[0], b = q[1], c = q[2], d = q[3];
      acc += a.x ^ b.y ^ c.z ^ d.w;
      if (MODE == 7) { uint4 e = q[4], f = q[5], h = q[6], k = q[7]; acc += e.x ^ f.y ^ h.z ^ k.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    return;
  }
  const int per = MODE == 6 ? 8 : 4;  // lanes per line
  for (int it = 0; it < iters; ++it) {
    u64 qs = mix((gid / per) * 0x9E3779B97F4A7C15ULL + 7 + (u64)it * 1315423911ULL);
    u64 g = qs % nlines;
    const uint4* p = tab + g * per + (gid % per);
    v4u a;
    if (MODE == 0 || MODE == 6) LOADV("");
    if (MODE == 1) LOADV("nt");
    if (MODE == 2) LOADV("sc0");
    if (MODE == 3) LOADV("sc1");
    if (MODE == 4) LOADV("sc0 sc1");
    if (MODE == 5) LOADV("sc1 nt");
    acc += a.x ^ a.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
  size_t sizes[] = {160ull << 20, 1280ull << 20};
  int iters = 64;
  u32* sink; hipMalloc(&sink, 64);
  const char* names[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc1 nt", "128B line", "lane/128B", "lane/64B"};
  for (size_t sz : sizes) {
    uint4* tab; if (hipMalloc(&tab, sz) != hipSuccess) { printf("alloc fail\n"); return 1; }
    hipMemset(tab, 1, sz);
    for (int mode = 0; mode < 9; ++mode) {
      const int wg = 8192;
      u64 nlines = sz / ((mode == 6 || mode == 7) ? 128 : 64);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        switch (mode) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 4: hipLaunchKernelGGL(probe<4>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 5: hipLaunchKernelGGL(probe<5>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 6: hipLaunchKernelGGL(probe<6>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          case 7: hipLaunchKernelGGL(probe<7>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
          default: hipLaunchKernelGGL(probe<8>, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      double lines = (double)wg * 256 / (mode == 6 ? 8 : mode >= 7 ? 1 : 4) * iters;
      printf("table %5zu MB %-10s: %8.3f ms  %8.1f Mlines/s  %7.1f GB/s\n", sz >> 20, names[mode], ms, lines / ms / 1e3,
             lines * ((mode == 6 || mode == 7) ? 128 : 64) / ms / 1e6);
    }
    hipFree(tab);
  }
  return 0;
}
