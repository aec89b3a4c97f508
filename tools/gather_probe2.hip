// tools/gather_probe2.hip -- dependent random reads (the next address needs the loaded data, like a backward-search
// step): how does the sustained granule rate depend on table size, resident waves per CU and chains per lane?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
// CH chains per lane, each step: two granules (4 x dwordx4 each) whose data feed the next step's addresses
template <int CH>
__global__ __launch_bounds__(256) void probe(const uint4* tab, u64 ngran, int steps, u32* sink) {
  extern __shared__ char pad[];
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 s[CH];
  for (int c = 0; c < CH; ++c) s[c] = mix(gid * CH + c + 1);
  u32 acc = 0;
  for (int it = 0; it < steps; ++it) {
    uint4 v[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      u64 g = s[c] % ngran, g2 = (s[c] >> 20) % ngran;
      const uint4* q = tab + g * 4; const uint4* r = tab + g2 * 4;
      v[c][0] = q[0]; v[c][1] = q[1]; v[c][2] = q[2]; v[c][3] = q[3];
      v[c][4] = r[0]; v[c][5] = r[1]; v[c][6] = r[2]; v[c][7] = r[3];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      u32 x = v[c][0].x ^ v[c][1].y ^ v[c][2].z ^ v[c][3].w ^ v[c][4].x ^ v[c][5].y ^ v[c][6].z ^ v[c][7].w;
      acc += x;
      s[c] = mix(s[c] + x + it);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
  size_t sizes[] = {160ull << 20, 1280ull << 20};
  u32* sink; (void)hipMalloc(&sink, 64);
  for (size_t sz : sizes) {
    uint4* tab; if (hipMalloc(&tab, sz) != hipSuccess) return 1;
    (void)hipMemset(tab, 0, sz);
    u64 ngran = sz / 64;
    for (int ch : {1, 2}) for (unsigned lds : {0u, 36000u, 60000u, 100000u}) {
      int steps = 150; unsigned wg = 16384 / ch;
      auto launch = [&]() {
        if (ch == 1) hipLaunchKernelGGL(probe<1>, dim3(wg), dim3(256), lds, 0, tab, ngran, steps, sink);
        else hipLaunchKernelGGL(probe<2>, dim3(wg), dim3(256), lds, 0, tab, ngran, steps, sink);
      };
      launch(); (void)hipDeviceSynchronize();
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0); launch(); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 2;
      double gran = (double)wg * 256 * ch * steps * 2;
      printf("table %5zu MB chains/lane %d lds %6u: %8.3f ms  %7.1f Ggran/s\n", sz >> 20, ch, lds, ms, gran / ms / 1e6);
    }
    (void)hipFree(tab);
  }
  return 0;
}
