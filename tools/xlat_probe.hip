// tools/xlat_probe.hip -- does HOW a big table is allocated change what random 128-byte line reads from it sustain?
// The cooperative finder's pattern (eight lanes fetch the eight 16-byte pieces of one random line) on tables of 0.3, 6 and
// 25 GB -- one strand's two-step table at BASELINE configs[1], [2], [4] -- allocated with hipMalloc, with
// hipExtMallocWithFlags(hipDeviceMallocContiguous), and through the virtual-memory API (hipMemCreate + hipMemMap at the
// recommended granularity).  Address translation is what holds the finder at the large shapes (profiles/r02_finder_translation.txt:
// UTCL2 busy 0.99): a physically contiguous table could be mapped with larger fragments.
//   hipcc --offload-arch=gfx950 -O3 -o build/xlat_probe tools/xlat_probe.hip && gpurun -- build/xlat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__global__ __launch_bounds__(256) void probe(const uint4* tab, u64 nlines, int iters, u32* sink) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  for (int it = 0; it < iters; ++it) {
    const u64 g = mix((gid / 8) * 0x9E3779B97F4A7C15ULL + 7 + (u64)it * 1315423911ULL) % nlines;
    const uint4* p = tab + g * 8 + (gid % 8);
    v4u a;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
    acc += a.x ^ a.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void touch(uint4* tab, u64 n16) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) tab[i] = make_uint4((u32)i, 1, 2, 3);
}
static double run(const uint4* tab, size_t sz, u32* sink) {
  const int wg = 16384, iters = 64;
  const u64 nlines = sz / 128;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe, dim3(wg), dim3(256), 0, 0, tab, nlines, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  return (double)wg * 256 / 8 * iters / ms / 1e3;  // M lines/s
}
int main() {
  u32* sink; hipMalloc(&sink, 64);
  const size_t sizes[] = {302ull << 20, 6ull << 30, 25ull << 30};
  for (size_t sz : sizes) {
    for (int how = 0; how < 4; ++how) {
      uint4* tab = nullptr;
      hipMemGenericAllocationHandle_t h{};
      size_t gran = 0, mapped = 0;
      const char* name = how == 0 ? "hipMalloc" : how == 1 ? "hipExtMallocWithFlags(Contiguous)" : how == 2 ? "hipMemCreate+Map (recommended gran.)" : "hipMemCreate+Map (minimum gran.)";
      hipError_t e = hipSuccess;
      if (how == 0) e = hipMalloc((void**)&tab, sz);
      else if (how == 1) e = hipExtMallocWithFlags((void**)&tab, sz, hipDeviceMallocContiguous);
      else {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        e = hipMemGetAllocationGranularity(&gran, &prop, how == 2 ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum);
        if (e == hipSuccess) {
          mapped = (sz + gran - 1) / gran * gran;
          e = hipMemCreate(&h, mapped, &prop, 0);
        }
        if (e == hipSuccess) e = hipMemAddressReserve((void**)&tab, mapped, gran, nullptr, 0);
        if (e == hipSuccess) e = hipMemMap(tab, mapped, 0, h, 0);
        if (e == hipSuccess) {
          hipMemAccessDesc d{};
          d.location = prop.location;
          d.flags = hipMemAccessFlagsProtReadWrite;
          e = hipMemSetAccess(tab, mapped, &d, 1);
        }
      }
      if (e != hipSuccess) {
        printf("table %6zu MB %-38s: %s\n", sz >> 20, name, hipGetErrorString(e));
        (void)hipGetLastError();
        continue;
      }
      hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, tab, (u64)(sz / 16));
      hipDeviceSynchronize();
      const double r = run(tab, sz, sink);
      printf("table %6zu MB %-38s: %9.1f Mlines/s  %7.1f GB/s%s\n", sz >> 20, name, r, r * 128 / 1e3, gran ? "" : "");
      if (gran) printf("        (granularity %zu KB)\n", gran >> 10);
      fflush(stdout);
      if (how < 2) hipFree(tab);
      else { hipMemUnmap(tab, mapped); hipMemAddressFree(tab, mapped); hipMemRelease(h); }
    }
  }
  return 0;
}
