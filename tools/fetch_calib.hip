// tools/fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* report for the finder's access shapes?
// The microarch guide says FETCH_SIZE reads exactly half the bytes of a wide coalesced stream on gfx950 (128-byte
// requests tallied at 64 B) and that other access shapes are uncalibrated.  Four kernels over an 8 GiB table (far beyond
// L2 and the Infinity Cache; every line is touched once, lines visited through an odd-multiplier permutation):
//   k_calib_stream   16 B per lane, coalesced                                  -> bytes known exactly
//   k_calib_line1    one lane per 128-B line, ONE 16-B piece (offset 0)         -> one 64-B sector needed
//   k_calib_line2    one lane per 128-B line, pieces at offsets 0 and 80        -> both sectors (the two-step finder's shape)
//   k_calib_gran     one lane per 64-B granule, four 16-B pieces                -> one sector (the one-step finder's shape)
// Run under `rocprofv3 --kernel-trace --pmc <counter>` (one counter group per pass); tools/collect_profiles.py divides
// the counter by the known number of lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;

__global__ __launch_bounds__(256) void k_calib_stream(const uint4* __restrict__ tab, u64 n16, u32* sink) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  for (; i < n16; i += (u64)gridDim.x * 256) {
    uint4 a = tab[i];
    acc += a.x ^ a.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_line1(const uint4* __restrict__ tab, u64 nlines_mask, u64 count, u32* sink) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  for (; i < count; i += (u64)gridDim.x * 256) {
    const u64 line = (i * 0x9E3779B97F4A7C15ULL) & nlines_mask;
    uint4 a = tab[line * 8];
    acc += a.x ^ a.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_line2(const uint4* __restrict__ tab, u64 nlines_mask, u64 count, u32* sink) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  for (; i < count; i += (u64)gridDim.x * 256) {
    const u64 line = (i * 0x9E3779B97F4A7C15ULL) & nlines_mask;
    uint4 a = tab[line * 8], b = tab[line * 8 + 5];
    acc += a.x ^ b.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_gran(const uint4* __restrict__ tab, u64 ngran_mask, u64 count, u32* sink) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 acc = 0;
  for (; i < count; i += (u64)gridDim.x * 256) {
    const u64 g = (i * 0x9E3779B97F4A7C15ULL) & ngran_mask;
    const uint4* q = tab + g * 4;
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const u64 bytes = 8ull << 30;
  uint4* tab = nullptr;
  u32* sink = nullptr;
  if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tab, 1, bytes);
  hipDeviceSynchronize();
  const u64 nlines = bytes / 128, ngran = bytes / 64;
  const u64 count = 1ull << 25;  // lines / granules visited (4 GiB worth of 128-B lines: no line twice)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  const unsigned grid = 256 * 16;
  hipEventRecord(e0); hipLaunchKernelGGL(k_calib_stream, dim3(grid), dim3(256), 0, 0, (const uint4*)tab, (count * 128) / 16, sink); hipEventRecord(e1);
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("k_calib_stream bytes %llu  %.3f ms  %.1f GB/s\n", count * 128, ms, count * 128 / ms / 1e6);
  hipEventRecord(e0); hipLaunchKernelGGL(k_calib_line1, dim3(grid), dim3(256), 0, 0, (const uint4*)tab, nlines - 1, count, sink); hipEventRecord(e1);
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("k_calib_line1 lines %llu  %.3f ms  %.2f G lines/s\n", count, ms, count / ms / 1e6);
  hipEventRecord(e0); hipLaunchKernelGGL(k_calib_line2, dim3(grid), dim3(256), 0, 0, (const uint4*)tab, nlines - 1, count, sink); hipEventRecord(e1);
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("k_calib_line2 lines %llu  %.3f ms  %.2f G lines/s\n", count, ms, count / ms / 1e6);
  hipEventRecord(e0); hipLaunchKernelGGL(k_calib_gran, dim3(grid), dim3(256), 0, 0, (const uint4*)tab, ngran - 1, count, sink); hipEventRecord(e1);
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("k_calib_gran granules %llu  %.3f ms  %.2f G granules/s\n", count, ms, count / ms / 1e6);
  return 0;
}
