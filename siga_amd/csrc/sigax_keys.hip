// siga_amd/csrc/sigax_keys.hip -- locality keys of reads, for callers that decide themselves which GPU gets which reads
// (key-range sharding: bench.py --gpus N, siga_amd/sharding.py).  No counterpart in the reference (one process,
// reads in file order: src/overlap_builder.cpp:1113-1182); the key only decides placement, never a result.
//
// key(read) = min over the read's 16-mers of  hash(canonical 16-mer) << 16 | (L - 16 - off)
//   canonical = the smaller of the 16-mer and its reverse complement (2 bits per base, A C G T = 0 1 2 3, first base most
//               significant; any other byte counts as A), so a read and the reverse complement of its neighbour on the
//               genome agree;
//   hash      = bits 31..62 of canonical * 0x9E3779B97F4A7C15 (mod 2^64);
//   off       = where the read starts before that 16-mer on the canonical strand: the 16-mer's index i if it is the smaller
//               one itself, L - 16 - i if its reverse complement is.
// Reads sharing their minimizer (the 16-mer of the smallest hash) overlap the same <= 2L bases of the genome; sorted by key
// they are neighbours, ordered by where they start.  A read shorter than 16 bases has key 0.  The restatement the tests hold
// this against is siga_amd/sharding.py::locality_keys (torch, CPU).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/sigax.h"

int sigax_fail(int code, const char* fmt, ...);  // sigax_api.cpp

namespace {
constexpr unsigned KEY_K = 16;

__device__ __forceinline__ uint32_t base_code(unsigned char c) {
  // A C G T (either case) -> 0 1 2 3; anything else -> 0
  const unsigned u = c & 0xDFu;  // upper case
  return u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 0u;
}

// One lane per read, a byte per step: a lane walks its own 150-250 consecutive bytes (its cache lines stay in L1 for the
// next steps), and the reads of a wave lie side by side, so what a wave fetches is one contiguous stretch of the buffer --
// every byte comes from HBM once.  Once per read set, not per step.
__global__ __launch_bounds__(256) void k_locality_keys(const unsigned char* __restrict__ seqs, const unsigned long long* __restrict__ offs,
                                                       uint32_t n, unsigned long long* __restrict__ keys) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const unsigned long long b = offs[r], e = offs[r + 1];
  const uint32_t L = (uint32_t)(e - b);
  if (L < KEY_K) {
    keys[r] = 0ull;
    return;
  }
  const uint32_t span = L - KEY_K;  // last 16-mer index
  uint32_t f = 0u, g = 0u;
  unsigned long long best = ~0ull;
  for (uint32_t j = 0; j < L; ++j) {
    const uint32_t c = base_code(seqs[b + j]);
    f = (f << 2) | c;                    // 16 bases = 32 bits: the oldest falls off the top
    g = (g >> 2) | ((3u - c) << 30);     // reverse complement: the newest base is its first
    if (j + 1 >= KEY_K) {
      const uint32_t i = j + 1 - KEY_K;
      const bool fw = f <= g;
      const unsigned long long canon = fw ? f : g;
      const unsigned long long h = ((canon * 0x9E3779B97F4A7C15ull) >> 31) & 0xFFFFFFFFull;
      const uint32_t off = fw ? i : span - i;
      const unsigned long long k = (h << 16) | (unsigned long long)((span - off) & 0xFFFFu);
      best = k < best ? k : best;
    }
  }
  keys[r] = best;
}
}  // namespace

extern "C" int sigax_locality_keys(int device, const void* d_seqs, const void* d_offs, uint32_t n_reads, void* d_keys, void* stream) {
  if (n_reads == 0) return SIGAX_OK;
  if (!d_seqs || !d_offs || !d_keys) return sigax_fail(SIGAX_E_ARG, "NULL argument");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return sigax_fail(SIGAX_E_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  hipLaunchKernelGGL(k_locality_keys, dim3((n_reads + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)d_seqs,
                     (const unsigned long long*)d_offs, n_reads, (unsigned long long*)d_keys);
  e = hipGetLastError();
  if (e != hipSuccess) return sigax_fail(SIGAX_E_DEVICE, "k_locality_keys: %s", hipGetErrorString(e));
  return SIGAX_OK;
}
