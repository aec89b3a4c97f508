// siga_amd/csrc/sigax_index_build.hip -- `siga index` on the GPU: suffix order + BWT of a read set.
//
// Replaces, for one strand, SuffixArrayBuilder "sais2" (src/suffix_array_builder.cpp:472-674) + BWT(sa, reads)
// (src/bwt.cpp:7-32) + the .sai rows (src/suffix_array.cpp:17-44) as `siga index` chains them (src/indexer.cpp:80-104).
// The order of record is the plain suffix array of  T = r0 $ r1 $ ... r(n-1) $ <end>  with ONE '$' symbol that is
// smaller than A,C,G,T, comparisons running on past a '$' into the next read, and the end of the text smallest
// (SURVEY.md App. C "model B"; non-ACGT bases rank as '$', src/alphabet.h:19-39).
//
// MI355X-first design (nothing here resembles induced sorting): the text lives in HBM as a 3-bit big-endian stream, so
// the next 21 symbols of any suffix are two aligned 8-byte loads and one funnel shift, and lexicographic order of
// suffix prefixes is integer order of those 63-bit words.
//   1. histogram of 5-symbol prefixes (LDS) -> groups of consecutive prefixes that fit the sort workspace;
//   2. per group: collect (21-symbol key, position), one device radix sort (rocPRIM), equal-key segments;
//   3. a segment of <= 64 suffixes (at 30-50x coverage: the reads covering one genome position) is finished by ONE wave
//      in registers: rounds of "next 19 symbols" + a 64-lane rank sort until all lanes differ;
//      larger segments (repeats, deep coverage) take further global rounds keyed by (segment, next 10 symbols);
//   4. BWT symbol = symbol before each sorted position; .sai rows = read starts in that order; RL units with the
//      31-cap of src/bwt.cpp:17 are cut on the device.
// The unique end-of-text symbol guarantees termination; pathological inputs (thousands of identical reads in a row)
// cost many rounds and give up with SIGAX_E_CAPACITY after a bound, upon which the host falls back to its SA-IS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "sigax_kernels.h"

typedef unsigned long long u64;
typedef unsigned int u32;

int sigax_fail(int code, const char* fmt, ...);  // sigax_api.cpp

namespace {

#define IB_TRY(expr)                                                                                           \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) return sigax_fail(SIGAX_E_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// frees everything it was given when it goes out of scope (every early return of the builder)
// Device memory of the builders.  hipMalloc hands out CLEARED memory and hipFree waits for the device: with one
// hipMalloc / hipFree pair per buffer the second strand of `siga index` at BASELINE configs[2] took 3.0 s for 0.85 s of
// kernels, the first -- on memory the process had not dirtied yet -- 0.8 s (profiles/r03_index_kernel_stats.csv).  So a
// block that is released goes to a per-process cache and is handed out again (best fit on the same device, at most
// half as large again as asked for); the cache is emptied when the last pool or build session (sigax_build_session:
// the two strands of one `siga index`) ends, or when an allocation fails.  Nothing here relies on cleared memory:
// SIGAX_POOL_POISON=1 fills every block with 0xA5 before it is handed out (the index-build tests run with it).
namespace devcache {
struct Block {
  void* p;
  size_t bytes;
  int device;
};
static std::mutex mu;
static std::vector<Block> spare;
static int users = 0;
static void trim_locked() {
  for (const Block& b : spare) hipFree(b.p);
  spare.clear();
}
static void enter() {
  std::lock_guard<std::mutex> g(mu);
  ++users;
}
static void leave() {
  std::lock_guard<std::mutex> g(mu);
  if (--users <= 0) {
    users = 0;
    trim_locked();
  }
}
}  // namespace devcache

struct DevPool {
  std::vector<devcache::Block> live;
  int device = 0;
  DevPool() {
    (void)hipGetDevice(&device);
    devcache::enter();
  }
  DevPool(const DevPool&) = delete;
  DevPool& operator=(const DevPool&) = delete;
  ~DevPool() {
    {
      std::lock_guard<std::mutex> g(devcache::mu);
      for (const devcache::Block& b : live)
        if (b.p) devcache::spare.push_back(b);
    }
    devcache::leave();
  }
  template <typename T> hipError_t alloc(T** out, size_t count) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    bytes = bytes < ((size_t)1 << 20) ? (bytes + 255) & ~(size_t)255 : (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* p = nullptr;
    size_t got = bytes;
    {
      std::lock_guard<std::mutex> g(devcache::mu);
      size_t best = (size_t)-1;
      for (size_t k = 0; k < devcache::spare.size(); ++k) {
        const devcache::Block& b = devcache::spare[k];
        if (b.device != device || b.bytes < bytes || b.bytes > bytes + bytes / 2 + ((size_t)1 << 20)) continue;
        if (best == (size_t)-1 || b.bytes < devcache::spare[best].bytes) best = k;
      }
      if (best != (size_t)-1) {
        p = devcache::spare[best].p;
        got = devcache::spare[best].bytes;
        devcache::spare.erase(devcache::spare.begin() + (long)best);
      }
    }
    hipError_t e = hipSuccess;
    if (!p) {
      e = hipMalloc(&p, bytes);
      if (e != hipSuccess) {  // memory short: what the cache holds goes back first
        (void)hipGetLastError();
        {
          std::lock_guard<std::mutex> g(devcache::mu);
          devcache::trim_locked();
        }
        e = hipMalloc(&p, bytes);
      }
    }
    if (e == hipSuccess) {
      static const bool poison = getenv("SIGAX_POOL_POISON") != nullptr;
      if (poison) e = hipMemset(p, 0xA5, got);
      live.push_back({p, got, device});
    } else {
      p = nullptr;
    }
    *out = (T*)p;
    return e;
  }
  void release(void* p) {
    if (!p) return;
    for (devcache::Block& q : live)
      if (q.p == p) {
        // whatever still reads or writes the block has been waited for by the callers (they synchronise before they
        // release); the next user's work is ordered after it on the same (null) stream anyway
        std::lock_guard<std::mutex> g(devcache::mu);
        devcache::spare.push_back(q);
        q.p = nullptr;
      }
  }
};

extern "C" void sigax_build_session(int open) {
  if (open) devcache::enter();
  else devcache::leave();
}

// ---- the packed text ------------------------------------------------------------------------------------------
// symbol i occupies stream bits [3i, 3i+3), most significant first; codes: 0 end of text, 1 '$' (and non-ACGT), 2..5 ACGT
__device__ __forceinline__ u64 key63(const u64* __restrict__ P, u64 p) {
  const u64 b = 3 * p, w = b >> 6;
  const u32 s = (u32)b & 63u;
  const u64 hi = P[w], lo = P[w + 1];
  const u64 v = s ? ((hi << s) | (lo >> (64u - s))) : hi;
  return v >> 1;  // 21 symbols, the first in bits 62..60
}
__device__ __forceinline__ u32 text_code(unsigned char ch) {
  return ch == 'A' ? 2u : ch == 'C' ? 3u : ch == 'G' ? 4u : ch == 'T' ? 5u : 1u;
}

// one thread per 64-bit word of the stream; position of read r's first symbol = offs[r] + r
__global__ __launch_bounds__(256) void k_pack_text(const unsigned char* __restrict__ seqs, const u64* __restrict__ offs, u64 n_reads,
                                                   u64 n, int reverse, u64* __restrict__ P, u64 nw) {
  const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
  if (w >= nw) return;
  const u64 b0 = w * 64, i0 = b0 / 3, i1 = (b0 + 63) / 3;
  u64 word = 0;
  if (i0 < n) {
    u64 lo = 0, hi = n_reads;  // start(lo) <= i0 < start(hi), start(n_reads) = n
    while (hi - lo > 1) {
      const u64 mid = (lo + hi) >> 1;
      if (offs[mid] + mid <= i0) lo = mid; else hi = mid;
    }
    u64 r = lo, rs = offs[r] + r, rb = offs[r], len = offs[r + 1] - rb;
    for (u64 i = i0; i <= i1; ++i) {
      u64 c = 0;
      if (i < n) {
        u64 j = i - rs;
        while (j > len) {  // j == len is the read's '$'
          ++r;
          rb = offs[r];
          rs = rb + r;
          len = offs[r + 1] - rb;
          j = i - rs;
        }
        c = j < len ? text_code(seqs[rb + (reverse ? len - 1 - j : j)]) : 1u;
      }
      const long long sh = (long long)(b0 + 61) - (long long)(3 * i);  // -2 .. 63
      word |= sh >= 0 ? (c << sh) : (c >> (-sh));
    }
  }
  P[w] = word;
}

__global__ __launch_bounds__(256) void k_mark_starts(const u64* __restrict__ offs, u64 n_reads, u32* bits) {
  const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_reads) return;
  const u64 s = offs[r] + r;
  atomicOr(&bits[s >> 5], 1u << (s & 31u));
}

// ---- step 1: histogram of 5-symbol prefixes ---------------------------------------------------------------------
#define IB_BINS 32768
__global__ __launch_bounds__(1024) void k_hist5(const u64* __restrict__ P, u64 n, u64* hist) {
  __shared__ u32 h[IB_BINS];
  for (u32 i = threadIdx.x; i < IB_BINS; i += 1024) h[i] = 0;
  __syncthreads();
  for (u64 p = (u64)blockIdx.x * 1024 + threadIdx.x; p < n; p += (u64)gridDim.x * 1024) atomicAdd(&h[key63(P, p) >> 48], 1u);
  __syncthreads();
  for (u32 i = threadIdx.x; i < IB_BINS; i += 1024)
    if (h[i]) atomicAdd(&hist[i], (u64)h[i]);
}

// ---- step 2: the suffixes of one group of prefixes, with their first 21 symbols ------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(1024) void k_collect(const u64* __restrict__ P, u64 n, u32 blo, u32 bhi, u64* __restrict__ K,
                                                  IdxT* __restrict__ I, u64* count) {
  __shared__ u32 lcount;
  __shared__ u64 gbase;
  const u32 lane = threadIdx.x & 63u;
  const u64 lt = (1ull << lane) - 1ull;
  const u64 tile = 1024ull * 8;
  for (u64 t0 = (u64)blockIdx.x * tile; t0 < n; t0 += (u64)gridDim.x * tile) {  // block-uniform trip count
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    u64 kk[8];
    u32 woff[8];
    u32 inmask = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u64 p = t0 + (u64)k * 1024 + threadIdx.x;
      bool in = p < n;
      kk[k] = in ? key63(P, p) : 0ull;
      const u32 b = (u32)(kk[k] >> 48);
      in = in && b >= blo && b < bhi;
      const u64 m = __ballot(in);
      u32 base = 0;
      if (m) {
        if (lane == 0) base = atomicAdd(&lcount, (u32)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
      }
      woff[k] = base + (u32)__popcll(m & lt);
      if (in) inmask |= 1u << k;
    }
    __syncthreads();
    if (threadIdx.x == 0) gbase = atomicAdd(count, (u64)lcount);
    __syncthreads();
    const u64 gb = gbase;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((inmask >> k) & 1u) {
        K[gb + woff[k]] = kk[k];
        I[gb + woff[k]] = (IdxT)(t0 + (u64)k * 1024 + threadIdx.x);
      }
    __syncthreads();
  }
}

// ---- step 3a: one wave finishes a segment of 2..64 suffixes that agree on their first `depth` symbols -------------
__device__ __forceinline__ u64 readlane_u64(u64 v, u32 l) {
  const u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
struct FinishSh {
  u64 comp[64];
  u64 pos[64];
};
// descriptors (dstart, dlen) as rocPRIM's run_length_encode_non_trivial_runs leaves them; `home` (optional) maps a
// descriptor's start to the segment's first slot in I (segments are contiguous in I).  err[0] = rounds bound hit.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_seg_finish(const u64* __restrict__ P, u64 n, IdxT* __restrict__ I, const u32* __restrict__ dstart,
                                                   const u32* __restrict__ dlen, const u32* __restrict__ home, u32 ndesc, u32 depth,
                                                   u32 max_rounds, u32* err) {
  __shared__ FinishSh shm[4];
  const u32 wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  FinishSh& sh = shm[wid];
  const u64 wave = (u64)blockIdx.x * 4 + wid, nwaves = (u64)gridDim.x * 4;
  for (u64 c0 = wave * 64; c0 < ndesc; c0 += nwaves * 64) {
    const u64 di = c0 + lane;
    u32 mylen = 0, mystart = 0;
    if (di < ndesc) {
      mylen = dlen[di];
      mystart = dstart[di];
      if (home) mystart = home[mystart];
    }
    u64 todo = __ballot(mylen >= 2 && mylen <= 64);
    while (todo) {
      const u32 src = (u32)__builtin_ctzll(todo);
      todo &= todo - 1;
      const u32 len = __builtin_amdgcn_readlane(mylen, src);
      const u64 start = __builtin_amdgcn_readlane(mystart, src);
      const bool act = lane < len;
      u64 p = act ? (u64)I[start + lane] : 0ull;
      u32 cls = 0;  // first sorted slot of the lane's class of still-equal suffixes
      u32 d = depth, rounds = 0;
      for (;;) {
        u64 q = p + d;
        q = q > n ? n : q;  // a lane that is alone in its class may run past the end: any key will do for it
        const u64 comp = act ? (((u64)cls << 57) | (key63(P, q) >> 6)) : ~0ull;  // class, then the next 19 symbols
        u32 rank = 0;
        for (u32 j = 0; j < len; ++j) {
          const u64 cj = readlane_u64(comp, j);
          rank += (cj < comp || (cj == comp && j < lane)) ? 1u : 0u;
        }
        __builtin_amdgcn_wave_barrier();
        if (act) {
          sh.comp[rank] = comp;
          sh.pos[rank] = p;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u64 mine = ~0ull, prev = ~0ull;
        if (act) {
          mine = sh.comp[lane];
          p = sh.pos[lane];
          prev = lane ? sh.comp[lane - 1] : ~mine;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u64 heads = __ballot(act && mine != prev);
        if ((u32)__popcll(heads) == len) break;
        const u64 upto = heads & ((lane == 63 ? 0ull : (2ull << lane)) - 1ull);  // heads at or below this lane
        cls = act ? 63u - (u32)__builtin_clzll(upto) : 0u;
        d += 19;
        if (++rounds > max_rounds) {
          if (lane == 0) atomicOr(err, 1u);
          break;
        }
      }
      if (act) I[start + lane] = (IdxT)p;
    }
  }
}

// ---- step 3b: segments of more than 64 suffixes take further global rounds -----------------------------------------
struct LargeLen {
  __device__ u64 operator()(u32 c) const { return c > 64u ? (u64)c : 0ull; }
};
// one wave per large segment: copy it into the compacted arrays.  src_pos == NULL: the segment sits in I itself.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_large_compact(const IdxT* __restrict__ Isrc, const u32* __restrict__ src_pos, const u32* __restrict__ dstart,
                                                      const u32* __restrict__ dlen, const u64* __restrict__ doff, u32 ndesc, IdxT* __restrict__ Ic,
                                                      u32* __restrict__ pos, u32* __restrict__ seg) {
  const u32 wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const u64 wave = (u64)blockIdx.x * 4 + wid, nwaves = (u64)gridDim.x * 4;
  for (u64 c0 = wave * 64; c0 < ndesc; c0 += nwaves * 64) {
    const u64 di = c0 + lane;
    u32 mylen = 0, mystart = 0;
    u64 myoff = 0;
    if (di < ndesc) {
      mylen = dlen[di];
      mystart = dstart[di];
      myoff = doff[di];
    }
    u64 todo = __ballot(mylen > 64);
    while (todo) {
      const u32 src = (u32)__builtin_ctzll(todo);
      todo &= todo - 1;
      const u32 len = __builtin_amdgcn_readlane(mylen, src), start = __builtin_amdgcn_readlane(mystart, src);
      const u64 off = readlane_u64(myoff, src);
      for (u32 t = lane; t < len; t += 64) {
        Ic[off + t] = Isrc[start + t];
        pos[off + t] = src_pos ? src_pos[start + t] : start + t;
        seg[off + t] = (u32)off;
      }
    }
  }
}
template <typename IdxT>
__global__ __launch_bounds__(256) void k_large_keys(const u64* __restrict__ P, u64 n, const IdxT* __restrict__ Ic, const u32* __restrict__ seg, u64 mc,
                                                   u32 depth, u64* __restrict__ Kc) {
  const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
  if (j >= mc) return;
  u64 q = (u64)Ic[j] + depth;
  q = q > n ? n : q;
  Kc[j] = ((u64)seg[j] << 32) | ((key63(P, q) >> 33) << 2);  // segment, then the next 10 symbols
}
template <typename IdxT>
__global__ __launch_bounds__(256) void k_large_writeback(const IdxT* __restrict__ Ic, const u32* __restrict__ pos, u64 mc, IdxT* __restrict__ I) {
  const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
  if (j < mc) I[pos[j]] = Ic[j];
}

// ---- step 4: BWT symbols and .sai rows of a sorted group ------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(256) void k_emit(const u64* __restrict__ P, const u32* __restrict__ startbits, const u64* __restrict__ offs, u64 n_reads,
                                             const IdxT* __restrict__ I, u64 m, unsigned char* __restrict__ B, u32* __restrict__ R) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const u64 p = (u64)I[i];
  const u32 code = p == 0 ? 1u : (u32)(key63(P, p - 1) >> 60);
  B[i] = (unsigned char)(code - 1u);
  u32 rid = 0xFFFFFFFFu;
  if ((startbits[p >> 5] >> (p & 31u)) & 1u) {  // a full-read suffix: SA row with j == 0 (src/suffix_array_builder.cpp:520-531)
    u64 lo = 0, hi = n_reads;
    while (hi - lo > 1) {
      const u64 mid = (lo + hi) >> 1;
      if (offs[mid] + mid <= p) lo = mid; else hi = mid;
    }
    rid = (u32)lo;
  }
  R[i] = rid;
}
struct IsRow {
  __device__ bool operator()(u32 v) const { return v != 0xFFFFFFFFu; }
};

// RL units (src/rlstring.h:10-63) with the 31-cap of src/bwt.cpp:17: a unit starts where the symbol changes or where
// the run's offset is a multiple of 31.  One thread per 64 symbols; count pass, scan, write pass.
#define RL_CHUNK 64
// where the run holding a chunk's first symbol starts: per chunk the last position (+1) at which the symbol changes, then
// an inclusive max-scan over the chunks (a long run -- a homopolymer region at depth -- must not be walked back per chunk)
__global__ __launch_bounds__(256) void k_rl_heads(const unsigned char* __restrict__ B, u64 n, u64* __restrict__ lasthead) {
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 s = t * RL_CHUNK;
  if (s >= n) return;
  const u64 e = s + RL_CHUNK < n ? s + RL_CHUNK : n;
  u64 last = 0;  // 0 = no run starts in this chunk
  for (u64 i = s; i < e; ++i)
    if (i == 0 || B[i] != B[i - 1]) last = i + 1;
  lasthead[t] = last;
}
struct MaxU64 {
  __device__ u64 operator()(u64 a, u64 b) const { return a > b ? a : b; }
};
template <bool WRITE>
__global__ __launch_bounds__(256) void k_rl_units(const unsigned char* __restrict__ B, u64 n, const u64* __restrict__ headscan, u32* __restrict__ cnt,
                                                 const u64* __restrict__ uoff, unsigned char* __restrict__ runs) {
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 s = t * RL_CHUNK;
  if (s >= n) return;
  const u64 e = s + RL_CHUNK < n ? s + RL_CHUNK : n;
  // offset (mod 31) of the chunk's first symbol inside its run
  u32 o = 0;
  if (s > 0 && B[s] == B[s - 1]) o = (u32)((s - (headscan[t - 1] - 1)) % 31);
  u32 count = 0;
  u64 w = WRITE ? uoff[t] : 0;
  unsigned char prev = B[s];
  for (u64 i = s; i < e; ++i) {
    const unsigned char c = B[i];
    if (c != prev) {
      o = 0;
      prev = c;
    }
    if (o == 0) {
      if (WRITE) {
        u32 len = 1;
        while (len < 31 && i + len < n && B[i + len] == c) ++len;
        runs[w++] = (unsigned char)((c << 5) | len);
      }
      ++count;
    }
    o = o + 1 == 31 ? 0 : o + 1;
  }
  if (!WRITE) cnt[t] = count;
}

// ---- .bwt payload -> rank granules (fm_layout.h) on the device ---------------------------------------------------------
// RL units (src/rlstring.h:10-63): rank << 5 | count.  One thread per 32-symbol chunk finds the run holding its first
// symbol (binary search in the scanned run lengths), walks the runs and writes the chunk's three bit planes; the four
// chunks of a granule add up their A,C,G,T counts, which four scans turn into the granules' counters.
__global__ __launch_bounds__(256) void k_run_lengths(const unsigned char* __restrict__ runs, u64 n_runs, u32* __restrict__ cnt, u32* bad) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_runs) return;
  const u32 u = runs[i];
  cnt[i] = u & 31u;
  if ((u >> 5) > 4u) atomicOr(bad, 1u);
}
__global__ __launch_bounds__(256) void k_decode_chunks(const unsigned char* __restrict__ runs, const u64* __restrict__ roffs, u64 n_runs, u64 nsym,
                                                      u64 nchunks, u32* __restrict__ gran, u32* __restrict__ gcnt, u64 ngran) {
  const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 p0 = 0, p1 = 0, p2 = 0;
  const u64 s = c * 32;
  if (c < nchunks && s < nsym) {
    u64 lo = 0, hi = n_runs;  // roffs[lo] <= s < roffs[hi]
    while (hi - lo > 1) {
      const u64 mid = (lo + hi) >> 1;
      if (roffs[mid] <= s) lo = mid; else hi = mid;
    }
    u64 r = lo;
    u64 left = roffs[r + 1] - s;  // symbols of run r from s on
    u32 sym = runs[r] >> 5;
    const u32 take = (u32)(nsym - s < 32 ? nsym - s : 32);
    u32 b = 0;
    while (b < take) {
      while (left == 0) {
        ++r;
        left = runs[r] & 31u;
        sym = runs[r] >> 5;
      }
      const u32 k = (u32)(left < (u64)(take - b) ? left : (u64)(take - b));
      const u32 m = (k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u)) << b;
      if (sym & 1u) p0 |= m;
      if (sym & 2u) p1 |= m;
      if (sym & 4u) p2 |= m;
      b += k;
      left -= k;
    }
  }
  if (c < nchunks) {
    u32* q = gran + c * 4;
    q[1] = p0; q[2] = p1; q[3] = p2;
  }
  // A = p0 & ~p1, C = p1 & ~p0, G = p0 & p1, T = p2; sum over the granule's four chunks (adjacent lanes)
  u32 a = __popc(p0 & ~p1), cc = __popc(p1 & ~p0), g = __popc(p0 & p1), t = __popc(p2);
  a += __shfl_xor(a, 1, 64); cc += __shfl_xor(cc, 1, 64); g += __shfl_xor(g, 1, 64); t += __shfl_xor(t, 1, 64);
  a += __shfl_xor(a, 2, 64); cc += __shfl_xor(cc, 2, 64); g += __shfl_xor(g, 2, 64); t += __shfl_xor(t, 2, 64);
  if ((c & 3u) == 0 && (c >> 2) < ngran) {
    const u64 gi = c >> 2;
    gcnt[gi] = a; gcnt[ngran + gi] = cc; gcnt[2 * ngran + gi] = g; gcnt[3 * ngran + gi] = t;
  }
}
// counter `col` of every granule (relative to its superblock in wide mode) and the superblock table
__global__ __launch_bounds__(256) void k_granule_headers(const u64* __restrict__ goffs, u64 ngran, u32 col, int wide, u32* __restrict__ gran,
                                                        u64* __restrict__ super) {
  const u64 gi = (u64)blockIdx.x * 256 + threadIdx.x;
  if (gi >= ngran) return;
  const u32 sh = SIGAX_SUPER_SHIFT - 7;  // granules per superblock = 2^sh
  const u64 first = (gi >> sh) << sh;
  const u64 base = wide ? goffs[first] : 0ull;
  gran[gi * 16 + col * 4] = (u32)(goffs[gi] - base);
  if (gi == first) super[(gi >> sh) * 4 + col] = base;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Phase {
  bool on;
  double t;
  Phase() : on(getenv("SIGAX_BUILD_TIMING") != nullptr), t(now_s()) {}
  void lap(const char* what) {
    if (!on) return;
    hipDeviceSynchronize();
    const double n = now_s();
    fprintf(stderr, "[sigax build] %-34s %8.3f s\n", what, n - t);
    t = n;
  }
};

template <typename IdxT>
static int sort_group(const u64* P, u64 n, u64 m, rocprim::double_buffer<u64>& K, rocprim::double_buffer<IdxT>& I, DevPool& pool,
                      u32* d_err, int n_cu, u64* large_rounds) {
  // round 0: all 63 key bits
  {
    size_t tb = 0;
    IB_TRY(rocprim::radix_sort_pairs(nullptr, tb, K, I, m, 0, 63));
    void* tmp = nullptr;
    IB_TRY(pool.alloc((char**)&tmp, tb));
    IB_TRY(rocprim::radix_sort_pairs(tmp, tb, K, I, m, 0, 63));
    IB_TRY(hipDeviceSynchronize());
    pool.release(tmp);
  }
  // equal-key segments of two and more suffixes
  u32 *dstart = nullptr, *dlen = nullptr, *dn = nullptr;
  const u64 maxd = m / 2 + 1;
  IB_TRY(pool.alloc(&dstart, maxd));
  IB_TRY(pool.alloc(&dlen, maxd));
  IB_TRY(pool.alloc(&dn, 4));
  auto segments = [&](const u64* keys, u64 cnt, u32* nd_out) -> int {
    size_t tb = 0;
    IB_TRY(rocprim::run_length_encode_non_trivial_runs(nullptr, tb, keys, (unsigned)cnt, dstart, dlen, dn));
    void* tmp = nullptr;
    IB_TRY(pool.alloc((char**)&tmp, tb));
    IB_TRY(rocprim::run_length_encode_non_trivial_runs(tmp, tb, keys, (unsigned)cnt, dstart, dlen, dn));
    IB_TRY(hipMemcpy(nd_out, dn, 4, hipMemcpyDeviceToHost));
    pool.release(tmp);
    return SIGAX_OK;
  };
  u32 nd = 0;
  int rc = segments(K.current(), m, &nd);
  if (rc != SIGAX_OK) return rc;
  const u32 max_rounds = 1u << 16;
  const unsigned fin_grid = (unsigned)std::min<u64>((u64)n_cu * 8, (nd + 255) / 256 + 1);
  if (nd) hipLaunchKernelGGL(k_seg_finish<IdxT>, dim3(fin_grid), dim3(256), 0, 0, P, n, I.current(), dstart, dlen, (const u32*)nullptr, nd, 21u, max_rounds, d_err);
  // larger segments: compact them, then rounds of (segment, next 10 symbols)
  u64* doff = nullptr;
  IB_TRY(pool.alloc(&doff, maxd + 1));
  auto large_total = [&](u32 ndesc, u64* total) -> int {
    if (ndesc == 0) {
      *total = 0;
      return SIGAX_OK;
    }
    auto in = rocprim::make_transform_iterator(dlen, LargeLen());
    size_t tb = 0;
    IB_TRY(rocprim::exclusive_scan(nullptr, tb, in, doff, 0ull, (size_t)ndesc + 1, rocprim::plus<u64>()));
    void* tmp = nullptr;
    IB_TRY(pool.alloc((char**)&tmp, tb));
    // one entry past the end (dlen[ndesc] is allocated scratch; its value only lands beyond doff[ndesc]) gives the total
    IB_TRY(rocprim::exclusive_scan(tmp, tb, in, doff, 0ull, (size_t)ndesc + 1, rocprim::plus<u64>()));
    IB_TRY(hipMemcpy(total, doff + ndesc, 8, hipMemcpyDeviceToHost));
    pool.release(tmp);
    return SIGAX_OK;
  };
  u64 mc = 0;
  if ((rc = large_total(nd, &mc)) != SIGAX_OK) return rc;
  if (mc == 0) {
    pool.release(dstart); pool.release(dlen); pool.release(dn); pool.release(doff);
    return SIGAX_OK;
  }
  // three position buffers: this round's input, the sort's second buffer, and the next round's compacted input
  IdxT* Ibuf[3] = {nullptr, nullptr, nullptr};
  u64* Kc[2] = {nullptr, nullptr};
  u32 *pos[2] = {nullptr, nullptr}, *seg[2] = {nullptr, nullptr};
  for (int k = 0; k < 3; ++k) IB_TRY(pool.alloc(&Ibuf[k], mc));
  for (int k = 0; k < 2; ++k) {
    IB_TRY(pool.alloc(&Kc[k], mc));
    IB_TRY(pool.alloc(&pos[k], mc));
    IB_TRY(pool.alloc(&seg[k], mc));
  }
  const unsigned cgrid = (unsigned)std::min<u64>((u64)n_cu * 8, (nd + 255) / 256 + 1);
  hipLaunchKernelGGL(k_large_compact<IdxT>, dim3(cgrid), dim3(256), 0, 0, (const IdxT*)I.current(), (const u32*)nullptr, dstart, dlen, (const u64*)doff, nd, Ibuf[0], pos[0], seg[0]);
  int cur = 0;  // pos[cur] / seg[cur] describe Ibuf[0]
  u32 depth = 21;
  void* stmp = nullptr;
  size_t stmp_bytes = 0;
  for (u64 round = 0; mc > 0; ++round) {
    if (round > (1u << 20)) return sigax_fail(SIGAX_E_CAPACITY, "suffix sort: a repeat deeper than %u symbols (identical reads in a row?)", depth);
    ++*large_rounds;
    hipLaunchKernelGGL(k_large_keys<IdxT>, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, 0, P, n, (const IdxT*)Ibuf[0], (const u32*)seg[cur], mc, depth, Kc[0]);
    rocprim::double_buffer<u64> kb(Kc[0], Kc[1]);
    rocprim::double_buffer<IdxT> ib(Ibuf[0], Ibuf[1]);
    unsigned hibit = 33;
    while (hibit < 64 && (mc >> (hibit - 32)) != 0) ++hibit;  // the segment ids are below mc
    size_t tb = 0;
    IB_TRY(rocprim::radix_sort_pairs(nullptr, tb, kb, ib, mc, 2, hibit));
    if (tb > stmp_bytes) {
      if (stmp) pool.release(stmp);
      IB_TRY(pool.alloc((char**)&stmp, tb));
      stmp_bytes = tb;
    }
    IB_TRY(rocprim::radix_sort_pairs(stmp, tb, kb, ib, mc, 2, hibit));
    IdxT* Is = ib.current();
    const u64* Ks = kb.current();
    // pos / seg are positional (a segment keeps its slots), only the suffixes inside a segment moved
    hipLaunchKernelGGL(k_large_writeback<IdxT>, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, 0, (const IdxT*)Is, (const u32*)pos[cur], mc, I.current());
    if ((rc = segments(Ks, mc, &nd)) != SIGAX_OK) return rc;
    depth += 10;
    if (nd) {
      const unsigned g = (unsigned)std::min<u64>((u64)n_cu * 8, (nd + 255) / 256 + 1);
      hipLaunchKernelGGL(k_seg_finish<IdxT>, dim3(g), dim3(256), 0, 0, P, n, I.current(), dstart, dlen, (const u32*)pos[cur], nd, depth, max_rounds, d_err);
    }
    u64 mc2 = 0;
    if ((rc = large_total(nd, &mc2)) != SIGAX_OK) return rc;
    if (mc2) {
      const unsigned g = (unsigned)std::min<u64>((u64)n_cu * 8, (nd + 255) / 256 + 1);
      hipLaunchKernelGGL(k_large_compact<IdxT>, dim3(g), dim3(256), 0, 0, (const IdxT*)Is, (const u32*)pos[cur], dstart, dlen, (const u64*)doff, nd, Ibuf[2], pos[cur ^ 1], seg[cur ^ 1]);
      IB_TRY(hipDeviceSynchronize());
      std::swap(Ibuf[0], Ibuf[2]);  // next round's input; the other two are free for its sort
      cur ^= 1;
    }
    mc = mc2;
  }
  IB_TRY(hipDeviceSynchronize());
  for (int k = 0; k < 3; ++k) pool.release(Ibuf[k]);
  for (int k = 0; k < 2; ++k) {
    pool.release(Kc[k]); pool.release(pos[k]); pool.release(seg[k]);
  }
  if (stmp) pool.release(stmp);
  pool.release(dstart); pool.release(dlen); pool.release(dn); pool.release(doff);
  return SIGAX_OK;
}

template <typename IdxT>
static int build_strand(const char* seqs, const uint64_t* offs, uint64_t n_reads, int reverse, int device, uint8_t** runs_out,
                        uint64_t* n_runs_out, uint32_t** sai_out, uint64_t* n_symbols_out) {
  Phase ph;
  DevPool pool;
  const u64 nbases = offs[n_reads];
  const u64 n = nbases + n_reads;  // symbols incl. one '$' per read; the end-of-text symbol sits at index n
  int n_cu = 256;
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
  unsigned char* d_seqs = nullptr;
  u64* d_offs = nullptr;
  IB_TRY(pool.alloc(&d_seqs, nbases + 16));
  IB_TRY(pool.alloc(&d_offs, n_reads + 1));
  IB_TRY(hipMemcpy(d_seqs, seqs, nbases, hipMemcpyHostToDevice));
  IB_TRY(hipMemcpy(d_offs, offs, (n_reads + 1) * 8, hipMemcpyHostToDevice));
  const u64 nw = (3 * (n + 1) + 63) / 64 + 4;  // + zero words: keys near the end read past the last symbol
  u64* P = nullptr;
  u32* startbits = nullptr;
  IB_TRY(pool.alloc(&P, nw));
  IB_TRY(pool.alloc(&startbits, (n >> 5) + 2));
  IB_TRY(hipMemset(startbits, 0, ((n >> 5) + 2) * 4));
  hipLaunchKernelGGL(k_pack_text, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, (const unsigned char*)d_seqs, (const u64*)d_offs, n_reads, n, reverse, P, nw);
  hipLaunchKernelGGL(k_mark_starts, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, 0, (const u64*)d_offs, n_reads, startbits);
  IB_TRY(hipDeviceSynchronize());
  pool.release(d_seqs);
  ph.lap("upload + pack text");

  u64* d_hist = nullptr;
  IB_TRY(pool.alloc(&d_hist, IB_BINS));
  IB_TRY(hipMemset(d_hist, 0, IB_BINS * 8));
  hipLaunchKernelGGL(k_hist5, dim3((unsigned)n_cu), dim3(1024), 0, 0, (const u64*)P, n, d_hist);
  std::vector<u64> hist(IB_BINS);
  IB_TRY(hipMemcpy(hist.data(), d_hist, IB_BINS * 8, hipMemcpyDeviceToHost));
  ph.lap("prefix histogram");

  unsigned char* B = nullptr;
  u32* d_sai = nullptr;
  u64* d_count = nullptr;
  u32* d_err = nullptr;
  IB_TRY(pool.alloc(&B, n + 64));
  IB_TRY(pool.alloc(&d_sai, n_reads + 1));
  IB_TRY(pool.alloc(&d_count, 2));
  IB_TRY(pool.alloc(&d_err, 1));
  IB_TRY(hipMemset(d_err, 0, 4));

  // groups of consecutive prefixes that fit the sort workspace (keys + positions, double-buffered, + segment lists)
  size_t free_b = 0, total_b = 0;
  IB_TRY(hipMemGetInfo(&free_b, &total_b));
  u64 cap = (1ull << 31) - (1ull << 24);
  const char* envcap = getenv("SIGAX_BUILD_GROUP");  // tests force several groups on small inputs
  if (envcap) cap = std::max<u64>(strtoull(envcap, nullptr, 10), 1);
  cap = std::min<u64>(cap, (u64)(free_b * 0.7) / (16 + 2 * sizeof(IdxT) + 20));
  u64 biggest = 0;
  for (u64 h : hist) biggest = std::max(biggest, h);
  if (biggest > cap && !envcap)
    return sigax_fail(SIGAX_E_CAPACITY, "suffix sort: %llu suffixes share a 5-symbol prefix, the workspace holds %llu", biggest, cap);
  cap = std::max(cap, biggest);

  u64 done = 0, sai_done = 0, large_rounds = 0, ngroups = 0;
  u32 b = 0;
  while (b < IB_BINS) {
    u32 e = b;
    u64 m = 0;
    while (e < IB_BINS && m + hist[e] <= cap) m += hist[e++];
    if (e == b) return sigax_fail(SIGAX_E_CAPACITY, "suffix sort: prefix bin %u does not fit the workspace", b);
    if (b == 0) {  // the end-of-text suffix (position n) is SA[0]: not a BWT row of the .bwt (src/bwt.cpp:7-32 over n symbols)
      // key63(P, n) has code 0 first -> bin 0; it is never collected because k_collect stops at n
    }
    if (m) {
      ++ngroups;
      u64 *K0 = nullptr, *K1 = nullptr;
      IdxT *I0 = nullptr, *I1 = nullptr;
      IB_TRY(pool.alloc(&K0, m));
      IB_TRY(pool.alloc(&K1, m));
      IB_TRY(pool.alloc(&I0, m));
      IB_TRY(pool.alloc(&I1, m));
      IB_TRY(hipMemset(d_count, 0, 8));
      hipLaunchKernelGGL(k_collect<IdxT>, dim3((unsigned)n_cu * 2), dim3(1024), 0, 0, (const u64*)P, n, b, e, K0, I0, d_count);
      u64 got = 0;
      IB_TRY(hipMemcpy(&got, d_count, 8, hipMemcpyDeviceToHost));
      if (got != m) return sigax_fail(SIGAX_E_DEVICE, "suffix sort: collected %llu suffixes, histogram said %llu", got, m);
      rocprim::double_buffer<u64> K(K0, K1);
      rocprim::double_buffer<IdxT> I(I0, I1);
      int rc = sort_group<IdxT>(P, n, m, K, I, pool, d_err, n_cu, &large_rounds);
      if (rc != SIGAX_OK) return rc;
      // BWT symbols + read-start rows; the spare key buffer holds the row ids
      u32* R = reinterpret_cast<u32*>(K.alternate());
      hipLaunchKernelGGL(k_emit<IdxT>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, (const u64*)P, (const u32*)startbits, (const u64*)d_offs, n_reads,
                         (const IdxT*)I.current(), m, B + done, R);
      size_t tb = 0;
      IB_TRY(rocprim::select(nullptr, tb, R, d_sai + sai_done, d_count, (size_t)m, IsRow()));
      void* tmp = nullptr;
      IB_TRY(pool.alloc((char**)&tmp, tb));
      IB_TRY(rocprim::select(tmp, tb, R, d_sai + sai_done, d_count, (size_t)m, IsRow()));
      u64 rows = 0;
      IB_TRY(hipMemcpy(&rows, d_count, 8, hipMemcpyDeviceToHost));
      sai_done += rows;
      done += m;
      pool.release(tmp);
      pool.release(K0); pool.release(K1); pool.release(I0); pool.release(I1);
    }
    b = e;
  }
  u32 err = 0;
  IB_TRY(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
  if (err) return sigax_fail(SIGAX_E_CAPACITY, "suffix sort: a segment needed more than 65536 rounds (identical reads in a row?)");
  if (done != n || sai_done != n_reads)
    return sigax_fail(SIGAX_E_DEVICE, "suffix sort: %llu of %llu rows, %llu of %llu read starts", done, n, sai_done, (u64)n_reads);
  if (ph.on) fprintf(stderr, "[sigax build] %llu symbols, %llu groups, %llu large rounds\n", n, ngroups, large_rounds);
  ph.lap("sort groups + emit");

  // RL units
  const u64 nchunks = (n + RL_CHUNK - 1) / RL_CHUNK;
  u32* cnt = nullptr;
  u64 *uoff = nullptr, *partial = nullptr, *total = nullptr, *headscan = nullptr, *lasthead = nullptr;
  IB_TRY(pool.alloc(&headscan, nchunks + 1));
  IB_TRY(pool.alloc(&lasthead, nchunks + 1));
  IB_TRY(pool.alloc(&cnt, nchunks + 1));
  IB_TRY(pool.alloc(&uoff, nchunks + 2));
  IB_TRY(pool.alloc(&partial, scan_partials_needed(nchunks)));
  IB_TRY(pool.alloc(&total, 1));
  u64 nruns = 0;
  if (n) {
    hipLaunchKernelGGL(k_rl_heads, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, 0, (const unsigned char*)B, n, lasthead);
    {
      size_t tb = 0;
      IB_TRY(rocprim::inclusive_scan(nullptr, tb, lasthead, headscan, (size_t)nchunks, MaxU64()));
      void* tmp = nullptr;
      IB_TRY(pool.alloc((char**)&tmp, tb));
      IB_TRY(rocprim::inclusive_scan(tmp, tb, lasthead, headscan, (size_t)nchunks, MaxU64()));
      IB_TRY(hipDeviceSynchronize());
      pool.release(tmp);
      pool.release(lasthead);
    }
    hipLaunchKernelGGL(k_rl_units<false>, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, 0, (const unsigned char*)B, n, (const u64*)headscan, cnt, (const u64*)nullptr, (unsigned char*)nullptr);
    launch_scan(cnt, nchunks, partial, uoff, total, 0);
    IB_TRY(hipMemcpy(&nruns, total, 8, hipMemcpyDeviceToHost));
  }
  unsigned char* d_runs = nullptr;
  IB_TRY(pool.alloc(&d_runs, nruns + 16));
  if (n) hipLaunchKernelGGL(k_rl_units<true>, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, 0, (const unsigned char*)B, n, (const u64*)headscan, (u32*)nullptr, (const u64*)uoff, d_runs);
  IB_TRY(hipDeviceSynchronize());
  IB_TRY(hipGetLastError());
  uint8_t* runs = (uint8_t*)malloc(std::max<u64>(nruns, 1));
  uint32_t* sai = (uint32_t*)malloc(std::max<u64>(n_reads, 1) * 4);
  if (!runs || !sai) {
    free(runs);
    free(sai);
    return sigax_fail(SIGAX_E_ARG, "host allocation failed");
  }
  hipError_t e1 = nruns ? hipMemcpy(runs, d_runs, nruns, hipMemcpyDeviceToHost) : hipSuccess;
  hipError_t e2 = n_reads ? hipMemcpy(sai, d_sai, n_reads * 4, hipMemcpyDeviceToHost) : hipSuccess;
  if (e1 != hipSuccess || e2 != hipSuccess) {
    free(runs);
    free(sai);
    return sigax_fail(SIGAX_E_DEVICE, "copying the index back: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  }
  ph.lap("RL units + download");
  *runs_out = runs;
  *n_runs_out = nruns;
  *sai_out = sai;
  *n_symbols_out = n;
  return SIGAX_OK;
}

}  // namespace

// RL units on the host -> granules + superblock table in device memory.  C[] = FMIndex::_pred (src/fmindex.cpp:156-160).
int sigax_decode_strand(const uint8_t* runs, u64 n_runs, u64 nsym, bool wide, void** d_gran, u64* gran_bytes, void** d_super,
                        u64* super_bytes, u64 C[5], u64 total[5]) {
  DevPool pool;
  *d_gran = *d_super = nullptr;
  const u64 ngran = nsym / SIGAX_GRANULE_SYMS + 1, nchunks = ngran * 4;
  const u64 nsuper = ((ngran - 1) >> (SIGAX_SUPER_SHIFT - 7)) + 1;
  unsigned char* d_runs = nullptr;
  u32 *cnt = nullptr, *bad = nullptr, *gcnt = nullptr;
  u64 *roffs = nullptr, *partial = nullptr, *tot = nullptr, *goffs = nullptr;
  u32* gran = nullptr;
  u64* super = nullptr;
  IB_TRY(hipMalloc((void**)&gran, ngran * 64));
  hipError_t es = hipMalloc((void**)&super, nsuper * 32);
  if (es != hipSuccess) {
    hipFree(gran);
    return sigax_fail(SIGAX_E_DEVICE, "hipMalloc(superblock table): %s", hipGetErrorString(es));
  }
  struct Guard {  // the two outputs are handed over only on success
    void *a, *b;
    bool keep;
    ~Guard() {
      if (!keep) {
        hipFree(a);
        hipFree(b);
      }
    }
  } guard{gran, super, false};
  IB_TRY(pool.alloc(&d_runs, n_runs + 16));
  IB_TRY(pool.alloc(&cnt, n_runs + 1));
  IB_TRY(pool.alloc(&roffs, n_runs + 2));
  IB_TRY(pool.alloc(&partial, scan_partials_needed(std::max(n_runs, ngran))));
  IB_TRY(pool.alloc(&tot, 1));
  IB_TRY(pool.alloc(&bad, 1));
  IB_TRY(hipMemset(bad, 0, 4));
  u64 covered = 0;
  if (n_runs) {
    IB_TRY(hipMemcpy(d_runs, runs, n_runs, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_run_lengths, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, 0, (const unsigned char*)d_runs, n_runs, cnt, bad);
    launch_scan(cnt, n_runs, partial, roffs, tot, 0);
    IB_TRY(hipMemcpy(&covered, tot, 8, hipMemcpyDeviceToHost));
    u32 b = 0;
    IB_TRY(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
    if (b) return sigax_fail(SIGAX_E_IO, "invalid RL unit (symbol rank above 4) in the .bwt payload");
  }
  if (covered != nsym) return sigax_fail(SIGAX_E_IO, "run lengths (%llu) do not add up to the symbol count (%llu)", covered, nsym);
  pool.release(cnt);
  IB_TRY(pool.alloc(&gcnt, 4 * ngran));
  IB_TRY(pool.alloc(&goffs, ngran + 2));
  hipLaunchKernelGGL(k_decode_chunks, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, 0, (const unsigned char*)d_runs, (const u64*)roffs, n_runs,
                     nsym, nchunks, gran, gcnt, ngran);
  u64 sum = 0;
  for (u32 col = 0; col < 4; ++col) {
    launch_scan(gcnt + (u64)col * ngran, ngran, partial, goffs, tot, 0);
    hipLaunchKernelGGL(k_granule_headers, dim3((unsigned)((ngran + 255) / 256)), dim3(256), 0, 0, (const u64*)goffs, ngran, col, wide ? 1 : 0, gran, super);
    IB_TRY(hipMemcpy(&total[1 + col], tot, 8, hipMemcpyDeviceToHost));
    sum += total[1 + col];
  }
  IB_TRY(hipDeviceSynchronize());
  IB_TRY(hipGetLastError());
  total[0] = nsym - sum;
  C[0] = 0;
  for (int k = 1; k < 5; ++k) C[k] = C[k - 1] + total[k - 1];
  guard.keep = true;
  *d_gran = gran;
  *d_super = super;
  *gran_bytes = ngran * 64;
  *super_bytes = nsuper * 32;
  return SIGAX_OK;
}


// -------------------------------------------------------------------------------------------------------
// Locality order of a batch of reads for the block finder (sigax_api.cpp: enqueue): reads that overlap walk nested BWT
// intervals a few steps apart, so when they sit in neighbouring lanes their rank lines -- and, on big indexes, the page
// translations -- are still cached (reads handed over in genome order: +14 % at BASELINE configs[1], +23 % at the
// configs[2] shape).  Real input comes in any order, so the reads of a batch are grouped by their minimizer (the canonical
// 16-mer with the smallest hash): reads of one class share a genome 16-mer.  A counting sort in three launches -- keys +
// histogram, scan, scatter -- because what the ordering costs a batch is the LATENCY of its launches on a GPU the other
// batches' kernels keep full (every launch waits a millisecond or two for workgroup slots): the radix sort of round 2,
// some twenty dependent launches, took 17 to 40 ms of a batch's chain at the configs[2] shape for 1 ms of work.  The order
// inside a class is whatever the scatter's atomics make it (the results do not depend on the order, only the finder's
// cache behaviour does).  `bounds` = the sub-batches' slot ranges: the order is a permutation inside each.
// -------------------------------------------------------------------------------------------------------
namespace {
struct OrderBounds { u32 n, b[9]; };
// Class = the top bits of the minimizer's hash: 2^20 classes (a batch of a few million reads at a few-fold coverage holds
// about a million minimizers; reads put side by side only help each other when they really share one).  SIGAX_ORDER_BITS.
static u32 order_class_bits() {
  static const u32 bits = [] {
    const char* env = getenv("SIGAX_ORDER_BITS");
    const int b = env ? atoi(env) : 20;
    return (u32)(b < 8 ? 8 : b > 20 ? 20 : b);  // 3 bits of sub-batch + the class share the low 23 bits of a key
  }();
  return bits;
}
// One thread per read, one wave per workgroup; its `per` reads (one byte range of the batch; 64, or fewer when the reads
// are long, so that the range fits the LDS tile) are first copied to LDS with coalesced word loads -- round 2's kernel had
// every thread walk its read byte by byte in global memory, 150 scattered loads per read, and took 2.4 ms per 1 M reads (250
// bp reads read in place: 13.6 ms).  Tiles whose bytes still do not fit are read in place.  The tile is 16 KB: this kernel
// runs beside the finders of the batches in flight, which leave a CU 4 to 40 KB of LDS, and a workgroup that asks for
// 48 KB waits for finder workgroups to retire (measured: 25 to 40 ms for 1 ms of work at the BASELINE configs[2] shape).
#define KEYS_LDS_BYTES 16384u
#define KEYS_NT 64
template <class GetByte>
__device__ __forceinline__ u32 read_class(u32 L, u32 bits, u32* ord_out, GetByte get) {
  u32 fwd = 0, rev = 0, have = 0, best = 0xFFFFFFFFu, bs = 0, bo = 0;
  for (u32 i = 0; i < L; ++i) {
    const u32 ch = get(i);
    const u32 c = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    if (c > 3u) { have = 0; continue; }
    fwd = (fwd << 2) | c;
    rev = (rev >> 2) | ((3u - c) << 30);
    if (++have < 16u) continue;
    const u32 st = fwd <= rev ? 0u : 1u;
    u32 h = (st ? rev : fwd) * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    if (h < best) { best = h; bs = st; bo = i - 15u; }
  }
  // where the minimizer sits in the read, in 4-base steps, turned so that it grows with the read's start in the genome
  const u32 o9 = (bo >> 2) > 0x1FFu ? 0x1FFu : (bo >> 2);
  *ord_out = bs ? o9 : 0x1FFu - o9;
  // the strand the minimizer was seen on is part of the class: reads of one class then lie the same way round on the
  // genome, so that it is the same chain (= the same wave of the finder) of each that walks the same rows
  return ((best >> (33u - bits)) << 1) | bs;
}
__global__ __launch_bounds__(KEYS_NT) void k_read_keys(const unsigned char* seqs, const u64* offs, u32 n, u32 per, u32 bits, OrderBounds ob, u32* keys,
                                                       u32* hist) {
  __shared__ __attribute__((aligned(16))) u32 tile[KEYS_LDS_BYTES / 4];
  const u32 r0 = blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
  const u64 lo = offs[r0], hi = offs[r1];
  const u64 alo = (reinterpret_cast<u64>(seqs) + lo) & ~3ull;  // whole aligned words: the first one may start before the first base
  const u64 nbytes = reinterpret_cast<u64>(seqs) + hi - alo;
  const bool staged = nbytes + 4 <= (u64)KEYS_LDS_BYTES;
  if (staged) {
    const u32* src = reinterpret_cast<const u32*>(alo);
    const u32 nw = (u32)((nbytes + 3) >> 2);
    for (u32 w = threadIdx.x; w < nw; w += KEYS_NT) tile[w] = src[w];
  }
  __syncthreads();
  const u32 r = r0 + threadIdx.x;
  if (threadIdx.x >= per || r >= n) return;
  u32 sub = 0;
  for (u32 i = 1; i < ob.n; ++i) sub += r >= ob.b[i] ? 1u : 0u;
  const u64 b0 = offs[r];
  const u32 L = (u32)(offs[r + 1] - b0);
  u32 cls, ord;
  if (staged) {
    const u32 d = (u32)(reinterpret_cast<u64>(seqs) + b0 - alo);  // this read's first base in the tile
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(tile) + d;
    cls = read_class(L, bits, &ord, [&](u32 i) { return (u32)tb[i]; });
  } else {
    cls = read_class(L, bits, &ord, [&](u32 i) { return (u32)seqs[b0 + i]; });
  }
  const u32 bucket = (sub << bits) | cls;
  keys[r] = bucket;
  (void)ord;  // the order by start inside a class (an insertion sort per class, a fourth launch) was tried: it buys the
              // finder nothing measurable at the configs[2] shape (21.65 ms alone either way, 23.3 ms unordered)
  atomicAdd(&hist[bucket], 1u);
}
// exclusive scan of the class counts in place, one workgroup
__global__ __launch_bounds__(1024) void k_order_scan(u32* hist, u32 nb) {
  __shared__ u32 part[1024];
  const u32 per = (nb + 1023u) / 1024u;
  const u32 lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
  u32 sum = 0;
  for (u32 i = lo; i < hi; ++i) sum += hist[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (u32 off = 1; off < 1024u; off <<= 1) {
    const u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - sum;
  for (u32 i = lo; i < hi; ++i) {
    const u32 c = hist[i];
    hist[i] = run;
    run += c;
  }
}
__global__ __launch_bounds__(256) void k_order_scatter_reads(const u32* keys, u32 n, u32* cursor, u32* perm) {
  const u32 r = blockIdx.x * 256 + threadIdx.x;
  if (r < n) perm[atomicAdd(&cursor[keys[r]], 1u)] = r;
}
}  // namespace

size_t sigax_order_reads_tmp_bytes(uint32_t nsub) { return ((size_t)(nsub ? nsub : 1) << order_class_bits()) * 4; }  // the class counters

int sigax_order_reads(const unsigned char* d_seqs, const u64* d_offs, u32 n, u32 max_len, const u32* bounds, u32 nsub, u32* keys, u32* vals,
                      void* tmp, size_t tmp_bytes, const u32** result, hipStream_t st) {
  *result = vals;
  if (n == 0) return SIGAX_OK;
  OrderBounds ob;
  ob.n = nsub > 8 ? 8 : nsub;
  for (u32 i = 0; i < 9; ++i) ob.b[i] = i <= ob.n ? bounds[i] : n;
  const u32 bits = order_class_bits();
  const u32 nb = ob.n << bits;
  if ((size_t)nb * 4 > tmp_bytes) return sigax_fail(SIGAX_E_ARG, "ordering scratch too small");
  u32* hist = (u32*)tmp;
  hipError_t e = hipMemsetAsync(hist, 0, (size_t)nb * 4, st);
  if (e == hipSuccess) {
    u32 per = KEYS_NT;  // reads per workgroup: as many as fit the LDS tile, a power of two
    while (per > 8 && (u64)per * max_len + 8 > KEYS_LDS_BYTES) per >>= 1;
    hipLaunchKernelGGL(k_read_keys, dim3((n + per - 1) / per), dim3(KEYS_NT), 0, st, d_seqs, d_offs, n, per, bits, ob, keys, hist);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(1024), 0, st, hist, nb);
    hipLaunchKernelGGL(k_order_scatter_reads, dim3((n + 255) / 256), dim3(256), 0, st, (const u32*)keys, n, hist, vals);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return sigax_fail(SIGAX_E_DEVICE, "ordering the batch's reads: %s", hipGetErrorString(e));
  }
  return SIGAX_OK;
}

extern "C" int sigax_build_strand(const char* seqs, const uint64_t* offs, uint64_t n_reads, int reverse, int device,
                                  uint8_t** runs, uint64_t* n_runs, uint32_t** sai, uint64_t* n_symbols) {
  if (!offs || !runs || !n_runs || !sai || !n_symbols || (n_reads && offs[n_reads] && !seqs)) return sigax_fail(SIGAX_E_ARG, "NULL argument");
  *runs = nullptr;
  *sai = nullptr;
  *n_runs = *n_symbols = 0;
  if (n_reads >= 0xFFFFFFFFull) return sigax_fail(SIGAX_E_ARG, "too many reads for the .sai format");  // src/suffix_array.h:33-34
  for (uint64_t i = 0; i < n_reads; ++i)
    if (offs[i + 1] < offs[i]) return sigax_fail(SIGAX_E_ARG, "bad read offsets");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sigax_fail(SIGAX_E_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return sigax_fail(SIGAX_E_ARG, "device %d out of range (%d visible)", device, ndev);
  IB_TRY(hipSetDevice(device));
  const u64 n = offs[n_reads] + n_reads;
  if (n_reads == 0) {  // an empty read set: empty BWT, no rows
    *runs = (uint8_t*)malloc(1);
    *sai = (uint32_t*)malloc(4);
    return SIGAX_OK;
  }
  if (n + 64 < 0xFFFFFFFFull) return build_strand<uint32_t>(seqs, offs, n_reads, reverse, device, runs, n_runs, sai, n_symbols);
  return build_strand<u64>(seqs, offs, n_reads, reverse, device, runs, n_runs, sai, n_symbols);
}

extern "C" void sigax_free(void* p) { free(p); }
