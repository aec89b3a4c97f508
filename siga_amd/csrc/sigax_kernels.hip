// siga_amd/csrc/sigax_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the `siga overlap` hot path.
//
//   k_find            OverlapBlockFinder::find for 4 orientations per read   (src/overlap_builder.cpp:838-912)
//   k_filter_extract  SubMaximalBlockFilter + ContainmentBlockRemover + IrreducibleBlockListExtractor +
//                     the list plumbing of OverlapBuilder::overlap           (src/overlap_builder.cpp:706-836,
//                                                                             914-1182)
//   k_order_*         ordered compaction of the per-read block lists (replaces the hits text round trip,
//                     src/overlap_builder.cpp:234-254,269-280)
//   k_edge_*          Hit2OverlapConverter::convert                          (src/overlap_builder.cpp:345-375)
//   k_occ_batch       FMIndex::getOcc                                        (src/fmindex.cpp:188-231,320-323)
//   k_kmer_count      FMIndex::Interval::occurrences                         (src/fmindex.h:67-86)
//
// Integer / index work only: HBM- and latency-bound gathers of 64-byte rank granules (fm_layout.h); no MFMA.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sigax_kernels.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// Non-temporal hints on data that passes through once -- the finder's candidate records on their way out (bit 2), the
// records on their way into filter/extract (4), the row-table line a block looks up (1): they leave more of the caches to
// the finder's tables.  Measured at BASELINE configs[1] (gpurun_out/r3w/): 118.6 M reads/s without, 119.4-120.5 M with one
// of them, 121.0 M with all three; nothing at the configs[2] shape.  -DSIGAX_NT=0 turns them off.
#ifndef SIGAX_NT
#define SIGAX_NT 7
#endif

// -------------------------------------------------------------------------------------------------------
// rank: number of A,C,G,T in BWT[0, p)   == FMIndex::getOcc(p - 1) without the '$' column
// -------------------------------------------------------------------------------------------------------
struct Cnt4 {
  u64 a, c, g, t;
};

// Position type of an index: u32 while the BWT has fewer than 2^32 symbols (half the registers and ALU work), else u64.
template <bool WIDE> struct PosOf { typedef u32 type; };
template <> struct PosOf<true> { typedef u64 type; };
template <typename P> struct Cnt4P {
  P a, c, g, t;
};

// A candidate block as the finder leaves it in the arena: one naturally aligned record (32 B with u32 positions, 64 B
// with u64) = one or two 16-byte stores per half instead of the five scattered ones of an 80-byte sigax_block.
//   capped[0] = [c0lo, c0lo + d - 1], capped[1] = [c1lo, c1lo + d - 1]   (both intervals of a pair have one size)
//   raw[0]    = [r0lo, r0lo + sz - 1], raw[1]   = [r1lo, r1lo + sz - 1]
template <bool WIDE> struct Cand;
template <> struct __attribute__((aligned(32))) Cand<false> {
  u32 c0lo, d, c1lo, r0lo, r1lo, sz, len, af;
};
template <> struct __attribute__((aligned(64))) Cand<true> {
  u64 c0lo, d, c1lo, r0lo, r1lo, sz;
  u32 len, af;
  u64 pad;
};
static_assert(sizeof(Cand<false>) == 32 && sizeof(Cand<true>) == 64, "candidate record sizes");

__device__ __forceinline__ void cand_store(Cand<false>* dst, u32 c0lo, u32 d, u32 c1lo, u32 r0lo, u32 r1lo, u32 sz, u32 len, u32 af) {
  uint4* q = reinterpret_cast<uint4*>(dst);
  q[0] = make_uint4(c0lo, d, c1lo, r0lo);
  q[1] = make_uint4(r1lo, sz, len, af);
}
__device__ __forceinline__ void cand_store(Cand<true>* dst, u64 c0lo, u64 d, u64 c1lo, u64 r0lo, u64 r1lo, u64 sz, u32 len, u32 af) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(dst);
  q[0] = make_ulonglong2(c0lo, d);
  q[1] = make_ulonglong2(c1lo, r0lo);
  q[2] = make_ulonglong2(r1lo, sz);
  q[3] = make_ulonglong2((u64)len | ((u64)af << 32), 0ull);
}
template <bool WIDE>
__device__ __forceinline__ Cand<WIDE> cand_load(const Cand<WIDE>* src);
template <>
__device__ __forceinline__ Cand<false> cand_load<false>(const Cand<false>* src) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
#if (SIGAX_NT & 4)
  typedef u32 nt4 __attribute__((ext_vector_type(4)));
  const nt4 na = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(q)), nb = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(q) + 1);
  uint4 a = make_uint4(na.x, na.y, na.z, na.w), b = make_uint4(nb.x, nb.y, nb.z, nb.w);
#else
  uint4 a = q[0], b = q[1];
#endif
  Cand<false> c;
  c.c0lo = a.x; c.d = a.y; c.c1lo = a.z; c.r0lo = a.w; c.r1lo = b.x; c.sz = b.y; c.len = b.z; c.af = b.w;
  return c;
}
template <>
__device__ __forceinline__ Cand<true> cand_load<true>(const Cand<true>* src) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(src);
  ulonglong2 a = q[0], b = q[1], c2 = q[2], d = q[3];
  Cand<true> c;
  c.c0lo = a.x; c.d = a.y; c.c1lo = b.x; c.r0lo = b.y; c.r1lo = c2.x; c.sz = c2.y; c.len = (u32)d.x; c.af = (u32)(d.x >> 32);
  c.pad = 0;
  return c;
}

__device__ __forceinline__ void chunk_count(const uint4& k, int take, u32& a, u32& c, u32& g, u32& t) {
  u32 m = take >= 32 ? 0xFFFFFFFFu : (take <= 0 ? 0u : ((1u << take) - 1u));
  u32 x0 = k.y & m, x1 = k.z & m, x2 = k.w & m;
  a += __popc(x0 & ~x1);
  c += __popc(x1 & ~x0);
  g += __popc(x0 & x1);
  t += __popc(x2);
}

// By-value view of one strand: granule table, superblock table, length, and which LDS row holds its C[]/totals.
struct FmRef {
  const uint4* g;
  const u64* super;
  u64 n;
  u32 which;
};
__device__ __forceinline__ FmRef fm_ref(const FmStrand& s, u32 which) {
  FmRef r;
  r.g = reinterpret_cast<const uint4*>(s.granules);
  r.super = s.super;
  r.n = s.n;
  r.which = which;
  return r;
}
__device__ __forceinline__ FmRef fm_pick(bool first, const FmRef& a, const FmRef& b) {
  FmRef r;
  r.g = first ? a.g : b.g;
  r.super = first ? a.super : b.super;
  r.n = first ? a.n : b.n;
  r.which = first ? a.which : b.which;
  return r;
}
// C[] and symbol totals of both strands staged in LDS: [which][rank]
struct FmTables {
  u64 C[2][5];
  u64 T[2][5];
};
__device__ __forceinline__ void fm_tables_load(FmTables& t, const FmStrand& fwd, const FmStrand& rev) {
  if (threadIdx.x < 5) {
    t.C[0][threadIdx.x] = fwd.C[threadIdx.x];
    t.C[1][threadIdx.x] = rev.C[threadIdx.x];
    t.T[0][threadIdx.x] = fwd.total[threadIdx.x];
    t.T[1][threadIdx.x] = rev.total[threadIdx.x];
  }
  __syncthreads();
}

template <bool WIDE>
__device__ __forceinline__ Cnt4 fm_rank(const FmRef& s, u64 p) {
  p = p > s.n ? s.n : p;  // never leave the table, whatever an invalid interval holds
  u64 gi = p >> 7;
  int r = (int)(p & 127u);
  const uint4* q = s.g + gi * 4;
  uint4 k0 = q[0], k1 = q[1], k2 = q[2], k3 = q[3];
  u32 a = k0.x, c = k1.x, g = k2.x, t = k3.x;
  chunk_count(k0, r, a, c, g, t);
  chunk_count(k1, r - 32, a, c, g, t);
  chunk_count(k2, r - 64, a, c, g, t);
  chunk_count(k3, r - 96, a, c, g, t);
  Cnt4 o;
  o.a = a; o.c = c; o.g = g; o.t = t;
  if (WIDE) {
    const u64* sb = s.super + (p >> SIGAX_SUPER_SHIFT) * 4;
    o.a += sb[0]; o.c += sb[1]; o.g += sb[2]; o.t += sb[3];
  }
  return o;
}

// all five columns; v[0] = '$'
template <bool WIDE>
__device__ __forceinline__ void fm_rank5(const FmRef& s, u64 p, u64 v[5]) {
  u64 pc = p > s.n ? s.n : p;
  Cnt4 k = fm_rank<WIDE>(s, pc);
  v[1] = k.a; v[2] = k.c; v[3] = k.g; v[4] = k.t;
  v[0] = pc - (k.a + k.c + k.g + k.t);
}

// A,C,G,T counts of BWT[0, p) in the index's position type
template <bool WIDE>
__device__ __forceinline__ Cnt4P<typename PosOf<WIDE>::type> fm_rank4p(const FmRef& s, typename PosOf<WIDE>::type p) {
  typedef typename PosOf<WIDE>::type P;
  u64 pc = (u64)p > s.n ? s.n : (u64)p;
  const uint4* q = s.g + (pc >> 7) * 4;
  int r = (int)(pc & 127u);
  uint4 k0 = q[0], k1 = q[1], k2 = q[2], k3 = q[3];
  u32 a = k0.x, c = k1.x, g = k2.x, t = k3.x;
  chunk_count(k0, r, a, c, g, t);
  chunk_count(k1, r - 32, a, c, g, t);
  chunk_count(k2, r - 64, a, c, g, t);
  chunk_count(k3, r - 96, a, c, g, t);
  Cnt4P<P> o;
  o.a = a; o.c = c; o.g = g; o.t = t;
  if (WIDE) {
    const u64* sb = s.super + (pc >> SIGAX_SUPER_SHIFT) * 4;
    o.a += (P)sb[0]; o.c += (P)sb[1]; o.g += (P)sb[2]; o.t += (P)sb[3];
  }
  return o;
}

// The finder's loads of one step, issued back to back and waited for once: both rank granules (4 x 16 B each; the
// pieces of a granule share a line) and the read's base for this step.  Left to the compiler, the conditional record
// stores of the loop make it wait with vmcnt(0) part-way through issuing these loads and issue the base's load after
// that wait, which puts one or two extra cache round trips on every step of the chain.
typedef u32 v4u __attribute__((ext_vector_type(4)));
struct Gran {
  v4u k0, k1, k2, k3;
};
__device__ __forceinline__ void find_step_loads(const void* qa, const void* qb, const unsigned char* pc, Gran& a, Gran& b, u32& ch) {
  asm volatile(
      "global_load_dwordx4 %0, %9, off\n\t"
      "global_load_dwordx4 %4, %10, off\n\t"
      "global_load_ubyte %8, %11, off\n\t"
      "global_load_dwordx4 %1, %9, off offset:16\n\t"
      "global_load_dwordx4 %2, %9, off offset:32\n\t"
      "global_load_dwordx4 %3, %9, off offset:48\n\t"
      "global_load_dwordx4 %5, %10, off offset:16\n\t"
      "global_load_dwordx4 %6, %10, off offset:32\n\t"
      "global_load_dwordx4 %7, %10, off offset:48\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.k0), "=&v"(a.k1), "=&v"(a.k2), "=&v"(a.k3), "=&v"(b.k0), "=&v"(b.k1), "=&v"(b.k2), "=&v"(b.k3), "=&v"(ch)
      : "v"(qa), "v"(qb), "v"(pc)
      : "memory");
}
// Same with wave-uniform bases in scalar registers and 32-bit byte offsets per lane (granule table < 4 GiB, i.e. the
// u32-position index; the reads of one wave lie within 4 GiB of the wave's first read).
__device__ __forceinline__ void find_step_loads_s(const void* gbase, u32 offa, u32 offb, const void* sbase, u32 offc, Gran& a, Gran& b,
                                                  u32& ch) {
  asm volatile(
      "global_load_dwordx4 %0, %9, %12\n\t"
      "global_load_dwordx4 %4, %10, %12\n\t"
      "global_load_ubyte %8, %11, %13\n\t"
      "global_load_dwordx4 %1, %9, %12 offset:16\n\t"
      "global_load_dwordx4 %2, %9, %12 offset:32\n\t"
      "global_load_dwordx4 %3, %9, %12 offset:48\n\t"
      "global_load_dwordx4 %5, %10, %12 offset:16\n\t"
      "global_load_dwordx4 %6, %10, %12 offset:32\n\t"
      "global_load_dwordx4 %7, %10, %12 offset:48\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.k0), "=&v"(a.k1), "=&v"(a.k2), "=&v"(a.k3), "=&v"(b.k0), "=&v"(b.k1), "=&v"(b.k2), "=&v"(b.k3), "=&v"(ch)
      : "v"(offa), "v"(offb), "v"(offc), "s"(gbase), "s"(sbase)
      : "memory");
}
// The two variants without the base load: the workgroup's reads are staged in LDS (the usual case)
__device__ __forceinline__ void find_step_loads8(const void* qa, const void* qb, Gran& a, Gran& b) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off\n\t"
      "global_load_dwordx4 %4, %9, off\n\t"
      "global_load_dwordx4 %1, %8, off offset:16\n\t"
      "global_load_dwordx4 %2, %8, off offset:32\n\t"
      "global_load_dwordx4 %3, %8, off offset:48\n\t"
      "global_load_dwordx4 %5, %9, off offset:16\n\t"
      "global_load_dwordx4 %6, %9, off offset:32\n\t"
      "global_load_dwordx4 %7, %9, off offset:48\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.k0), "=&v"(a.k1), "=&v"(a.k2), "=&v"(a.k3), "=&v"(b.k0), "=&v"(b.k1), "=&v"(b.k2), "=&v"(b.k3)
      : "v"(qa), "v"(qb)
      : "memory");
}
__device__ __forceinline__ void find_step_loads8_s(const void* gbase, u32 offa, u32 offb, Gran& a, Gran& b) {
  asm volatile(
      "global_load_dwordx4 %0, %8, %10\n\t"
      "global_load_dwordx4 %4, %9, %10\n\t"
      "global_load_dwordx4 %1, %8, %10 offset:16\n\t"
      "global_load_dwordx4 %2, %8, %10 offset:32\n\t"
      "global_load_dwordx4 %3, %8, %10 offset:48\n\t"
      "global_load_dwordx4 %5, %9, %10 offset:16\n\t"
      "global_load_dwordx4 %6, %9, %10 offset:32\n\t"
      "global_load_dwordx4 %7, %9, %10 offset:48\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.k0), "=&v"(a.k1), "=&v"(a.k2), "=&v"(a.k3), "=&v"(b.k0), "=&v"(b.k1), "=&v"(b.k2), "=&v"(b.k3)
      : "v"(offa), "v"(offb), "s"(gbase)
      : "memory");
}
__device__ __forceinline__ void chunk_count(const v4u& k, int take, u32& a, u32& c, u32& g, u32& t) {
  u32 m = take >= 32 ? 0xFFFFFFFFu : (take <= 0 ? 0u : ((1u << take) - 1u));
  u32 x0 = k.y & m, x1 = k.z & m, x2 = k.w & m;
  a += __popc(x0 & ~x1);
  c += __popc(x1 & ~x0);
  g += __popc(x0 & x1);
  t += __popc(x2);
}
// fm_rank4p on a granule that is already in registers; pc = the clamped position
template <bool WIDE>
__device__ __forceinline__ Cnt4P<typename PosOf<WIDE>::type> fm_rank4p_from(const FmRef& s, u64 pc, const Gran& q) {
  typedef typename PosOf<WIDE>::type P;
  int r = (int)(pc & 127u);
  u32 a = q.k0.x, c = q.k1.x, g = q.k2.x, t = q.k3.x;
  chunk_count(q.k0, r, a, c, g, t);
  chunk_count(q.k1, r - 32, a, c, g, t);
  chunk_count(q.k2, r - 64, a, c, g, t);
  chunk_count(q.k3, r - 96, a, c, g, t);
  Cnt4P<P> o;
  o.a = a; o.c = c; o.g = g; o.t = t;
  if (WIDE) {
    const u64* sb = s.super + (pc >> SIGAX_SUPER_SHIFT) * 4;
    o.a += (P)sb[0]; o.c += (P)sb[1]; o.g += (P)sb[2]; o.t += (P)sb[3];
  }
  return o;
}

// BWT symbol at position i (FMIndex::getChar, src/fmindex.cpp:233-246)
__device__ __forceinline__ u32 fm_char(const FmRef& s, u64 i) {
  const u32* q = reinterpret_cast<const u32*>(s.g) + (i >> 7) * 16 + ((i >> 5) & 3) * 4;
  u32 b = (u32)i & 31u;
  return ((q[1] >> b) & 1u) | (((q[2] >> b) & 1u) << 1) | (((q[3] >> b) & 1u) << 2);
}

// alphabet.h:19-39 and kseq.cpp:18-27: byte -> rank; complement in rank space (non-ACGT -> 0 either way)
// Branch-free (a chain of equality tests on one value is lowered to a divergent branch tree): bits 1-2 of 'A','C','T','G'
// are 0,1,2,3; the byte expected at that index and its rank come out of two constants.
__device__ __forceinline__ u32 base_rank(u32 ch) {
  const u32 i = (ch >> 1) & 3u;
  const u32 expect = (0x47544341u >> (i * 8)) & 0xFFu;  // "ACTG"
  const u32 rank = (0x3421u >> (i * 4)) & 0xFu;         // 1, 2, 4, 3
  return expect == ch ? rank : 0u;
}
// v[r] for r in 0..4 out of registers, as a select tree on the bits of r (again: no equality chain, no branches)
template <typename T>
__device__ __forceinline__ T sel5(u32 r, T v0, T v1, T v2, T v3, T v4) {
  const bool b0 = (r & 1u) != 0, b1 = (r & 2u) != 0, b2 = (r & 4u) != 0;
  const T t01 = b0 ? v1 : v0, t23 = b0 ? v3 : v2;
  const T t = b1 ? t23 : t01;
  return b2 ? v4 : t;
}
__device__ __forceinline__ u32 comp_rank(u32 r) { return r ? 5u - r : 0u; }

__device__ __forceinline__ u64 wave_sum(u64 v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// -------------------------------------------------------------------------------------------------------
// k_occ_batch / k_kmer_count
// -------------------------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void k_occ_batch(FmStrand s, const u64* pos, u64 n, u64* out) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u64 v[5];
  fm_rank5<WIDE>(fm_ref(s, 0), pos[i] + 1, v);  // fmindex.cpp:191: ++i, so 2^64-1 wraps to the empty prefix
  for (int k = 0; k < 5; ++k) out[i * 5 + k] = v[k];
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_kmer_count(FmStrand s, const unsigned char* kmers, u32 k, u64 n, u64* out) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned char* w = kmers + i * k;
  FmRef f = fm_ref(s, 0);
  u64 C[5] = {s.C[0], s.C[1], s.C[2], s.C[3], s.C[4]};
  u64 T[5] = {s.total[0], s.total[1], s.total[2], s.total[3], s.total[4]};
  auto sel = [](const u64 v[5], u32 r) { return r == 0 ? v[0] : r == 1 ? v[1] : r == 2 ? v[2] : r == 3 ? v[3] : v[4]; };
  // Interval::get (fmindex.h:67-79): init with the last symbol, update while valid
  u32 r = base_rank(w[k - 1]);
  u64 lo = sel(C, r), hi = lo + sel(T, r) - 1;
  for (u32 j = k - 1; j > 0; --j) {
    if (!(hi != ~0ull && hi >= lo)) break;
    r = base_rank(w[j - 1]);
    u64 l[5], u[5];
    fm_rank5<WIDE>(f, lo, l);       // getOcc(c, lower - 1)
    fm_rank5<WIDE>(f, hi + 1, u);   // getOcc(c, upper)
    lo = sel(C, r) + sel(l, r);
    hi = sel(C, r) + sel(u, r) - 1;
  }
  out[i] = (hi != ~0ull && hi >= lo) ? hi - lo + 1 : 0;
}

// -------------------------------------------------------------------------------------------------------
// Row table + stretch text of fm_layout.h.  One lane per symbol of rank 0 in the text (a read's terminator, or a non-ACGT
// base, which the index stores as rank 0 too: alphabet.h:19-39): from the row of the suffix that starts there (rows
// 0 .. C['A']-1) walk backwards (LF) to the row whose BWT symbol has rank 0.  k_stretch_scan learns the distance and
// Occ('$') there (and the longest distance of the index, which sizes the entries); k_rows_fill walks again and writes
// (steps still to go, that Occ) at every row passed and the symbols passed into the stretch's text row.  Every row lies on
// exactly one such walk.
// -------------------------------------------------------------------------------------------------------
template <bool WIDE>
__device__ __forceinline__ u32 lf_row(const FmRef& f, const u64* C, u64 p, u64* next) {
  const u32 c = fm_char(f, p);
  const Cnt4 k = fm_rank<WIDE>(f, p);
  if (c == 0) *next = p - (k.a + k.c + k.g + k.t);  // Occ('$', p - 1)
  else *next = C[c] + (c == 1 ? k.a : c == 2 ? k.c : c == 3 ? k.g : k.t);
  return c;
}
#define STRETCH_BAD (~0ull)  // not a BWT of '$'-terminated reads: the walk did not end
template <bool WIDE>
__global__ __launch_bounds__(256) void k_stretch_scan(FmStrand s, u64 n_stretch, u64* info, u32* maxlen) {
  const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
  u32 mine = 0;
  if (j < n_stretch) {
    const FmRef f = fm_ref(s, 0);
    const u64 C[5] = {s.C[0], s.C[1], s.C[2], s.C[3], s.C[4]};
    u64 p = j, len = 0, v = STRETCH_BAD;
    for (;;) {
      u64 q;
      if (lf_row<WIDE>(f, C, p, &q) == 0) { v = (len << 32) | (q & 0xFFFFFFFFull); break; }
      p = q;
      if (++len >= s.n || len >= (1ull << 28)) { len = 0; break; }
    }
    info[j] = v;
    mine = (u32)len;
  }
  for (int off = 32; off > 0; off >>= 1) mine = max(mine, (u32)__shfl_xor((int)mine, off, 64));
  if ((threadIdx.x & 63u) == 0 && mine) atomicMax(maxlen, mine);
}
// entry p of a bit-packed table of `bits`-bit values (zeroed before): set / get
__device__ __forceinline__ void packed_or(unsigned char* tab, u64 p, u32 bits, u64 v) {
  const u64 B = p * bits;
  u64* w = reinterpret_cast<u64*>(tab) + (B >> 6);
  const u32 sh = (u32)B & 63u;
  atomicOr(w, v << sh);
  if (sh + bits > 64u) atomicOr(w + 1, v >> (64u - sh));
}
__device__ __forceinline__ u64 packed_get(const unsigned char* tab, u64 p, u32 bits) {
  const u64 B = p * bits;
  u64 v;
#if (SIGAX_NT & 1)
  {  // the row-table line is used once
    typedef u64 __attribute__((aligned(1))) u64u;
    v = __builtin_nontemporal_load(reinterpret_cast<const u64u*>(tab + (B >> 3)));
  }
#else
  __builtin_memcpy(&v, tab + (B >> 3), 8);  // one unaligned 8-byte load (the table is padded by 8 bytes)
#endif
  return (v >> ((u32)B & 7u)) & ((1ull << bits) - 1ull);
}
// (stretch, offset) of the suffix at `row` of a strand that has the row table; returns the entry's symbols (fm_layout.h)
__device__ __forceinline__ u64 row_lookup(const FmStrand& st, u64 row, u32& ld, u32& t) {
  const u64 v = packed_get(st.sa, row < st.n ? row : 0ull, st.sa_bits);  // never leave the table, whatever a block holds
  ld = (u32)(v & ((1ull << st.ld_bits) - 1ull));
  t = (u32)(v >> st.ld_bits) & ((1u << st.t_bits) - 1u);
  return v >> (st.ld_bits + st.t_bits);
}
// The symbols on the backward path from offset t of stretch ld, 2 bits each (rank - 1), next one in the top two bits:
// offsets t-1, t-2, ... -- at least SIGAX_TEXT_WINDOW of them, or all t (the caller knows from t where the read ends).
__device__ __forceinline__ u64 text_window(const FmStrand& st, u32 ld, u32 t) {
  if (t == 0) return 0ull;
  const u32 end = 2u * t;                                  // the row's bits [0, end) hold offsets 0 .. t-1
  const u32 b = end > 64u ? (end - 64u + 7u) >> 3 : 0u;  // first byte of the 8 that end at or past bit `end`
  u64 w;
  __builtin_memcpy(&w, st.text + (u64)ld * st.text_stride + b, 8);  // rows are padded (fm_layout.h)
  return w << (64u - (end - 8u * b));
}
// -------------------------------------------------------------------------------------------------------
// Self-check of an index against the order of record (sigax_index_check_order): the row table IS the suffix array, the
// stretch text IS the reads, so whether the BWT rows are in suffix order can be seen on the device without a second
// suffix sort: one lane per pair of adjacent rows compares the two suffixes of r0 $ r1 $ ... r(n-1) $ -- one '$' smaller
// than A, C, G, T, comparisons running on past a '$' into the next read, the end of the text smallest (the order `siga
// index` produces: SURVEY.md App. C model B; src/suffix_array_builder.cpp:472-674).  28 symbols per 8-byte load.
// -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 text_bits(const FmStrand& st, u32 ld, u32 off) {  // symbols off, off+1, ... lowest first
  u64 w;
  __builtin_memcpy(&w, st.text + (u64)ld * st.text_stride + (off >> 2), 8);
  return w >> (2u * (off & 3u));
}
// direct map of strand X as extension index (fm_layout.h): sai_y = the other strand's .sai ids, isai_x = inverse of X's
__global__ __launch_bounds__(256) void k_xmap(const u32* sai_y, const u32* isai_x, const u32* slen_x, u64 n, u64* xmap) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u32 ldx = isai_x[sai_y[i]];
  xmap[i] = (u64)ldx | ((u64)slen_x[ldx] << 32);
}
__global__ __launch_bounds__(256) void k_isai(const u32* sai, u64 n, u32* isai) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) isai[sai[i]] = (u32)i;
}
__global__ __launch_bounds__(256) void k_suffix_order_check(FmStrand st, const u32* sai, const u32* isai, const u32* read_len, u64 n_strings,
                                                            u64* bad) {
  const u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
  if (p + 1 >= st.n) return;
  u32 la, ta, lb, tb;
  row_lookup(st, p, la, ta);
  row_lookup(st, p + 1, lb, tb);
  int verdict = -1;  // 1: suffix(p) < suffix(p + 1); 0: not
  for (u32 hops = 0; verdict < 0 && hops < 4096u; ++hops) {
    const u32 ida = sai[la], idb = sai[lb];
    const u32 ra = read_len[ida] - ta, rb = read_len[idb] - tb;  // symbols left in each read
    const u32 m = ra < rb ? ra : rb;
    for (u32 k = 0; k < m && verdict < 0;) {
      const u32 c = m - k < 28u ? m - k : 28u;
      const u64 x = (text_bits(st, la, ta + k) ^ text_bits(st, lb, tb + k)) & ((1ull << (2u * c)) - 1ull);
      if (x) {
        const u32 d = (u32)__builtin_ctzll(x) >> 1;
        const u32 sa = (u32)(text_bits(st, la, ta + k + d) & 3ull), sb = (u32)(text_bits(st, lb, tb + k + d) & 3ull);
        verdict = sa < sb ? 1 : 0;
      }
      k += c;
    }
    if (verdict >= 0) break;
    if (ra != rb) { verdict = ra < rb ? 1 : 0; break; }  // a '$' against a base
    // both reads end here: '$' against '$', on into the reads that follow them in the text (the end of the text is smallest)
    if ((u64)ida + 1 >= n_strings || (u64)idb + 1 >= n_strings) { verdict = ((u64)ida + 1 >= n_strings && (u64)idb + 1 < n_strings) ? 1 : 0; break; }
    la = isai[ida + 1]; ta = 0;
    lb = isai[idb + 1]; tb = 0;
  }
  if (verdict == 0) {
    atomicAdd(&bad[0], 1ull);
    atomicMin(&bad[1], p);
  } else if (verdict < 0) {
    atomicAdd(&bad[2], 1ull);  // undecided after 4096 reads' worth of ties (thousands of identical reads in a row)
  }
}

#define ROWS_KMAX 14  // most symbols an entry can carry (fm_layout.h)
template <bool WIDE>
__global__ __launch_bounds__(256) void k_rows_fill(FmStrand s, u64 n_stretch, const u64* info, unsigned char* sa, u32 sa_bits, u32 ld_bits, u32 t_bits,
                                                   unsigned char* text, u32 text_stride, u32* slen) {
  const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_stretch) return;
  const u64 v = info[j];
  if (v == STRETCH_BAD) return;  // rows of such a walk keep entry 0: "ends here, stretch 0" is never consulted then
  const FmRef f = fm_ref(s, 0);
  const u64 C[5] = {s.C[0], s.C[1], s.C[2], s.C[3], s.C[4]};
  const u64 ld = v & 0xFFFFFFFFull;
  if (slen != nullptr) slen[ld] = (u32)(v >> 32);  // the stretch's length, by its rank among the '$' rows
  const u32 K = sa != nullptr ? (sa_bits - ld_bits - t_bits) >> 1 : 0u;  // symbols carried by an entry
  u64 p = j;
  unsigned char* trow = text ? text + ld * text_stride : nullptr;
  // A window of the last ROWS_KMAX rows slides along: when the K-th symbol after a row has been seen its entry is complete
  // (R[ROWS_KMAX - 1] = the newest row; wsym = the symbols seen as rank - 1, newest in the lowest two bits, so the K
  // symbols that follow the row K places back read first-symbol-highest: the order text_window() hands them out in).
  u64 R[ROWS_KMAX];
  u32 wsym = 0, vm = 0;
#pragma unroll
  for (int i = 0; i < ROWS_KMAX; ++i) R[i] = 0;
  auto insert = [&](u64 row, u32 c, bool real, u64 t_of_row) {
#pragma unroll
    for (int i = 0; i + 1 < ROWS_KMAX; ++i) R[i] = R[i + 1];
    R[ROWS_KMAX - 1] = row;
    wsym = (wsym << 2) | ((c - 1u) & 3u);
    vm = (vm << 1) | (real ? 1u : 0u);
    if (K == 0 || sa == nullptr) return;
    u64 r0 = R[0];
#pragma unroll
    for (int i = 1; i < ROWS_KMAX; ++i)
      if ((u32)(ROWS_KMAX - i) == K) r0 = R[i];
    if ((vm >> (K - 1u)) & 1u) {
      const u64 syms = (u64)wsym & ((1ull << (2u * K)) - 1ull);
      packed_or(sa, r0, sa_bits, (syms << (ld_bits + t_bits)) | ((t_of_row + (K - 1u)) << ld_bits) | ld);
    }
  };
  for (u64 t = v >> 32;; --t) {
    if (K == 0 && sa != nullptr) packed_or(sa, p, sa_bits, (t << ld_bits) | ld);
    u64 q;
    const u32 c = lf_row<WIDE>(f, C, p, &q);  // the symbol at offset t - 1; rank 0 exactly when t == 0
    insert(p, c, true, t);
    if (t == 0) break;
    if (trow) packed_or(trow, t - 1, 2u, (u64)((c - 1u) & 3u));  // offset t - 1 at bits 2 (t - 1) .. of the stretch's own (zeroed) row
    p = q;
  }
  // the rows whose K symbols run past the read's first base: rank 0 from there on (t counts on below zero for the formula)
  for (u32 i = 1; i < K; ++i) insert(0, 0, false, 0ull - i);
}

// -------------------------------------------------------------------------------------------------------
// Start table of the block finder (fm_layout.h): one lane per 12-mer walks IntervalPair::init + eleven updateL steps
// (src/overlap_builder.cpp:91-122) with `prim` as the primary index.
// -------------------------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void k_start_build(FmStrand prim, FmStrand other, void* tab) {
  typedef typename PosOf<WIDE>::type P;
  const u32 code = blockIdx.x * 256 + threadIdx.x;
  if (code >= (1u << (2 * SIGAX_START_K))) return;
  const FmRef f = fm_ref(prim, 0);
  u32 r = 1u + ((code >> (2 * (SIGAX_START_K - 1))) & 3u);
  P lo0 = (P)prim.C[r], sz = (P)prim.total[r], lo1 = (P)other.C[r];
  u32 s = 1;
  for (int i = 1; i < SIGAX_START_K && sz != 0; ++i, ++s) {
    r = 1u + ((code >> (2 * (SIGAX_START_K - 1 - i))) & 3u);
    const u64 n = prim.n;
    const u64 pl = (u64)lo0 > n ? n : (u64)lo0, pu0 = (u64)(P)(lo0 + sz), pu = pu0 > n ? n : pu0;
    const Cnt4P<P> l = fm_rank4p<WIDE>(f, (P)pl), u = fm_rank4p<WIDE>(f, (P)pu);
    const P da = u.a - l.a, dc = u.c - l.c, dg = u.g - l.g, dt = u.t - l.t;
    const P dd = sz - (da + dc + dg + dt);
    const P acc = r == 1 ? dd : r == 2 ? dd + da : r == 3 ? dd + da + dc : dd + da + dc + dg;
    const P lc = r == 1 ? l.a : r == 2 ? l.c : r == 3 ? l.g : l.t;
    const P dcur = r == 1 ? da : r == 2 ? dc : r == 3 ? dg : dt;
    lo1 += acc;
    lo0 = (P)prim.C[r] + lc;
    sz = dcur;
  }
  if (WIDE) {
    ulonglong2* q = reinterpret_cast<ulonglong2*>(tab) + 2ull * code;
    q[0] = make_ulonglong2((u64)lo0, (u64)lo1);
    q[1] = make_ulonglong2((u64)sz, (u64)s);
  } else {
    reinterpret_cast<uint4*>(tab)[code] = make_uint4((u32)lo0, (u32)lo1, (u32)sz, s);
  }
}

// -------------------------------------------------------------------------------------------------------
// Deep start table of the block finder (fm_layout.h): the distinct K-mers of a strand's text are the runs of equal
// K-symbol prefixes among adjacent rows of its row table (k_deep_scan counts them, then lists each run's first row); one
// lane per distinct K-mer then walks IntervalPair::init + K - 1 updateL steps (src/overlap_builder.cpp:91-122) with that
// strand as primary index -- the arithmetic of k_start_build and of the finder itself -- and puts the state into the hash
// table (k_deep_fill).  The walk must come out at the run's own first row: anything else (an index whose rows are not in
// suffix order, a row table of a walk that did not end) raises *err and the host throws the table away.
// -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 deep_slot(u64 k0, u64 k1, u64 nslots) {
  u64 h = (k0 ^ (k1 * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
  h ^= h >> 31;
  h *= 0x94D049BB133111EBull;
  h ^= h >> 29;
  return __umul64hi(h, nslots);
}
// the first K symbols of the suffix at row p in text order, symbol j at bits 2 j of f0 (j < 28) / 2 (j - 28) of f1; false
// when the row's stretch ends before that
__device__ __forceinline__ bool deep_row_kmer(const FmStrand& st, const u32* slen, u64 n_stretch, u64 p, u32 K, u64& f0, u64& f1) {
  u32 ld, t;
  row_lookup(st, p, ld, t);
  f0 = f1 = 0;
  if ((u64)ld >= n_stretch) return false;
  if ((u64)t + K > (u64)slen[ld]) return false;
  const u64 m56 = (1ull << 56) - 1ull;
  f0 = text_bits(st, ld, t) & (K >= 28u ? m56 : ((1ull << (2u * K)) - 1ull));
  if (K > 28u) f1 = text_bits(st, ld, t + 28u) & (K >= 56u ? m56 : ((1ull << (2u * (K - 28u))) - 1ull));
  return true;
}
__global__ __launch_bounds__(256) void k_deep_scan(FmStrand st, const u32* slen, u64 n_stretch, u32 K, u64* count, u64* list, u64 list_cap) {
  const u32 lane = threadIdx.x & 63u;
  // waves walk the rows in strides of the grid (a launch holds fewer than 2^32 threads; BASELINE configs[4] has 1.3e10 rows)
  for (u64 p0 = (u64)blockIdx.x * 256; p0 < st.n; p0 += (u64)gridDim.x * 256) {
    const u64 p = p0 + threadIdx.x;
    u64 f0 = 0, f1 = 0;
    bool v = false;
    if (p < st.n) v = deep_row_kmer(st, slen, n_stretch, p, K, f0, f1);
    // the row before: the lane below has it, except for the wave's first lane
    u64 g0 = __shfl_up(f0, 1, 64), g1 = __shfl_up(f1, 1, 64);
    bool pv = __shfl_up((int)v, 1, 64) != 0;
    if (lane == 0) {
      pv = false;
      if (p > 0 && p < st.n) pv = deep_row_kmer(st, slen, n_stretch, p - 1, K, g0, g1);
    }
    const bool first = v && (!pv || g0 != f0 || g1 != f1);
    const u64 m = __ballot(first);
    if (!m) continue;
    u64 base = 0;
    if (lane == 0) base = atomicAdd(count, (u64)__popcll(m));
    base = __shfl(base, 0, 64);
    if (list != nullptr && first) {
      const u64 i = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
      if (i < list_cap) list[i] = p;
    }
  }
}
template <bool WIDE>
__global__ __launch_bounds__(256) void k_deep_fill(FmStrand prim, FmStrand other, const u32* slen, u64 n_stretch, u32 K, const u64* list, u64 n_list,
                                                   u64* tab, u64 nslots, u64* err) {
  typedef typename PosOf<WIDE>::type P;
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;  // (fewer than 2^32 distinct K-mers: the host checks)
  if (i >= n_list) return;
  const u64 p = list[i];
  u64 f0, f1;
  if (!deep_row_kmer(prim, slen, n_stretch, p, K, f0, f1)) {
    atomicAdd(err, 1ull);
    return;
  }
  auto sym = [&](u32 j) -> u32 { return (u32)((j < 28u ? f0 >> (2u * j) : f1 >> (2u * (j - 28u))) & 3ull); };
  const FmRef f = fm_ref(prim, 0);
  // the chain consumes the K-mer from its last symbol backwards
  u32 r = 1u + sym(K - 1u);
  P lo0 = (P)prim.C[r], sz = (P)prim.total[r], lo1 = (P)other.C[r];
  u64 k0 = (u64)(r - 1u), k1 = 0;
  for (u32 c = 1; c < K; ++c) {
    r = 1u + sym(K - 1u - c);
    if (c < 32u) k0 |= (u64)(r - 1u) << (2u * c);
    else k1 |= (u64)(r - 1u) << (2u * (c - 32u));
    if (sz == 0) continue;
    const u64 n = prim.n;
    const u64 pl = (u64)lo0 > n ? n : (u64)lo0, pu0 = (u64)(P)(lo0 + sz), pu = pu0 > n ? n : pu0;
    const Cnt4P<P> l = fm_rank4p<WIDE>(f, (P)pl), u = fm_rank4p<WIDE>(f, (P)pu);
    const P da = u.a - l.a, dc = u.c - l.c, dg = u.g - l.g, dt = u.t - l.t;
    const P dd = sz - (da + dc + dg + dt);
    const P acc = r == 1 ? dd : r == 2 ? dd + da : r == 3 ? dd + da + dc : dd + da + dc + dg;
    const P lc = r == 1 ? l.a : r == 2 ? l.c : r == 3 ? l.g : l.t;
    const P dcur = r == 1 ? da : r == 2 ? dc : r == 3 ? dg : dt;
    lo1 += acc;
    lo0 = (P)prim.C[r] + lc;
    sz = dcur;
  }
  if (sz == 0 || (u64)lo0 != p || (WIDE && (((u64)lo0 | (u64)lo1 | (u64)sz) >> 40) != 0)) {
    atomicAdd(err, 1ull);
    return;
  }
  k1 |= (u64)(0x8000u | K) << 48;
  u64 slot = deep_slot(k0, k1, nslots);
  for (u64 tries = 0; tries < nslots; ++tries) {
    if (atomicCAS(&tab[slot * 4 + 1], 0ull, k1) == 0ull) {
      tab[slot * 4] = k0;
      if (WIDE) {
        tab[slot * 4 + 2] = (u64)lo0 | ((u64)lo1 << 40);
        tab[slot * 4 + 3] = ((u64)lo1 >> 24) | ((u64)sz << 16);
      } else {
        tab[slot * 4 + 2] = (u64)(u32)lo0 | ((u64)(u32)lo1 << 32);
        tab[slot * 4 + 3] = (u64)(u32)sz;
      }
      return;
    }
    slot = slot + 1 == nslots ? 0 : slot + 1;
  }
  atomicAdd(err, 1ull);
}
// the chain's state after the K-mer (k0, k1 = its code with the tag of fm_layout.h), or false: not a K-mer of the text
template <bool WIDE>
__device__ __forceinline__ bool deep_lookup(const void* tab, u64 nslots, u64 k0, u64 k1, typename PosOf<WIDE>::type& lo0,
                                            typename PosOf<WIDE>::type& lo1, typename PosOf<WIDE>::type& sz) {
  typedef typename PosOf<WIDE>::type P;
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(tab);
  u64 slot = deep_slot(k0, k1, nslots);
  for (;;) {
    const ulonglong2 key = q[2 * slot], pay = q[2 * slot + 1];
    if (key.y == 0ull) return false;  // an empty slot ends the probe sequence (the table is never full)
    if (key.x == k0 && key.y == k1) {
      if (WIDE) {
        const u64 m40 = (1ull << 40) - 1ull;
        lo0 = (P)(pay.x & m40);
        lo1 = (P)((pay.x >> 40) | ((pay.y & 0xFFFFull) << 24));
        sz = (P)((pay.y >> 16) & m40);
      } else {
        lo0 = (P)(u32)pay.x;
        lo1 = (P)(u32)(pay.x >> 32);
        sz = (P)(u32)pay.y;
      }
      return true;
    }
    slot = slot + 1 == nslots ? 0 : slot + 1;
  }
}

// -------------------------------------------------------------------------------------------------------
// k_find: one lane per (read, orientation) chain; the four waves of a workgroup are the four chains of 64 reads.  Per step: two rank granules on the chain's primary index
// (positions lower-1 and upper of IntervalPair::updateL, src/overlap_builder.cpp:95-122); the '$' probe of
// src/overlap_builder.cpp:861-871 reuses them.  Blocks go to the chain's slots of the candidate arena in
// increasing overlap length = the reference's push order.
// -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_block(sigax_block* dst, u64 c0lo, u64 c0hi, u64 c1lo, u64 c1hi, u64 r0lo, u64 r0hi,
                                            u64 r1lo, u64 r1hi, u32 len, u32 af) {
  ulonglong2* d = reinterpret_cast<ulonglong2*>(dst);
  d[0] = make_ulonglong2(c0lo, c0hi);
  d[1] = make_ulonglong2(c1lo, c1hi);
  d[2] = make_ulonglong2(r0lo, r0hi);
  d[3] = make_ulonglong2(r1lo, r1hi);
  d[4] = make_ulonglong2((u64)len | ((u64)af << 32), 0ull);
}

// ---- two-step table (fm_layout.h) -------------------------------------------------------------------------------
// built on the device from the one-step granules: one wave per 64 rows
template <bool WIDE>
__global__ __launch_bounds__(256) void k_build2_a(FmStrand s, u32* gran2, u32* cnt, u64 ng2) {
  const FmRef F = fm_ref(s, 0);
  const u32 lane = threadIdx.x & 63u;
  const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (u64)gridDim.x * 4;
  for (u64 g = wave; g < ng2; g += nw) {
    const u64 p = g * 64 + lane;
    u32 c1 = 0, c2 = 0;
    if (p < s.n) {
      c1 = fm_char(F, p);
      if (c1 >= 1 && c1 <= 4) {
        const Cnt4 k = fm_rank<WIDE>(F, p);
        const u64 rk = c1 == 1 ? k.a : c1 == 2 ? k.c : c1 == 3 ? k.g : k.t;
        c2 = fm_char(F, s.C[c1] + rk);  // BWT[LF(p)]
      } else {
        c1 = c1 > 4 ? 0 : c1;
      }
    }
    const u64 y1 = __ballot((c1 & 1u) != 0), z1 = __ballot((c1 & 2u) != 0), w1 = __ballot((c1 & 4u) != 0);
    const u64 y2 = __ballot((c2 & 1u) != 0), z2 = __ballot((c2 & 2u) != 0), w2 = __ballot((c2 & 4u) != 0);
    u32 mine = 0;  // lane t < 20 ends up with the granule's count for header word t
    for (u32 b = 1; b <= 4; ++b) {
      const u32 v = (u32)__popcll(__ballot(c1 == b));
      if (lane == b - 1) mine = v;
    }
    for (u32 c = 1; c <= 4; ++c)
      for (u32 x = 1; x <= 4; ++x) {
        const u32 v = (u32)__popcll(__ballot(c1 == c && c2 == x));
        if (lane == 4 + (c - 1) * 4 + (x - 1)) mine = v;
      }
    if (lane < 20) cnt[(u64)lane * ng2 + g] = mine;
    if (lane == 0) {
      u32* q = gran2 + g * SIGAX_GRAN2_WORDS + 20;
      q[0] = (u32)y1; q[1] = (u32)(y1 >> 32); q[2] = (u32)z1; q[3] = (u32)(z1 >> 32); q[4] = (u32)w1; q[5] = (u32)(w1 >> 32);
      q[6] = (u32)y2; q[7] = (u32)(y2 >> 32); q[8] = (u32)z2; q[9] = (u32)(z2 >> 32); q[10] = (u32)w2; q[11] = (u32)(w2 >> 32);
    }
  }
}
// counter `col` of every line; with 64-bit positions relative to the line's superblock, whose base goes to super2
__global__ __launch_bounds__(256) void k_build2_b(const u64* offs, u32* gran2, u32 col, u64 ng2, u64* super2) {
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  if (g >= ng2) return;
  u64 base = 0;
  if (super2) {
    const u32 sh = SIGAX_SUPER_SHIFT - 6;  // lines per superblock = 2^sh
    const u64 first = (g >> sh) << sh;
    base = offs[first];
    if (g == first) super2[(g >> sh) * 20 + col] = base;
  }
  gran2[g * SIGAX_GRAN2_WORDS + col] = (u32)(offs[g] - base);
}

#ifndef SIGAX_FIND_POLICY
#define SIGAX_FIND_POLICY ""  // cache-policy bits of the finder's table loads (A/B builds: " nt", " sc1", " sc0 sc1")
#endif
struct Gran2 {  // the five 16-byte pieces of a two-step granule one step needs
  v4u s, pc, p5, p6, p7;  // one-symbol counts; pair counts [first symbol c][x = A..T]; planes
};
// both positions of a double step: ten loads back to back, one wait (u32 offsets from a scalar base: table < 4 GiB)
__device__ __forceinline__ void find_step2_loads(const void* base, u32 offa, u32 offa_c, u32 offb, u32 offb_c, Gran2& a, Gran2& b) {
  asm volatile(
      "global_load_dwordx4 %0, %10, %14" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %5, %12, %14" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %1, %11, %14" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %2, %10, %14 offset:80" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %3, %10, %14 offset:96" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %4, %10, %14 offset:112" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %6, %13, %14" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %7, %12, %14 offset:80" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %8, %12, %14 offset:96" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %9, %12, %14 offset:112" SIGAX_FIND_POLICY "\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.s), "=&v"(a.pc), "=&v"(a.p5), "=&v"(a.p6), "=&v"(a.p7), "=&v"(b.s), "=&v"(b.pc), "=&v"(b.p5), "=&v"(b.p6),
        "=&v"(b.p7)
      : "v"(offa), "v"(offa_c), "v"(offb), "v"(offb_c), "s"(base)
      : "memory");
}
// The same with the upper position's five loads issued only for the lanes in `two` (a wave mask): five chains in six have
// both positions in one line, and a load instruction costs the CU's L1 one tag lookup per ACTIVE lane -- the per-lane
// finder's time goes with the CUs it runs on (tools/cu_split.sh), not with what the memory behind them can deliver.
// The other lanes' b is undefined: the caller copies a.
__device__ __forceinline__ void find_step2_loads_masked(const void* base, u32 offa, u32 offa_c, u32 offb, u32 offb_c, u64 two, Gran2& a,
                                                        Gran2& b) {
  u64 saved;
  asm volatile(
      "global_load_dwordx4 %0, %11, %15" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %1, %12, %15" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %2, %11, %15 offset:80" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %3, %11, %15 offset:96" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %4, %11, %15 offset:112" SIGAX_FIND_POLICY "\n\t"
      "s_and_saveexec_b64 %10, %16\n\t"
      "global_load_dwordx4 %5, %13, %15" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %6, %14, %15" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %7, %13, %15 offset:80" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %8, %13, %15 offset:96" SIGAX_FIND_POLICY "\n\t"
      "global_load_dwordx4 %9, %13, %15 offset:112" SIGAX_FIND_POLICY "\n\t"
      "s_mov_b64 exec, %10\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.s), "=&v"(a.pc), "=&v"(a.p5), "=&v"(a.p6), "=&v"(a.p7), "=&v"(b.s), "=&v"(b.pc), "=&v"(b.p5), "=&v"(b.p6),
        "=&v"(b.p7), "=&s"(saved)
      : "v"(offa), "v"(offa_c), "v"(offb), "v"(offb_c), "s"(base), "s"(two)
      : "memory", "scc");
}
// The same through LDS, cooperatively (the finder for tables beyond the translation reach of per-lane gathers): eight
// lanes fetch the eight 16-byte pieces of ONE 128-byte line with one instruction that writes LDS directly
// (global_load_lds_dwordx4: destination = wave-uniform base + lane x 16), so a wave instruction touches 8 lines instead
// of 64.  Every chain's lower line goes to region A (8 instructions for the wave's 64 chains); five chains in six have
// their upper position in the same line, so only the chains whose upper line differs are listed (ballot + prefix count)
// and fetched into the small region B, sixteen chains per pass (almost always one pass of two instructions).  Then
// every lane reads the five pieces it needs of its own line(s).  Slot of (chain 8t + j, piece p) in instruction t's 1 KiB:
// (p ^ (t & 1)) * 8 + j, which spreads the sixteen lanes of a ds_read_b128 group over the sixteen 16-byte bank groups.
// la / lb = line index of the lane's lower / upper position (0 for idle lanes), c = first symbol (1..4).
#define COOP_A_U4 512u   // region A: 64 chains x 8 pieces
#define COOP_B_U4 128u   // region B: 16 chains x 8 pieces per pass
#define COOP_WAVE_U4 (COOP_A_U4 + COOP_B_U4 + 16u)  // + 64 u32 of list
__device__ __forceinline__ void find_step2_dma(const uint32_t* gran2, u32 la, u32 lb, u32 c, uint4* stage, u32 lane, Gran2& a, Gran2& b) {
  const u32 j = lane & 7u, pp = lane >> 3;
  const char* base = reinterpret_cast<const char*>(gran2);
  u32* list = reinterpret_cast<u32*>(stage + COOP_A_U4 + COOP_B_U4);
  const bool differ = lb != la;
  const u64 dmask = __ballot(differ);
  const u32 nd = (u32)__popcll(dmask);
  const u32 r = (u32)__popcll(dmask & ((1ull << lane) - 1ull));  // my place among the chains with a second line
  // the records parked by the previous step have been read back (find_flush) and the previous step's pieces consumed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (differ) list[r] = lb;
#pragma unroll
  for (u32 t = 0; t < 8; ++t) {
    const u32 lA = (u32)__shfl((int)la, (int)(8u * t + j), 64);
    const u32 p = pp ^ (t & 1u);
    const char* sa = base + (u64)lA * 128u + p * 16u;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                     (__attribute__((address_space(3))) void*)(u32)(size_t)(stage + t * 64u), 16, 0, 0);
  }
  const u32 tt = lane >> 3, x = tt & 1u;
  auto ld = [](const uint4* q) { const uint4 v = *q; v4u w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; return w; };
  bool gotA = false;
  const u32 npass = (nd + 15u) >> 4;
  for (u32 ps = 0; ps < npass || !gotA; ++ps) {  // wave-uniform trip count; at least once, for region A
    if (ps < npass) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the list is written; the previous pass's readers are done
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (u32 t = 0; t < 2; ++t) {
        const u32 rk = 16u * ps + 8u * t + j;
        if (rk < nd) {
          const u32 lB = list[rk];
          const u32 p = pp ^ (t & 1u);
          const char* sb = base + (u64)lB * 128u + p * 16u;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                           (__attribute__((address_space(3))) void*)(u32)(size_t)(stage + COOP_A_U4 + t * 64u), 16, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!gotA) {
      const uint4* qa = stage + tt * 64u + j;
      a.s = ld(qa + ((0u ^ x) * 8u)); a.pc = ld(qa + ((c ^ x) * 8u)); a.p5 = ld(qa + ((5u ^ x) * 8u)); a.p6 = ld(qa + ((6u ^ x) * 8u)); a.p7 = ld(qa + ((7u ^ x) * 8u));
      gotA = true;
    }
    if (differ && (r >> 4) == ps) {
      const u32 q = r & 15u, tb = q >> 3, xb = tb & 1u;
      const uint4* qb = stage + COOP_A_U4 + tb * 64u + (q & 7u);
      b.s = ld(qb + ((0u ^ xb) * 8u)); b.pc = ld(qb + ((c ^ xb) * 8u)); b.p5 = ld(qb + ((5u ^ xb) * 8u)); b.p6 = ld(qb + ((6u ^ xb) * 8u)); b.p7 = ld(qb + ((7u ^ xb) * 8u));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!differ) b = a;
}

struct Rank2 {
  u32 a, c, g, t;      // rows j < p with c1(j) = A, C, G, T
  u32 pa, pc, pg, pt;  // rows j < p with c1(j) = c and c2(j) = A, C, G, T
};
__device__ __forceinline__ Rank2 rank2_from(const Gran2& q, u32 r, u32 c) {  // r = p & 63, c in 1..4
  const u32 m0 = r >= 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
  const u32 m1 = r > 32 ? ((1u << (r - 32)) - 1u) : 0u;
  const bool cb0 = ((c - 1u) & 1u) != 0, cb1 = ((c - 1u) & 2u) != 0;
  Rank2 o;
  o.a = q.s.x; o.c = q.s.y; o.g = q.s.z; o.t = q.s.w;
  o.pa = q.pc.x; o.pc = q.pc.y; o.pg = q.pc.z; o.pt = q.pc.w;
  auto word = [&](u32 y, u32 z, u32 w, u32 y2, u32 z2, u32 w2, u32 m) {
    const u32 A = y & ~z, C = z & ~y, G = y & z, T = w;
    o.a += __popc(A & m); o.c += __popc(C & m); o.g += __popc(G & m); o.t += __popc(T & m);
    const u32 E = (cb1 ? (cb0 ? T : G) : (cb0 ? C : A)) & m;  // rows whose first symbol is c
    o.pa += __popc(y2 & ~z2 & E); o.pc += __popc(z2 & ~y2 & E); o.pg += __popc(y2 & z2 & E); o.pt += __popc(w2 & E);
  };
  word(q.p5.x, q.p5.z, q.p6.x, q.p6.z, q.p7.x, q.p7.z, m0);
  word(q.p5.y, q.p5.w, q.p6.y, q.p6.w, q.p7.y, q.p7.w, m1);
  return o;
}
// constants of a double step, per strand: Cc[c][e] = Occ(e, C[c]); Cd[c] = Occ('$', C[c])
template <bool WIDE>
struct Find2TablesT {
  typedef typename PosOf<WIDE>::type P;
  P Cc[2][4][4];
  P Cd[2][4];
};
typedef Find2TablesT<false> Find2Tables;
template <bool WIDE>
__device__ __forceinline__ void find2_tables_load(Find2TablesT<WIDE>& t, const FmStrand& fwd, const FmStrand& rev) {
  typedef typename PosOf<WIDE>::type P;
  if (threadIdx.x < 8) {
    const u32 which = threadIdx.x >> 2, c = (threadIdx.x & 3u) + 1u;
    const FmStrand& st = which ? rev : fwd;
    const FmRef F = fm_ref(st, which);
    const u64 pos = st.C[c];
    const Cnt4 k = fm_rank<WIDE>(F, pos);
    t.Cc[which][c - 1][0] = (P)k.a; t.Cc[which][c - 1][1] = (P)k.c; t.Cc[which][c - 1][2] = (P)k.g; t.Cc[which][c - 1][3] = (P)k.t;
    t.Cd[which][c - 1] = (P)(pos - (k.a + k.c + k.g + k.t));
  }
}

// Candidate records leave the finder as full 64-byte lines.  A lane that produced a record parks it in its LDS row
// (two 32-byte records per row with u32 positions, one 64-byte record with u64); once the row is full the FOUR lanes
// of the lane's quad (all lanes of a wave stay in the loop until its last chain is done) each take a 16-byte piece of
// that row and write the pieces with one store instruction: one 64-byte request per line instead of one 16-byte
// request per piece (measured at C2: the finder's 160 M scattered piece stores cost 1.8 of its 12.3 ms; the memory
// system counts requests, not bytes).  `tag` = destination byte address | number of 16-byte pieces to write - 1.
template <int NT>
struct FindStageT {  // a half-filled row waits here for the lane's next record, across steps
  static constexpr bool direct = false;
  uint4 row_[NT][4];
  u64 tag_[NT];
  __device__ __forceinline__ uint4* row(u32 t) { return row_[t]; }
  __device__ __forceinline__ u64& tag(u32 t) { return tag_[t]; }
};
typedef FindStageT<256> FindStage;
typedef FindStageT<128> FindStageC;
// No parking: records go to the arena as they are made (two or four 16-byte stores).  For the cooperative finder, whose
// residency is set by its LDS: without the 9 KB of rows a workgroup of 64 reads x 150 bp takes 31 KB, five fit a CU instead
// of four, and at the C3 shape the finder follows its occupancy (tools/coop_occupancy.sh: 35.4 / 27.3 / 24.4 ms at 2 / 3 / 4).
struct FindStageNone {
  static constexpr bool direct = true;
  __device__ __forceinline__ uint4* row(u32) { return nullptr; }
  __device__ __forceinline__ u64& tag(u32) { return dummy; }
  u64 dummy;
};

template <class SG>
__device__ __forceinline__ void find_flush(SG& sg, bool want, u32 tid) {
  if constexpr (SG::direct) return;
  const u32 lane = tid & 63u, q0 = tid & ~3u, piece = tid & 3u;
  __builtin_amdgcn_wave_barrier();
  u64 bal = __ballot(want);
  u32 qm = (u32)(bal >> (lane & 60u)) & 0xFu;  // lanes of my quad with a full row
  while (__ballot(qm != 0) != 0) {             // wave-uniform trip count; quads with nothing left idle through
    if (qm != 0) {
      const u32 src = q0 + (u32)__builtin_ctz(qm);
      const u64 tag = sg.tag(src);
      if (piece <= (u32)(tag & 3u)) {
        const uint4 v = sg.row(src)[piece];
#if (SIGAX_NT & 2)
        {
          typedef u32 nt4 __attribute__((ext_vector_type(4)));
          nt4 w;
          w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
          __builtin_nontemporal_store(w, reinterpret_cast<nt4*>((tag & ~(u64)3) + piece * 16u));
        }
#else
        *reinterpret_cast<uint4*>((tag & ~(u64)3) + piece * 16u) = v;
#endif
      }
      qm &= qm - 1u;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ u32 rd_rank(const unsigned char* rd, u32 i) { return ((u32)rd[i >> 1] >> ((i & 1u) * 4u)) & 15u; }

// STAGED: the workgroup's 64 reads (one contiguous byte range of the batch) are in LDS at `rd`, first byte = the
// 4-byte-aligned address at or below the first read's first base; rd_base = that address's offset in A.seqs.
// COOP: the double step's lines come through LDS (find_step2_dma); NT = threads of the workgroup
template <bool WIDE, bool STAGED, bool TWO, bool COOP = false, int NT = 256, class SG = FindStage>
__device__ __forceinline__ void find_body(const FindArgs& A, FmTables& tb, SG& sg, const Find2TablesT<WIDE>& t2,
                                          const unsigned char* rd, u64 rd_base, uint4* coop_stage = nullptr, u32 tile = blockIdx.x) {
  // A workgroup = 64 reads; wave o of it walks chain o of each, so everything that depends on the chain (which index
  // is primary, complementing, the direction the read is consumed in) is wave-uniform and lives in scalar registers.
  // (With chains_per_wg == 2 a workgroup is 128 reads and two chains: the launch gathers from one strand's tables.)
  const u32 tid = threadIdx.x;
  const u32 wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const u32 o = A.chain_base + (wv & (A.chains_per_wg - 1u));
  // the slot this lane works on, and the read in it: the batch's locality order (FindArgs::perm) or the slot itself
  const u32 wslot = (wv / A.chains_per_wg) * 64u + (tid & 63u);
  const u32 slot = A.read_begin + tile * ((u32)NT / A.chains_per_wg) + wslot;
  const bool inr = slot < A.read_end;
  const u32 read = (A.perm != nullptr && inr) ? A.perm[slot] : slot;
  typedef typename PosOf<WIDE>::type P;
  u32 nocc = 0;
  u32 nsec = 0;  // wave total (scalar): distinct 64-byte sectors of the rank tables asked for (a two-step line is two)
  u32 nb = 0, flagbits = 0;
  bool live = inr && ((A.chain_mask >> o) & 1u);  // overlap: 0xF, or 0x5 without the opposite strand; duplicate: 0x9
  u64 b0 = 0;
  u32 L = 0;
  if (live) {
    b0 = A.offs[read];
    L = (u32)(A.offs[read + 1] - b0);
    live = L > 0;
  }
  const bool pf = o < 2;  // primary index: fmi for chains 0,1; rfmi for 2,3 (overlap_builder.cpp:1120-1132)
  const FmRef F = fm_ref(A.fwd, 0), R = fm_ref(A.rev, 1);
  const FmRef PI = fm_pick(pf, F, R);
  const FmRef OI = fm_pick(pf, R, F);
  const u64* CP = tb.C[PI.which];
  const u64* CO = tb.C[OI.which];
  const void* PI2 = pf ? (const void*)A.fwd.gran2 : (const void*)A.rev.gran2;  // two-step table of the primary index
  const u64* PS2 = pf ? A.fwd.super2 : A.rev.super2;                           // ... its superblock bases (64-bit positions)
  const bool comp = (o & 1u) != 0;            // chains 1 (revcomp) and 3 (complement)
  const bool fromStart = (o == 1 || o == 2);  // reversed strings are consumed from the read's first base
  const u32 af = o == 0 ? SIGAX_AF_CHAIN0 : o == 1 ? SIGAX_AF_CHAIN1 : o == 2 ? SIGAX_AF_CHAIN2 : SIGAX_AF_CHAIN3;
  const unsigned char* sq = A.seqs + b0;
  // this read's first base in the staged copy (gathered by slot when the batch has a locality order)
  const u32 rdo = A.perm != nullptr ? wslot * A.stage_stride : (u32)(b0 - rd_base);
  // wave-uniform base of the wave's reads (lane 0's read always exists) + this lane's distance from it
  const u64 b0w = ((u64)__builtin_amdgcn_readfirstlane((u32)(b0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)b0);
  const unsigned char* sq0 = A.seqs + b0w;
  const u32 sqd = (u32)(b0 - b0w);
  // cap is even (sigax_api.cpp) so every chain's slots start on a 64-byte line
  // byte address of this chain's slot i (computed where a record is parked: cheaper than two live registers)
  auto slot_addr = [&](u32 i) -> u64 {
    return reinterpret_cast<u64>(A.arena) + (((u64)read * 4 + o) * A.cap + i) * sizeof(Cand<WIDE>);
  };

  P lo0 = 0, sz = 0, lo1 = 0;
  u32 s = 1;                  // this chain's current match length
  bool started = false;
  if (STAGED && A.deep_k != 0u) {
    // the chain's first K = deep_k <= min-overlap symbols from the deep start table of its primary index (fm_layout.h): the
    // state after them in one lookup, when they are all ACGT and the K-mer occurs in the indexed reads at all
    const u32 K = A.deep_k;
    const bool can = live && L >= K;
    if (__ballot(can) != 0) {
      u64 k0 = 0, k1 = 0;
      bool acgt = can;
      for (u32 i = 0; i < K; ++i) {
        u32 r = 1u;
        if (can) {
          const u32 at = rdo + (fromStart ? i : L - 1u - i);
          r = COOP ? rd_rank(rd, at) : base_rank(rd[at]);
        }
        if (comp) r = comp_rank(r);
        acgt = acgt && r != 0u;
        const u64 c = (u64)((r - 1u) & 3u);
        if (i < 32u) k0 |= c << (2u * i);
        else k1 |= c << (2u * (i - 32u));
      }
      if (acgt) {
        k1 |= (u64)(0x8000u | K) << 48;
        P a0 = 0, a1 = 0, az = 0;
        if (deep_lookup<WIDE>(pf ? A.fwd.deep : A.rev.deep, pf ? A.fwd.deep_slots : A.rev.deep_slots, k0, k1, a0, a1, az)) {
          lo0 = a0; lo1 = a1; sz = az; s = K;
          started = true;
        }
      }
    }
  }
  if (STAGED && live && !started && A.start_ok && L >= (u32)SIGAX_START_K) {
    // the chain's first twelve symbols from the start table of its primary index (fm_layout.h), when they are all ACGT
    const void* ST = pf ? A.fwd.start : A.rev.start;
    u32 code = 0;
    bool acgt = true;
#pragma unroll
    for (u32 i = 0; i < (u32)SIGAX_START_K; ++i) {
      const u32 at = rdo + (fromStart ? i : L - 1u - i);
      u32 r = COOP ? rd_rank(rd, at) : base_rank(rd[at]);
      if (comp) r = comp_rank(r);
      acgt = acgt && r != 0u;
      code = (code << 2) | ((r - 1u) & 3u);
    }
    if (acgt) {
      if (WIDE) {
        const ulonglong2* q = reinterpret_cast<const ulonglong2*>(ST) + 2ull * code;
        const ulonglong2 a = q[0], b = q[1];
        lo0 = (P)a.x; lo1 = (P)a.y; sz = (P)b.x; s = (u32)b.y;
      } else {
        const uint4 e = reinterpret_cast<const uint4*>(ST)[code];
        lo0 = (P)e.x; lo1 = (P)e.y; sz = (P)e.z; s = e.w;
      }
      started = true;
    }
  }
  nsec += (u32)__popcll(__ballot(started));
  if (live && !started) {
    u32 r = (STAGED && COOP) ? rd_rank(rd, rdo + (fromStart ? 0 : L - 1)) : base_rank(STAGED ? rd[rdo + (fromStart ? 0 : L - 1)] : sq[fromStart ? 0 : L - 1]);
    if (comp) r = comp_rank(r);
    // IntervalPair::init (overlap_builder.cpp:91-94, fmindex.h:90-93)
    lo0 = (P)CP[r]; sz = (P)tb.T[PI.which][r]; lo1 = (P)CO[r];
  }
  // All 64 lanes stay in the loop until the wave's last chain is done (a finished chain idles, predicated off):
  // the quad-cooperative record stores need every lane of a quad present.
  bool full = !live;          // the arena slots of this chain ran out ("cannot happen": cap comes from the longest read)
  // probe = ranges; probe.updateL('$') (overlap_builder.cpp:861-865) is valid <=> d > 0: park the candidate block in
  // this lane's row; returns whether the row is now full (to be written by the quad)
  auto emit = [&](P c0lo, P d, P c1lo, P r0lo, P szv, u32 len) -> bool {
    bool fl = false;
    if constexpr (SG::direct) {
      if (nb < A.cap - 1 && !full) {
        cand_store(reinterpret_cast<Cand<WIDE>*>(slot_addr(nb)), c0lo, d, c1lo, r0lo, c1lo, szv, len, af);
      } else {
        flagbits |= 1u;
        full = true;
      }
      ++nb;
      return false;
    }
    if (nb < A.cap - 1 && !full) {
      if (WIDE) {
        cand_store(reinterpret_cast<Cand<WIDE>*>(sg.row(tid)), c0lo, d, c1lo, r0lo, c1lo, szv, len, af);
        sg.tag(tid) = slot_addr(nb) | 3u;
        fl = true;
      } else {
        cand_store(reinterpret_cast<Cand<WIDE>*>(sg.row(tid) + (nb & 1u) * 2), c0lo, d, c1lo, r0lo, c1lo, szv, len, af);
        if (nb & 1u) {
          sg.tag(tid) = slot_addr(nb - 1) | 3u;
          fl = true;
        }
      }
    } else {
      flagbits |= 1u;
      full = true;
    }
    ++nb;
    return fl;
  };
  for (;;) {
    const bool on = live && sz != 0 && s < L;  // an empty range stays empty, nothing more can be emitted
    if (__ballot(on) == 0) break;
    if (TWO) {
      // Double step: when every running chain of the wave has two more ACGT bases to consume, both backward steps
      // are taken from ONE pair of positions of the two-step table (fm_layout.h): half the memory lines per base.
      u32 c = 0, e = 0;
      const bool want2 = on && s + 1 < L;
      if (want2) {
        if (COOP) {  // staged as ranks
          c = rd_rank(rd, rdo + (fromStart ? s : L - 1 - s));
          e = rd_rank(rd, rdo + (fromStart ? s + 1 : L - 2 - s));
        } else {
          c = base_rank(rd[rdo + (fromStart ? s : L - 1 - s)]);
          e = base_rank(rd[rdo + (fromStart ? s + 1 : L - 2 - s)]);
        }
        if (comp) { c = comp_rank(c); e = comp_rank(e); }
      }
      const bool ok2 = want2 && c != 0 && e != 0;
      if (__ballot(on && !ok2) == 0) {
        bool fl1 = false, fl2 = false, two_lines = false;
        P lo1n = 0, lo0n = 0, szn = 0, ldn = 0, dd2 = 0, nlo1 = 0, nlo0 = 0, nsz = 0;
        Gran2 ga, gb;
        P pl = 0, pu = 0;
        if (on) {
          const P nn = (P)PI.n;  // never leave the table, whatever an invalid interval holds
          pl = lo0 > nn ? nn : lo0;
          const P up = (P)(lo0 + sz);
          pu = (up > nn || up < lo0) ? nn : up;
          two_lines = (pl >> 6) != (pu >> 6);
        }
        const u64 two_mask = COOP ? 0ull : __ballot(two_lines);
        if (COOP) {
          // all 64 lanes take part in the loads (an idle chain asks for line 0)
          find_step2_dma(reinterpret_cast<const uint32_t*>(PI2), (u32)(pl >> 6), (u32)(pu >> 6), on ? c : 1u,
                         coop_stage + wv * COOP_WAVE_U4, tid & 63u, ga, gb);
        }
        if (on) {
          if (!COOP) {
            const u32 oa = (u32)(pl >> 6) * 128u, ob = (u32)(pu >> 6) * 128u;
            if (A.mask_upper) {
              find_step2_loads_masked(PI2, oa, oa + 16u * c, ob, ob + 16u * c, two_mask, ga, gb);
              if (!two_lines) gb = ga;
            } else {
              find_step2_loads(PI2, oa, oa + 16u * c, ob, ob + 16u * c, ga, gb);
            }
          }
          const Rank2 lr = rank2_from(ga, (u32)pl & 63u, c), ur = rank2_from(gb, (u32)pu & 63u, c);
          // counts in the index's position type; with 64-bit positions the line's counters are relative to its superblock
          P la = lr.a, lcn = lr.c, lg = lr.g, lt = lr.t, ua = ur.a, ucn = ur.c, ug = ur.g, ut = ur.t;
          P lpa = lr.pa, lpc = lr.pc, lpg = lr.pg, lpt = lr.pt, upa = ur.pa, upc = ur.pc, upg = ur.pg, upt = ur.pt;
          if (WIDE) {
            const u64* sl = PS2 + ((u64)pl >> SIGAX_SUPER_SHIFT) * 20;
            const u64* su = PS2 + ((u64)pu >> SIGAX_SUPER_SHIFT) * 20;
            const u32 pb = 4u + (c - 1u) * 4u;
            la += (P)sl[0]; lcn += (P)sl[1]; lg += (P)sl[2]; lt += (P)sl[3];
            ua += (P)su[0]; ucn += (P)su[1]; ug += (P)su[2]; ut += (P)su[3];
            lpa += (P)sl[pb]; lpc += (P)sl[pb + 1]; lpg += (P)sl[pb + 2]; lpt += (P)sl[pb + 3];
            upa += (P)su[pb]; upc += (P)su[pb + 1]; upg += (P)su[pb + 2]; upt += (P)su[pb + 3];
          }
          const P da = ua - la, dc = ucn - lcn, dg = ug - lg, dt = ut - lt;
          const P dd = sz - (da + dc + dg + dt);  // '$' extensions of the current string
          const P ld = lo0 - (la + lcn + lg + lt);
          if (s >= A.minov && dd > 0) fl1 = emit(ld, dd, lo1, lo0, sz, s);
          // first step, symbol c (overlap_builder.cpp:112-122)
          const P acc1 = sel5<P>(c, (P)0, dd, dd + da, dd + da + dc, dd + da + dc + dg);
          const P lc = sel5<P>(c, ld, la, lcn, lg, lt);
          const P dcc = sel5<P>(c, dd, da, dc, dg, dt);
          lo1n = lo1 + acc1;
          lo0n = (P)CP[c] + lc;
          szn = dcc;
          // second step, symbol e, from the pair counts: Occ(e, lo0n) = Cc[c][e] + R2(e, c, lower), same for upper
          const P d2a = upa - lpa, d2c = upc - lpc, d2g = upg - lpg, d2t = upt - lpt;
          const P l2d = lc - (lpa + lpc + lpg + lpt);  // rows below with first symbol c and '$' before it
          dd2 = dcc - (d2a + d2c + d2g + d2t);
          ldn = t2.Cd[PI.which][c - 1] + l2d;               // Occ('$', lo0n)
          const P acc2 = sel5<P>(e, (P)0, dd2, dd2 + d2a, dd2 + d2a + d2c, dd2 + d2a + d2c + d2g);
          const P l2e = sel5<P>(e, (P)0, lpa, lpc, lpg, lpt);
          const P d2e = sel5<P>(e, (P)0, d2a, d2c, d2g, d2t);
          nlo1 = lo1n + acc2;
          nlo0 = (P)CP[e] + t2.Cc[PI.which][c - 1][e - 1] + l2e;
          nsz = d2e;
        }
        nsec += 2u * (u32)(__popcll(__ballot(on)) + __popcll(__ballot(two_lines)));
        find_flush(sg, fl1, tid);
        if (on) {
          if (s + 1 >= A.minov && dd2 > 0) fl2 = emit(ldn, dd2, lo1n, lo0n, szn, s + 1);
          lo1 = nlo1; lo0 = nlo0; sz = nsz;
          s += szn != 0 ? 2u : 1u;  // an empty range after the first step: the second is not taken (and sz is 0)
        }
        find_flush(sg, fl2, tid);
        continue;
      }
    }
    bool flush = false, two_gran = false;
    if (on) {
      const u64 pl = (u64)lo0 > PI.n ? PI.n : (u64)lo0;  // never leave the table, whatever an invalid interval holds
      const u64 pu0 = (u64)(P)(lo0 + sz);
      const u64 pu = pu0 > PI.n ? PI.n : pu0;
      Gran gl, gu;
      u32 ch;
      two_gran = (pl >> 7) != (pu >> 7);
      if (STAGED) {
        ch = COOP ? rd_rank(rd, rdo + (fromStart ? s : L - 1 - s)) : (u32)rd[rdo + (fromStart ? s : L - 1 - s)];
        if (WIDE) find_step_loads8(PI.g + (pl >> 7) * 4, PI.g + (pu >> 7) * 4, gl, gu);
        else find_step_loads8_s(PI.g, (u32)(pl >> 7) * 64u, (u32)(pu >> 7) * 64u, gl, gu);
      } else if (WIDE) {
        find_step_loads(PI.g + (pl >> 7) * 4, PI.g + (pu >> 7) * 4, sq + (fromStart ? s : L - 1 - s), gl, gu, ch);
      } else {
        find_step_loads_s(PI.g, (u32)(pl >> 7) * 64u, (u32)(pu >> 7) * 64u, sq0, sqd + (fromStart ? s : L - 1 - s), gl, gu, ch);
      }
      const Cnt4P<P> l = fm_rank4p_from<WIDE>(PI, pl, gl);
      const Cnt4P<P> u = fm_rank4p_from<WIDE>(PI, pu, gu);
      u32 r = (STAGED && COOP) ? ch : base_rank(ch & 0xFFu);
      if (comp) r = comp_rank(r);
      P da = u.a - l.a, dc = u.c - l.c, dg = u.g - l.g, dt = u.t - l.t;
      P dd = sz - (da + dc + dg + dt);  // '$' extensions
      const P ld = lo0 - (l.a + l.c + l.g + l.t);
      if (s >= A.minov && dd > 0) flush = emit(ld, dd, lo1, lo0, sz, s);
      // ranges.updateL(c) (overlap_builder.cpp:112-122), branch-free: acc = extensions by smaller symbols
      const P p1 = dd, p2 = dd + da, p3 = p2 + dc, p4 = p3 + dg;
      const P acc = sel5<P>(r, (P)0, p1, p2, p3, p4);
      const P lc = sel5<P>(r, ld, l.a, l.c, l.g, l.t);
      const P dcur = sel5<P>(r, dd, da, dc, dg, dt);
      lo1 += acc;
      lo0 = (P)CP[r] + lc;
      sz = dcur;
      ++s;
    }
    nsec += (u32)(__popcll(__ballot(on)) + __popcll(__ballot(two_gran)));
    find_flush(sg, flush, tid);
  }
  // a single record left in the row (u32 positions, odd count): 32 bytes = two pieces
  {
    const bool tail = !WIDE && live && !full && (nb & 1u);
    if (!SG::direct && tail) sg.tag(tid) = slot_addr(nb - 1) | 1u;
    find_flush(sg, tail, tid);
  }
  bool contain = false;
  u32 tail_sec = 0;
  nocc = live ? 2u * (s - 1u) : 0u;  // two rank positions per step taken
  if (live && sz != 0 && s >= L) {
    // full-length interval: substring test and containment block (overlap_builder.cpp:889-904)
    // two granules at a time (the step loader again): four at once would set the kernel's register count
    auto clampn = [](u64 p, u64 n) { return p > n ? n : p; };
    const u64 pl = clampn((u64)lo0, PI.n), pu = clampn((u64)(P)(lo0 + sz), PI.n);
    const u64 ql = clampn((u64)lo1, OI.n), qu = clampn((u64)(P)(lo1 + sz), OI.n);
    Gran ga, gb;
    u32 ignored;
    tail_sec = ((pl >> 7) != (pu >> 7) ? 2u : 1u) + ((ql >> 7) != (qu >> 7) ? 2u : 1u);
    find_step_loads(PI.g + (pl >> 7) * 4, PI.g + (pu >> 7) * 4, sq, ga, gb, ignored);
    const Cnt4P<P> l = fm_rank4p_from<WIDE>(PI, pl, ga);
    const Cnt4P<P> u = fm_rank4p_from<WIDE>(PI, pu, gb);
    find_step_loads(OI.g + (ql >> 7) * 4, OI.g + (qu >> 7) * 4, sq, ga, gb, ignored);
    const Cnt4P<P> lp = fm_rank4p_from<WIDE>(OI, ql, ga);
    const Cnt4P<P> up = fm_rank4p_from<WIDE>(OI, qu, gb);
    nocc += 4;
    bool dna = (u.a - l.a) | (u.c - l.c) | (u.g - l.g) | (u.t - l.t) | (up.a - lp.a) | (up.c - lp.c) | (up.g - lp.g) |
               (up.t - lp.t);
    if (dna) {
      flagbits |= SIGAX_CC_SUBSTRING;
    } else {
      // no DNA extension on either side: all sz extensions are '$', so probe.updateL('$') keeps the whole
      // range and probe.updateR('$') reuses the two positions of rext.
      P ld = lo0 - (l.a + l.c + l.g + l.t);
      P lpd = lo1 - (lp.a + lp.c + lp.g + lp.t);
      if constexpr (SG::direct) {
        cand_store(reinterpret_cast<Cand<WIDE>*>(slot_addr(A.cap - 1)), ld, sz, lpd, lo0, lo1, sz, L, af);
      } else {
        cand_store(reinterpret_cast<Cand<WIDE>*>(sg.row(tid)), ld, sz, lpd, lo0, lo1, sz, L, af);
        sg.tag(tid) = slot_addr(A.cap - 1) | (WIDE ? 3u : 1u);
      }
      flagbits |= SIGAX_CC_CONTAIN;
      contain = true;
    }
  }
  find_flush(sg, contain, tid);
  if (inr) {
    // a chain that ran out of slots (the host repeats the run with more) has records parked and never written: its
    // readers get an empty chain
    const u32 nbw = (flagbits & 1u) ? 0u : nb;
    u32 word = (nbw & SIGAX_CC_COUNT_MASK) | (flagbits & (SIGAX_CC_SUBSTRING | SIGAX_CC_CONTAIN));
    A.chain_cnt[(u64)read * 4 + o] = word;
  }
  u64 tot_occ = wave_sum((u64)nocc);
  u64 tot_blk = wave_sum((u64)nb + ((flagbits & SIGAX_CC_CONTAIN) ? 1u : 0u));
  u64 tot_err = wave_sum((u64)(flagbits & 1u));
  u64 tot_sec = (u64)nsec + wave_sum((u64)tail_sec);
  u32 mx = nb;  // the longest chain of the launch: what the host sizes the next run's slots by (sigax_api.cpp)
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (u32)__shfl_xor((int)mx, off, 64));
  if ((threadIdx.x & 63) == 0) {
    if (mx > A.max_seen) atomicMax(&A.dstat[DS_MAX_CHAIN], (u64)mx);
    if (tot_sec) atomicAdd(&A.dstat[DS_SEC_FIND], tot_sec);
    if (tot_occ) atomicAdd(&A.dstat[DS_OCC_FIND], tot_occ);
    if (tot_blk) atomicAdd(&A.dstat[DS_CAND_BLOCKS], tot_blk);
    if (tot_err) atomicAdd(&A.dstat[DS_FIND_OVERFLOW], tot_err);
  }
}

// Copies the workgroup's reads into the dynamic LDS when they fit (whole aligned words: the words holding the first and
// the last base belong to the same allocation as the bases); returns whether it did.
extern __shared__ __attribute__((aligned(16))) unsigned char find_dyn_lds[];
template <int NT = 256>
__device__ __forceinline__ bool find_stage_reads(const FindArgs& A, u64* rd_base) {
  const u32 per = (u32)NT / A.chains_per_wg;
  const u32 r0 = A.read_begin + blockIdx.x * per;
  const u32 r1 = r0 + per < A.read_end ? r0 + per : A.read_end;
  const u64 lo = A.offs[r0], hi = A.offs[r1];
  const u64 alo = (reinterpret_cast<u64>(A.seqs) + lo) & ~3ull;
  const u64 nbytes = reinterpret_cast<u64>(A.seqs) + hi - alo;
  *rd_base = alo - reinterpret_cast<u64>(A.seqs);
  if (nbytes + 4 > (u64)A.stage_bytes) return false;
  const u32* src = reinterpret_cast<const u32*>(alo);
  u32* dst = reinterpret_cast<u32*>(find_dyn_lds);
  const u32 nw = (u32)((nbytes + 3) >> 2);
  for (u32 w = threadIdx.x; w < nw; w += (u32)NT) dst[w] = src[w];
  return true;
}

// With a locality order the workgroup's reads are not one byte range: read j of the workgroup goes to dst + j * stride
// (a multiple of 4), copied by thread j in (unaligned) 4-byte words -- one dependent chain of perm -> offset -> bases per
// thread, all reads of the workgroup at once.
template <int NT>
__device__ __forceinline__ bool find_stage_reads_perm(const FindArgs& A, u32 tile) {
  const u32 per = (u32)NT / A.chains_per_wg;
  if ((u64)per * A.stage_stride > (u64)A.stage_bytes) return false;
  const u32 r0 = A.read_begin + tile * per;
  const u32 j = threadIdx.x;
  if (j < per && r0 + j < A.read_end) {
    const u32 r = A.perm[r0 + j];
    const u64 b = A.offs[r];
    const u32 len = (u32)(A.offs[r + 1] - b);
    const unsigned char* src = A.seqs + b;
    unsigned char* d = find_dyn_lds + j * A.stage_stride;
    const u32 nw = len >> 2;
    for (u32 k = 0; k < nw; ++k) {
      u32 w;
      __builtin_memcpy(&w, src + 4u * k, 4);
      reinterpret_cast<u32*>(d)[k] = w;
    }
    for (u32 i = nw << 2; i < len; ++i) d[i] = src[i];
  }
  return true;
}

// The cooperative finder stages its reads as 4-bit ranks, two per byte (its residency hangs on its LDS, and the rank is
// what a step needs anyway): byte i of the staged range = nibble i.
template <int NT>
__device__ __forceinline__ bool find_stage_reads_packed(const FindArgs& A, u64* rd_base, u32 tile) {
  const u32 per = (u32)NT / A.chains_per_wg;
  const u32 r0 = A.read_begin + tile * per;
  const u32 r1 = r0 + per < A.read_end ? r0 + per : A.read_end;
  const u64 lo = A.offs[r0], hi = A.offs[r1];
  const u64 alo = (reinterpret_cast<u64>(A.seqs) + lo) & ~3ull;
  const u64 nbytes = reinterpret_cast<u64>(A.seqs) + hi - alo;
  *rd_base = alo - reinterpret_cast<u64>(A.seqs);
  if ((nbytes + 3) / 2 + 4 > (u64)A.stage_bytes) return false;
  const u32* src = reinterpret_cast<const u32*>(alo);
  unsigned short* dst = reinterpret_cast<unsigned short*>(find_dyn_lds);
  const u32 nw = (u32)((nbytes + 3) >> 2);
  for (u32 w = threadIdx.x; w < nw; w += (u32)NT) {
    const u32 v = src[w];
    dst[w] = (unsigned short)(base_rank(v & 0xFFu) | (base_rank((v >> 8) & 0xFFu) << 4) | (base_rank((v >> 16) & 0xFFu) << 8) |
                              (base_rank(v >> 24) << 12));
  }
  return true;
}

// ... and with a locality order, gathered by slot like find_stage_reads_perm: read j's ranks start at nibble j * stride
template <int NT>
__device__ __forceinline__ bool find_stage_reads_packed_perm(const FindArgs& A, u32 tile) {
  const u32 per = (u32)NT / A.chains_per_wg;
  if ((u64)per * A.stage_stride / 2u > (u64)A.stage_bytes) return false;
  const u32 r0 = A.read_begin + tile * per;
  const u32 j = threadIdx.x;
  if (j < per && r0 + j < A.read_end) {
    const u32 r = A.perm[r0 + j];
    const u64 b = A.offs[r];
    const u32 len = (u32)(A.offs[r + 1] - b);
    const unsigned char* src = A.seqs + b;
    unsigned char* d = find_dyn_lds + j * (A.stage_stride / 2u);
    const u32 nw = len >> 2;
    for (u32 k = 0; k < nw; ++k) {
      u32 v;
      __builtin_memcpy(&v, src + 4u * k, 4);
      reinterpret_cast<unsigned short*>(d)[k] = (unsigned short)(base_rank(v & 0xFFu) | (base_rank((v >> 8) & 0xFFu) << 4) |
                                                                 (base_rank((v >> 16) & 0xFFu) << 8) | (base_rank(v >> 24) << 12));
    }
    for (u32 i = nw << 2; i < len; ++i) {
      const u32 rk = base_rank(src[i]);
      const u32 old = (i & 1u) ? d[i >> 1] : 0u;
      d[i >> 1] = (unsigned char)((i & 1u) ? ((old & 0x0Fu) | (rk << 4)) : rk);
    }
  }
  return true;
}

// u32 positions: held to 64 registers, so that two finder workgroups and three filter/extract waves per SIMD fit the
// 512-entry register file together (2 x 64 + 3 x 128)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_find_n(FindArgs A) {
  __shared__ FmTables tb;
  __shared__ FindStage sg;
  __shared__ Find2Tables t2;  // unused here
  u64 rd_base = 0;
  const bool staged = A.perm != nullptr ? find_stage_reads_perm<256>(A, blockIdx.x) : find_stage_reads(A, &rd_base);
  fm_tables_load(tb, A.fwd, A.rev);  // ends with the workgroup barrier that also publishes the staged reads
  if (staged) find_body<false, true, false>(A, tb, sg, t2, find_dyn_lds, rd_base);
  else find_body<false, false, false>(A, tb, sg, t2, find_dyn_lds, rd_base);
}
// u32 positions with the two-step table (index below 2^31 symbols, reads staged in LDS)
__global__ __launch_bounds__(256) void k_find_n2(FindArgs A) {
  __shared__ FmTables tb;
  __shared__ FindStage sg;
  __shared__ Find2Tables t2;
  u64 rd_base = 0;
  const bool staged = A.perm != nullptr ? find_stage_reads_perm<256>(A, blockIdx.x) : find_stage_reads(A, &rd_base);
  find2_tables_load(t2, A.fwd, A.rev);
  fm_tables_load(tb, A.fwd, A.rev);
#ifdef SIGAX_FIND_PRIO
  __builtin_amdgcn_s_setprio(SIGAX_FIND_PRIO);  // A/B: issue priority over the filter/extract waves on the same SIMD
#endif
  if (staged) find_body<false, true, true>(A, tb, sg, t2, find_dyn_lds, rd_base);
  else find_body<false, false, false>(A, tb, sg, t2, find_dyn_lds, rd_base);
}
// u32 positions, two-step table of any size below 2^32 symbols, lines fetched cooperatively through LDS: a workgroup =
// 2 waves = the two chains of one strand for 64 reads (launched once per strand)
// A workgroup walks tiles of 64 reads: tile = blockIdx.x, + gridDim.x, ...; normally the grid has one workgroup per tile
// (FindArgs::coop_grid caps it for measurements).
__global__ __launch_bounds__(128) void k_find_c2(FindArgs A) {
  __shared__ FmTables tb;
  __shared__ Find2Tables t2;
  __shared__ __attribute__((aligned(16))) uint4 stage[2 * COOP_WAVE_U4];
  find2_tables_load(t2, A.fwd, A.rev);
  fm_tables_load(tb, A.fwd, A.rev);
  const u32 ntiles = (A.read_end - A.read_begin + 63u) / 64u;
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    u64 rd_base = 0;
    const bool staged = A.perm != nullptr ? find_stage_reads_packed_perm<128>(A, tile) : find_stage_reads_packed<128>(A, &rd_base, tile);
    __syncthreads();
    FindStageNone sg;
    if (staged) find_body<false, true, true, true, 128, FindStageNone>(A, tb, sg, t2, find_dyn_lds, rd_base, stage, tile);
    else find_body<false, false, false, false, 128, FindStageNone>(A, tb, sg, t2, find_dyn_lds, rd_base, stage, tile);
    __syncthreads();  // the staged reads are overwritten by the next tile's
  }
}
// the same with 64-bit positions (indexes of 2^32 symbols and more: BASELINE configs[4])
__global__ __launch_bounds__(128) void k_find_c2w(FindArgs A) {
  __shared__ FmTables tb;
  __shared__ Find2TablesT<true> t2;
  __shared__ __attribute__((aligned(16))) uint4 stage[2 * COOP_WAVE_U4];
  find2_tables_load<true>(t2, A.fwd, A.rev);
  fm_tables_load(tb, A.fwd, A.rev);
  const u32 ntiles = (A.read_end - A.read_begin + 63u) / 64u;
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    u64 rd_base = 0;
    const bool staged = A.perm != nullptr ? find_stage_reads_packed_perm<128>(A, tile) : find_stage_reads_packed<128>(A, &rd_base, tile);
    __syncthreads();
    FindStageNone sg;
    if (staged) find_body<true, true, true, true, 128, FindStageNone>(A, tb, sg, t2, find_dyn_lds, rd_base, stage, tile);
    else find_body<true, false, false, false, 128, FindStageNone>(A, tb, sg, t2, find_dyn_lds, rd_base, stage, tile);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_find_w(FindArgs A) {
  __shared__ FmTables tb;
  __shared__ FindStage sg;
  __shared__ Find2TablesT<true> t2;  // unused here
  u64 rd_base = 0;
  const bool staged = A.perm != nullptr ? find_stage_reads_perm<256>(A, blockIdx.x) : find_stage_reads(A, &rd_base);
  fm_tables_load(tb, A.fwd, A.rev);
  if (staged) find_body<true, true, false>(A, tb, sg, t2, find_dyn_lds, rd_base);
  else find_body<true, false, false>(A, tb, sg, t2, find_dyn_lds, rd_base);
}

// -------------------------------------------------------------------------------------------------------
// k_filter_extract (general form): one lane per read, literal list emulation in a per-lane pool.
// -------------------------------------------------------------------------------------------------------
#define FX_COUNTDOWN 0x80000000u  // E::len / group count of a lone single-row block that waits for the end of its read (GFx::round)
struct Ent {  // a block whose capped pair may have been rewritten; raw/length/af live in the candidate arena
  u64 c0lo, c0hi, c1lo, c1hi;
  u32 src, len;
  u32 pad0, pad1;
};

__device__ __forceinline__ bool iv_valid(u64 lo, u64 hi) { return hi != ~0ull && hi >= lo; }  // fmindex.h:87-89
__device__ __forceinline__ bool intersecting(u64 s1, u64 e1, u64 s2, u64 e2) { return !(s1 > e2 || s2 > e1); }  // coord.h:37-40

template <bool WIDE>
struct Fx {
  const FxArgs& A;
  const FmTables& tb;
  FmRef F, R;     // forward / reverse strand
  Ent* pool;
  u32 top;        // bump pointer (entries)
  bool overflow;  // pool exhausted
  bool xerror;    // extract() returned false
  u64 nocc;
  u32 read, nout;
  const Cand<WIDE>* slots;  // this read's 4*cap candidate slots

  __device__ Fx(const FxArgs& a, const FmTables& t, Ent* p)
      : A(a), tb(t), F(fm_ref(a.fwd, 0)), R(fm_ref(a.rev, 1)), pool(p), top(0), overflow(false), xerror(false), nocc(0),
        read(0), nout(0), slots(nullptr) {}

  __device__ u32 af_of(u32 src) const {
    u32 ch = src / A.cap;
    return ch == 0 ? SIGAX_AF_CHAIN0 : ch == 1 ? SIGAX_AF_CHAIN1 : ch == 2 ? SIGAX_AF_CHAIN2 : SIGAX_AF_CHAIN3;
  }
  // OverlapBlock::index (overlap_builder.cpp:177-179): !TARGETREV ? rfmi : fmi.  Chains 0,1 are !TARGETREV.
  __device__ FmRef ext_index(u32 src) const { return fm_pick((src / A.cap) < 2, R, F); }

  __device__ bool need(u32 upto) {
    if (upto + nout > A.pool_cap) { overflow = true; return false; }
    return true;
  }

  __device__ void load_ent(Ent& e, u32 src) const {
    const Cand<WIDE> b = cand_load<WIDE>(slots + src);
    e.c0lo = b.c0lo; e.c0hi = (u64)b.c0lo + b.d - 1; e.c1lo = b.c1lo; e.c1hi = (u64)b.c1lo + b.d - 1;
    e.src = src; e.len = b.len; e.pad0 = e.pad1 = 0;
  }

  // outblocks->push_back: emitted entries wait at the tail of the pool (growing downwards) until the read is done
  __device__ void emit(const Ent& e) {
    if (top + nout + 1 > A.pool_cap) { overflow = true; return; }
    pool[A.pool_cap - 1 - nout] = e;
    ++nout;
  }
  // one allocation per read out of this lane's chunk of the unordered arena
  __device__ void flush(u64& fin_cur, u64& fin_end) {
    if (fin_cur + nout > fin_end) {
      u64 grab = nout > 64 ? nout : 64;
      fin_cur = atomicAdd(&A.dstat[DS_FIN_TOP], grab);
      fin_end = fin_cur + grab;
    }
    A.item_base[2ull * read] = fin_cur;
    for (u32 i = 0; i < nout; ++i) {
      u64 slot = fin_cur + i;
      if (slot >= A.fin_cap) break;
      const Ent& e = pool[A.pool_cap - 1 - i];
      const Cand<WIDE> b = cand_load<WIDE>(slots + e.src);
      store_block(A.fin + slot, e.c0lo, e.c0hi, e.c1lo, e.c1hi, b.r0lo, (u64)b.r0lo + b.sz - 1, b.r1lo, (u64)b.r1lo + b.sz - 1, b.len, b.af);
    }
    fin_cur += nout;
  }

  // IntervalPair::updateR(c, index) on the capped pair (overlap_builder.cpp:101-106,123-133)
  __device__ void updateR(Ent& e, u32 r, const FmRef& ix) {
    u64 l[5], u[5];
    fm_rank5<WIDE>(ix, e.c1lo, l);
    fm_rank5<WIDE>(ix, e.c1hi + 1, u);
    u64 acc = 0, lr = 0, ur = 0;
    for (u32 b = 0; b < 5; ++b) {
      if (b < r) acc += u[b] - l[b];
      if (b == r) { lr = l[b]; ur = u[b]; }
    }
    u64 pb = tb.C[ix.which][r];
    e.c0lo += acc;
    e.c0hi = e.c0lo + (ur - lr) - 1;
    e.c1lo = pb + lr;
    e.c1hi = pb + ur - 1;
  }
  // IntervalPair::updateL(c, index) (overlap_builder.cpp:95-100,112-122)
  __device__ void updateL(Ent& e, u32 r, const FmRef& ix) {
    u64 l[5], u[5];
    fm_rank5<WIDE>(ix, e.c0lo, l);
    fm_rank5<WIDE>(ix, e.c0hi + 1, u);
    u64 acc = 0, lr = 0, ur = 0;
    for (u32 b = 0; b < 5; ++b) {
      if (b < r) acc += u[b] - l[b];
      if (b == r) { lr = l[b]; ur = u[b]; }
    }
    u64 pb = tb.C[ix.which][r];
    e.c1lo += acc;
    e.c1hi = e.c1lo + (ur - lr) - 1;
    e.c0lo = pb + lr;
    e.c0hi = pb + ur - 1;
  }
  // OverlapBlock::ext (overlap_builder.cpp:181-187)
  __device__ void ext(const Ent& e, u64 x[5]) {
    FmRef ix = ext_index(e.src);
    u64 l[5], u[5];
    fm_rank5<WIDE>(ix, e.c1lo, l);
    fm_rank5<WIDE>(ix, e.c1hi + 1, u);
    for (int k = 0; k < 5; ++k) x[k] = u[k] - l[k];
    if (af_of(e.src) & 4u) {  // QUERYCOMP: AlphaCount::complement (alphabet.h:68-71)
      u64 t = x[1]; x[1] = x[4]; x[4] = t;
      t = x[2]; x[2] = x[3]; x[3] = t;
    }
  }

  // stable insertion sorts (std::list::sort is a stable merge sort)
  __device__ void sort_left(u32 base, u32 n) {
    for (u32 i = 1; i < n; ++i) {
      Ent k = pool[base + i];
      u32 j = i;
      while (j > 0 && k.c0lo < pool[base + j - 1].c0lo) { pool[base + j] = pool[base + j - 1]; --j; }
      pool[base + j] = k;
    }
  }
  __device__ void sort_len_desc(u32 base, u32 n) {
    for (u32 i = 1; i < n; ++i) {
      Ent k = pool[base + i];
      u32 j = i;
      while (j > 0 && k.len > pool[base + j - 1].len) { pool[base + j] = pool[base + j - 1]; --j; }
      pool[base + j] = k;
    }
  }

  // SubMaximalBlockFilter::resolve (overlap_builder.cpp:965-1082).  Writes the resolved list to pool[rb..) and
  // returns its size.  Pfm/Ofm are the filter's (_fmi,_rfmi): swapped for the reverse lists (:1147-1151).
  __device__ u32 resolve(const Ent& x, const Ent& y, u32 rb, u32 rcap, const FmRef& Pfm, const FmRef& Ofm) {
    const Ent* higher = &x;
    const Ent* lower = &y;
    if (higher->len < lower->len) { const Ent* t = higher; higher = lower; lower = t; }
    u32 k = 0;
    pool[rb + k++] = *higher;
    if (higher->len == lower->len) return k;  // equal lengths: same coordinates (else the reference only logs)
    if (!(lower->c0lo < higher->c0lo || lower->c0hi > higher->c0hi)) return k;
    // re-map every reverse position of the lower block to its forward position by walking the BWT
    const Cand<WIDE> lb = cand_load<WIDE>(slots + lower->src);
    u64* used = reinterpret_cast<u64*>(pool + rb + rcap);  // pairs (key, next); room checked by the caller
    u32 nused = 0;
    for (u64 j = lower->c1lo; j <= lower->c1hi; ++j) {
      Ent ti;  // ti.ranges = lower->raw
      ti.c0lo = lb.r0lo; ti.c0hi = (u64)lb.r0lo + lb.sz - 1; ti.c1lo = lb.r1lo; ti.c1hi = (u64)lb.r1lo + lb.sz - 1;
      ti.src = lower->src; ti.len = lower->len; ti.pad0 = ti.pad1 = 0;
      u64 tlo = j, thi = j;
      bool done = false;
      u32 guard = 0;
      while (!done) {
        u32 c = fm_char(Ofm, tlo);
        if (c == 0) {
          updateL(ti, 0, Pfm);
          nocc += 2;
          done = true;
        }
        // tracing.update(c, _rfmi) (fmindex.h:94-98)
        u64 l[5], u[5];
        fm_rank5<WIDE>(Ofm, tlo, l);
        fm_rank5<WIDE>(Ofm, thi + 1, u);
        u64 lr = 0, ur = 0;
        for (u32 b = 0; b < 5; ++b)
          if (b == c) { lr = l[b]; ur = u[b]; }
        u64 pb = tb.C[Ofm.which][c];
        tlo = pb + lr;
        thi = pb + ur - 1;
        updateR(ti, c, Ofm);
        nocc += 4;
        if (++guard > (1u << 20)) { overflow = true; return k; }
      }
      u64 forward;
      if (ti.c0lo == ti.c0hi) {
        forward = ti.c0lo;
      } else {  // duplicated read: next unused forward row (std::map usedmappig, :1050-1059)
        u64 key = ti.c0lo, idx = key;
        u32 f = 0;
        for (; f < nused; ++f)
          if (used[2 * f] == key) break;
        if (f < nused) idx = used[2 * f + 1];
        forward = idx;
        if (f == nused) { used[2 * f] = key; ++nused; }
        used[2 * f + 1] = idx + 1;
      }
      if (!intersecting(forward, forward, higher->c0lo, higher->c0hi)) {
        if (k >= rcap) { overflow = true; return k; }
        Ent sp = *lower;
        sp.c0lo = forward; sp.c0hi = forward; sp.c1lo = j; sp.c1hi = j;
        pool[rb + k++] = sp;
      }
    }
    return k;
  }

  // SubMaximalBlockFilter::filter (overlap_builder.cpp:919-954) on pool[base, base+n); returns the new size
  __device__ u32 filter(u32 base, u32 n, const FmRef& Pfm, const FmRef& Ofm) {
    if (n == 0) return 0;
    sort_left(base, n);
    u32 prev = 0, curr = 1;
    u32 guard = 0;
    while (curr < n) {
      Ent a = pool[base + prev];
      Ent b = pool[base + curr];
      if (intersecting(a.c0lo, a.c0hi, b.c0lo, b.c0hi)) {
        const Ent& lower = (a.len < b.len) ? a : b;  // resolve() keeps the longer whole and splits the other
        u64 w = lower.c1hi - lower.c1lo + 1;
        if (w > A.pool_cap) { overflow = true; return n; }
        u32 rcap = (u32)w + 1;
        u32 rb = base + n + rcap;  // leave room for the list to grow by the resolved entries
        if (!need(rb + rcap + (rcap * 16 + (u32)sizeof(Ent) - 1) / (u32)sizeof(Ent))) return n;
        u32 k = resolve(a, b, rb, rcap, Pfm, Ofm);
        if (overflow) return n;
        // resolved.sort(sorter)
        for (u32 i = 1; i < k; ++i) {
          Ent key = pool[rb + i];
          u32 j = i;
          while (j > 0 && key.c0lo < pool[rb + j - 1].c0lo) { pool[rb + j] = pool[rb + j - 1]; --j; }
          pool[rb + j] = key;
        }
        // blocks->erase(curr); blocks->erase(prev)   (adjacent)
        for (u32 i = prev; i + 2 < n; ++i) pool[base + i] = pool[base + i + 2];
        n -= 2;
        // blocks->merge(resolved, sorter): stable, *this before equivalent incoming
        int i = (int)n - 1, j = (int)k - 1;
        u32 dst = n + k;
        while (j >= 0) {
          if (i >= 0 && pool[rb + j].c0lo < pool[base + i].c0lo) pool[base + --dst] = pool[base + i--];
          else pool[base + --dst] = pool[rb + j--];
        }
        n += k;
        prev = 0;
        if (++guard > (1u << 16)) { overflow = true; return n; }
      } else {
        ++prev;
      }
      curr = prev + 1;
    }
    return n;
  }

  // updateR(c, &blocklist) (overlap_builder.cpp:818-832): update, drop invalid, keep order
  __device__ u32 updateR_list(u32 off, u32 cnt, u32 c) {
    u32 w = 0;
    for (u32 i = 0; i < cnt; ++i) {
      Ent e = pool[off + i];
      u32 b = (af_of(e.src) & 4u) ? comp_rank(c) : c;
      updateR(e, b, ext_index(e.src));
      if (iv_valid(e.c0lo, e.c0hi) && iv_valid(e.c1lo, e.c1hi)) pool[off + w++] = e;
    }
    return w;
  }

  // IrreducibleBlockListExtractor::extract (overlap_builder.cpp:711-809) on pool[base, base+n)
  __device__ bool extract(u32 base, u32 n) {
    sort_len_desc(base, n);
    top = base + n;
    // group table (list order) + this pass's incomings; sized with the pool so that a regrown pool also
    // admits more simultaneous branches
    const u32 GMAX = A.pool_cap / 12 > 32 ? A.pool_cap / 12 : 32;
    u32 dents = (GMAX * 2 * (u32)sizeof(uint2) + (u32)sizeof(Ent) - 1) / (u32)sizeof(Ent);
    if (!need(top + dents)) return true;
    uint2* D = reinterpret_cast<uint2*>(pool + top);  // groups (off, cnt) in list order
    uint2* I = D + GMAX;                              // incomings of the current pass
    top += dents;
    u32 ng = 1;
    D[0] = make_uint2(base, n);
    u32 guard = 0;
    while (ng > 0) {
      u32 ni = 0;
      u32 p = 0;
      while (p != ng) {
        u32 off = D[p].x, cnt = D[p].y;
        bool eraseGroup = true;
        if (cnt & FX_COUNTDOWN) {
          // a group of one single-row block waiting for the end of its read (GFx::round has the argument): c1lo = the
          // final row, c1hi = rounds still to go
          Ent& g = pool[off];
          if (g.c1hi == 0 || ng == 1 || (ng == 2 && p == 0 && ni == 0)) {
            nocc += 2 * (g.c1hi + 1);
            Ent br = g;
            br.c0hi = br.c0lo;
            br.c1hi = br.c1lo;
            emit(br);
          } else {
            g.c1hi -= 1;
            nocc += 2;
            eraseGroup = false;
          }
        } else if (cnt > 0) {
          u64 exts[5] = {0, 0, 0, 0, 0};
          u32 topLen = pool[off].len;
          u32 ntop = 0;
          for (; ntop < cnt && pool[off + ntop].len == topLen; ++ntop) {
            u64 x[5];
            ext(pool[off + ntop], x);
            nocc += 2;
            for (int k = 0; k < 5; ++k) exts[k] += x[k];
          }
          if (exts[0] > 0) {
            for (u32 j = 0; j < ntop; ++j) {
              u64 x[5];
              ext(pool[off + j], x);
              if (x[0] == 0) { xerror = true; return false; }  // "substring read found" (:754-757)
              Ent br = pool[off + j];
              updateR(br, 0, ext_index(br.src));
              emit(br);
            }
          } else {
            for (u32 j = ntop; j < cnt; ++j) {
              u64 x[5];
              ext(pool[off + j], x);
              nocc += 2;
              for (int k = 0; k < 5; ++k) exts[k] += x[k];
            }
            int nz = 0, first = -1;
            for (int k = 0; k < 5; ++k)
              if (exts[k] > 0) { ++nz; if (first < 0) first = k; }
            if (nz == 1) {
              D[p].y = updateR_list(off, cnt, (u32)first);
              eraseGroup = false;
            } else {
              for (int k = 0; k < 5; ++k) {
                if (exts[k] > 0) {
                  if (!need(top + cnt) || ni >= GMAX) { overflow = true; return true; }
                  for (u32 j = 0; j < cnt; ++j) pool[top + j] = pool[off + j];
                  u32 c2 = updateR_list(top, cnt, (u32)k);
                  if (c2 == 1 && pool[top].c1lo == pool[top].c1hi) {
                    const FmStrand& xs = (pool[top].src / A.cap) < 2 ? A.rev : A.fwd;  // ext_index()
                    if (xs.sa != nullptr) {
                      u32 ld, t;
                      row_lookup(xs, pool[top].c1lo, ld, t);
                      pool[top].c1lo = ld;
                      pool[top].c1hi = t;
                      c2 |= FX_COUNTDOWN;
                    }
                  }
                  I[ni++] = make_uint2(top, c2);
                  top += cnt;
                }
              }
            }
          }
        }
        // body `i = erase(i)` / `++i`, then the loop header's `++i`: stride-2 walk of the ring [g0..g(k-1), end]
        if (eraseGroup) {
          for (u32 i = p; i + 1 < ng; ++i) D[i] = D[i + 1];
          --ng;
        } else {
          p = (p + 1) % (ng + 1);
        }
        p = (p + 1) % (ng + 1);
        if (++guard > (1u << 22)) { overflow = true; return true; }
      }
      if (ng + ni > GMAX) { overflow = true; return true; }
      for (u32 i = 0; i < ni; ++i) D[ng++] = I[i];
    }
    return true;
  }

  // gather one find's blocks + the containment copies into pool[top..), filter, drop containments
  __device__ u32 build_list(u32 chain, u32 c_a, u32 c_b, const u32 cc[4], u64 L, const FmRef& Pfm, const FmRef& Ofm) {
    u32 base = top;
    u32 nchain = cc[chain] & SIGAX_CC_COUNT_MASK;
    if (!need(base + nchain + 2)) return 0;
    u32 n = 0;
    for (u32 k = 0; k < nchain; ++k) load_ent(pool[base + n++], chain * A.cap + k);
    if (cc[c_a] & SIGAX_CC_CONTAIN) load_ent(pool[base + n++], c_a * A.cap + (A.cap - 1));
    if (cc[c_b] & SIGAX_CC_CONTAIN) load_ent(pool[base + n++], c_b * A.cap + (A.cap - 1));
    n = filter(base, n, Pfm, Ofm);
    if (overflow) return 0;
    u32 w = 0;  // ContainmentBlockRemover (overlap_builder.cpp:1094-1111)
    for (u32 i = 0; i < n; ++i)
      if (pool[base + i].len != L) pool[base + w++] = pool[base + i];
    top = base + w;
    return w;
  }

  // OverlapBuilder::overlap from the list plumbing on (overlap_builder.cpp:1135-1181)
  __device__ void run(u32 r) {
    read = r;
    nout = 0;
    top = 0;
    overflow = false;
    xerror = false;
    slots = reinterpret_cast<const Cand<WIDE>*>(A.arena) + (u64)r * 4 * A.cap;
    u64 L = A.offs[r + 1] - A.offs[r];
    u32 cc[4];
    for (int o = 0; o < 4; ++o) cc[o] = A.chain_cnt[(u64)r * 4 + o];
    // containfwd = {chain 0, chain 1}, containrev = {chain 2, chain 3} go out first, unfiltered (:1161-1162)
    for (int o = 0; o < 4; ++o) {
      if (cc[o] & SIGAX_CC_CONTAIN) {
        Ent e;
        load_ent(e, o * A.cap + (A.cap - 1));
        emit(e);
      }
    }
    // The four filters are independent, so build prefix lists first and suffix lists on top: `suffixfwd +=
    // suffixrev` and `prefixfwd += prefixrev` are then contiguous ranges, and extract() of the suffix lists can
    // grow the pool above itself without touching the prefix lists.
    u32 pBase = top;
    u32 nPF = build_list(1, 0, 1, cc, L, F, R);
    u32 nPR = overflow ? 0 : build_list(2, 2, 3, cc, L, R, F);
    u32 sBase = top;
    u32 nSF = overflow ? 0 : build_list(0, 0, 1, cc, L, F, R);
    u32 nSR = overflow ? 0 : build_list(3, 2, 3, cc, L, R, F);
    if (overflow) return;
    u32 nS = nSF + nSR, nP = nPF + nPR;
    if (A.irreducible) {
      extract(sBase, nS);  // result.aborted |= ...: the second extract runs whatever the first returned
      if (overflow) return;
      extract(pBase, nP);
    } else {
      for (u32 i = 0; i < nS; ++i) emit(pool[sBase + i]);
      for (u32 i = 0; i < nP; ++i) emit(pool[pBase + i]);
    }
  }
};

// Held to 128 registers (it spills; it serves a fraction of a per cent of the reads): with its natural 170 no wave of it
// fits a SIMD beside the finder's resident waves, and a launch -- usually an EMPTY one -- waited milliseconds for a
// finder workgroup to retire on some CU while the batch's tail waited behind it.
template <bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_filter_extract(FxArgs A) {
  __shared__ FmTables tb;
  fm_tables_load(tb, A.fwd, A.rev);
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 nlanes = (u64)gridDim.x * 256;
  Fx<WIDE> fx(A, tb, A.pool + gid * A.pool_cap);
  u64 nocc = 0, nerr = 0, nover = 0, nsub = 0, nocc_back = 0, nerr_back = 0, nsub_back = 0;
  u64 fin_cur = 0, fin_end = 0;
  const u64 n_work = A.n_work_ptr ? *A.n_work_ptr : A.n_work;
  for (u64 w = gid; w < n_work; w += nlanes) {
    u32 r = A.work ? A.work[w] : (u32)w;
    // a side of this read that completed in the fast kernel already added its share: take it back
    for (int sd = 0; sd < 2; ++sd) {
      u32 w0 = A.occ_side[2ull * r + sd];
      nocc_back += w0 & 0x3FFFFFFFu;
      if (w0 & 0x80000000u) ++nerr_back;
      if (w0 & 0x40000000u) ++nsub_back;
      A.occ_side[2ull * r + sd] = 0;
    }
    fx.nocc = 0;
    fx.run(r);
    nocc += fx.nocc;
    if (fx.overflow) fx.nout = 0;
    fx.flush(fin_cur, fin_end);
    A.fin_cnt[2ull * r] = fx.nout;
    A.fin_cnt[2ull * r + 1] = 0;
    A.item_base[2ull * r + 1] = 0;
    if (fx.overflow) ++nover;
    if (fx.xerror) ++nerr;
    u32 sub = 0;
    for (int o = 0; o < 4; ++o) sub |= A.chain_cnt[(u64)r * 4 + o] & SIGAX_CC_SUBSTRING;
    A.substring[r] = sub ? 1 : 0;
    if (sub) ++nsub;
  }
  nocc = wave_sum(nocc); nerr = wave_sum(nerr); nover = wave_sum(nover); nsub = wave_sum(nsub);
  nocc_back = wave_sum(nocc_back); nerr_back = wave_sum(nerr_back); nsub_back = wave_sum(nsub_back);
  if ((threadIdx.x & 63) == 0) {
    if (nocc != nocc_back) atomicAdd(&A.dstat[DS_OCC_EXTRACT], nocc - nocc_back);  // wraps correctly when negative
    if (nocc) atomicAdd(&A.dstat[DS_SEC_EXTRACT], nocc);  // one granule per evaluation here (a fraction of a per cent of the reads)
    if (nerr != nerr_back) atomicAdd(&A.dstat[DS_EXTRACT_ERRORS], nerr - nerr_back);
    if (nover) atomicAdd(&A.dstat[DS_POOL_OVERFLOW], nover);
    if (nsub != nsub_back) atomicAdd(&A.dstat[DS_SUBSTRING], nsub - nsub_back);
  }
}

// -------------------------------------------------------------------------------------------------------
// k_filter_extract_fast: one lane group (32 or 64 lanes) per (read, side), one lane per overlap block.
//
// Side 0 = the suffix lists (finds 0 and 3), side 1 = the prefix lists (finds 1 and 2): the two halves of
// OverlapBuilder::overlap (overlap_builder.cpp:1135-1173) are independent until the final push order, which
// the ordered scatter restores (containments, suffix side, prefix side).  Same algorithm as Fx above,
// parallelised over the blocks of a list.  A group of the extractor (overlap_builder.cpp:720-722) is a 64-bit
// alive mask over fixed lane slots: block order inside a group is lane order, so "top-level blocks" are the
// alive lanes whose length equals the first alive lane's, erasing an invalid block clears its bit, and a
// branch copies the lanes' entries to a fresh 64-entry slot of the wave's pool.  The group being worked on
// lives in registers; with a single group (the common case: every overlapping read agrees on the next base) a
// whole extraction never touches the pool.  Symbol tests ("exts[k] > 0") are wave ballots; the sorted order is
// produced by a rank + LDS permutation; emitted blocks wait in LDS until the side is complete.
// Positions are u32 when the index has < 2^32 symbols (P), halving register use.
// A side that needs SubMaximalBlockFilter::resolve, has more than 64 blocks, or outgrows the wave's pool queues
// its read for the general kernel; nothing that the scatter will use is written before a side has completed.
// -------------------------------------------------------------------------------------------------------
#define FX_NSLOT 24
#define FX_OUTCAP 64
#define FX_WPOOL (FX_NSLOT * 64)

#define FX_FIN_CHUNK 512
#define OCC_SIDE_ERR 0x80000000u
#define OCC_SIDE_SUB 0x40000000u
#define OCC_SIDE_MASK 0x3FFFFFFFu


template <bool WIDE>
struct SideSh {
  typedef typename PosOf<WIDE>::type P;
  P e0[64], e1[64], e2[64], e3[64];  // permutation buffer / adjacency scratch
  u32 esrc[64], elen[64];
  P o0[FX_OUTCAP], o1[FX_OUTCAP], o2[FX_OUTCAP], o3[FX_OUTCAP];  // emitted blocks of this side
  u32 osrc[FX_OUTCAP];
  u64 alive[FX_NSLOT];
  unsigned char D[FX_NSLOT], I[FX_NSLOT];
  u32 nsec;  // over the whole launch: distinct 64-byte sectors of the rank tables this wave's lane groups asked for
#ifdef SIGAX_FX_PROFILE
  u32 prof[16];  // which path each extension round took (diagnostic builds only)
  u32 prof2[8];
#endif
};
#ifdef SIGAX_FX_PROFILE
#define FXP(i) do { if (gl == 0) atomicAdd(&sh.prof[i], 1u); } while (0)
#else
#define FXP(i) do { } while (0)
#endif

// scratch of the 64-lane launch for items of 65..256 blocks (four per lane): bounds / sorted entries
#define FX_BIGCAP 256
template <bool WIDE>
struct BigSh {
  typedef typename PosOf<WIDE>::type P;
  P b0[FX_BIGCAP], b1[FX_BIGCAP], b2[FX_BIGCAP], b3[FX_BIGCAP];
  u32 bsrc[FX_BIGCAP], blen[FX_BIGCAP];
};

__device__ __forceinline__ u64 readlane64(u64 v, u32 l) {
  u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
// Tell the compiler a value is the same in every lane (it cannot see that for values derived from the wave index or
// loaded through vector memory / LDS): keeps loop counters, masks and branch conditions in SGPRs.
__device__ __forceinline__ u32 uni(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ u64 widen(u32 v) { return v == 0xFFFFFFFFu ? ~0ull : (u64)v; }  // keep "-1" a -1
__device__ __forceinline__ u64 widen(u64 v) { return v; }

// rank vector at position p in the position type of the index ($ first)
template <bool WIDE>
__device__ __forceinline__ void fm_rank5p(const FmRef& s, typename PosOf<WIDE>::type p, typename PosOf<WIDE>::type v[5]) {
  typedef typename PosOf<WIDE>::type P;
  u64 pc = (u64)p > s.n ? s.n : (u64)p;
  u64 gi = pc >> 7;
  int r = (int)(pc & 127u);
  const uint4* q = s.g + gi * 4;
  uint4 k0 = q[0], k1 = q[1], k2 = q[2], k3 = q[3];
  u32 a = k0.x, c = k1.x, g = k2.x, t = k3.x;
  chunk_count(k0, r, a, c, g, t);
  chunk_count(k1, r - 32, a, c, g, t);
  chunk_count(k2, r - 64, a, c, g, t);
  chunk_count(k3, r - 96, a, c, g, t);
  P A = a, C = c, G = g, T = t;
  if (WIDE) {
    const u64* sb = s.super + (pc >> SIGAX_SUPER_SHIFT) * 4;
    A += (P)sb[0]; C += (P)sb[1]; G += (P)sb[2]; T += (P)sb[3];
  }
  v[1] = A; v[2] = C; v[3] = G; v[4] = T;
  v[0] = (P)pc - (A + C + G + T);
}

// Both ends of a range at once: getOcc(c, lo - 1) and getOcc(c, hi) as fm_rank5p(lo) and fm_rank5p(hi1 = hi + 1).  A
// narrow range has both positions in one granule, and the lanes for which that is so do not load it a second time
// (find_step2_loads_masked says what a load costs the CU).
template <bool WIDE>
__device__ __forceinline__ void fm_rank5p_pair(const FmRef& s, typename PosOf<WIDE>::type lo, typename PosOf<WIDE>::type hi1,
                                               typename PosOf<WIDE>::type l[5], typename PosOf<WIDE>::type u[5]) {
  typedef typename PosOf<WIDE>::type P;
  const u64 pl = (u64)lo > s.n ? s.n : (u64)lo, pu = (u64)hi1 > s.n ? s.n : (u64)hi1;
  const uint4* q = s.g + (pl >> 7) * 4;
  uint4 k0 = q[0], k1 = q[1], k2 = q[2], k3 = q[3];
  auto count = [&](u64 pc, P v[5]) {
    const int r = (int)(pc & 127u);
    u32 a = k0.x, c = k1.x, g = k2.x, t = k3.x;
    chunk_count(k0, r, a, c, g, t);
    chunk_count(k1, r - 32, a, c, g, t);
    chunk_count(k2, r - 64, a, c, g, t);
    chunk_count(k3, r - 96, a, c, g, t);
    P A = a, C = c, G = g, T = t;
    if (WIDE) {
      const u64* sb = s.super + (pc >> SIGAX_SUPER_SHIFT) * 4;
      A += (P)sb[0]; C += (P)sb[1]; G += (P)sb[2]; T += (P)sb[3];
    }
    v[1] = A; v[2] = C; v[3] = G; v[4] = T;
    v[0] = (P)pc - (A + C + G + T);
  };
  count(pl, l);
  if ((pu >> 7) != (pl >> 7)) {
    const uint4* qu = s.g + (pu >> 7) * 4;
    k0 = qu[0]; k1 = qu[1]; k2 = qu[2]; k3 = qu[3];
  }
  count(pu, u);
}

// LEAN (32-lane launch, 32-bit positions, irreducible mode, two-step tables present -- the common case): only the
// single-group rounds served by the two-step line are compiled in; an item that needs anything else (a branch, a range
// across lines, the exhaustive output order) is queued for the 64-lane launch, which has everything.  Without the rarely
// taken code the kernel needs no scratch (124 VGPRs, was 128 + 17 spilled) and runs 1.2-1.4x faster.
// LEANP = 2: the same for indexes without two-step tables: only the one-granule rounds.
// LEANP = 5 / 6: strict / branching, with the rounds read off the block's read text (row table or direct map + stretch text,
// fm_layout.h) instead of computed from rank lines: one lookup per block, then a text window per 28 rounds; items with a
// block of more than one row go on to the full launch.
// LEANP = 3 / 4: LEANP 1 / 2 plus branches of single-row blocks (branch_inreg) and the group ring of extract(): what reads
// with substitutions need.  The strict forms run first because the extra state costs them 17 VGPRs (89 -> 106) and, beside
// the finder, 9 % of the error-free step; they hand a branching item to the next launch, which has these forms.
template <bool WIDE, int W, int LEANP = 0>
struct GFx {
  typedef typename PosOf<WIDE>::type P;
  static constexpr bool LEAN = LEANP != 0;
  static constexpr bool BR = LEANP == 0 || LEANP == 3 || LEANP == 4 || LEANP == 6;  // follows in-register branches
  static constexpr bool TWO_ONLY = LEANP == 1 || LEANP == 3, ONE_ONLY = LEANP == 2 || LEANP == 4;
  static constexpr bool TEXT = LEANP == 5 || LEANP == 6;  // rounds from the reads' text only (single-row blocks)
  struct E {  // a block's capped pair in registers
    P c0lo, c0hi, c1lo, c1hi;
    u32 src;  // bits 30-31: which find produced it (0..3); bits 0-29: slot in the read's candidate region
    u32 len;
  };
  static __device__ u32 slot_of(u32 src) { return src & 0x3FFFFFFFu; }
  static __device__ u32 find_of(u32 src) { return src >> 30; }
  const FxArgs& A;
  const FmTables& tb;
  SideSh<WIDE>& sh;
  FmRef F, R;
  Ent* wpool;  // FX_NSLOT group slots of 64 entries (branch copies only), shared by the wave's lane groups
  BigSh<WIDE>* big;  // 64-lane launch only: scratch for items of more than 64 blocks
  const Find2TablesT<WIDE>* t2;  // constants of the two-step table, or NULL when the index has none
  u32 lane;    // lane in the wave
  u32 gb;      // first lane of this lane's group (0; 0 or 32 when W == 32; 0, 16, 32 or 48 when W == 16)
  static constexpr u32 kOutCap = FX_OUTCAP * (u32)W / 64u;            // a group's share of the wave's output slots
  static constexpr u32 kNSlot = (u32)W < FX_NSLOT ? (u32)W : FX_NSLOT;  // group slots: the table lives in the group's lanes
  u32 gl;      // lane inside the group
  u64 glt;     // group lanes below this one, as a mask over group lanes
  const Cand<WIDE>* slots;
  u32 nout;
  u32 nocc;
  bool xerror;
  u64 fin_cur, fin_end;  // this wave's chunk of the unordered final-block arena (wave-uniform)

  __device__ GFx(const FxArgs& a, const FmTables& t, SideSh<WIDE>& s, Ent* wp, const Find2TablesT<WIDE>* tt)
      : A(a), tb(t), sh(s), F(fm_ref(a.fwd, 0)), R(fm_ref(a.rev, 1)), wpool(wp), big(nullptr), t2(tt), lane(threadIdx.x & 63u),
        gb(threadIdx.x & (63u & ~(u32)(W - 1))), gl(threadIdx.x & (u32)(W - 1)),
        glt((1ull << (threadIdx.x & (u32)(W - 1))) - 1ull), slots(nullptr), nout(0), nocc(0),
        xerror(false), fin_cur(0), fin_end(0), nslot(1), ni(0), gAlive(0), gD(0), gI(0) {}

  // ---- lane-group primitives: a group is the whole wave (W == 64), a half (W == 32) or a quarter of it (W == 16) ----
  __device__ u64 gballot(bool p) const {
    u64 b = __ballot(p);
    return W == 64 ? b : ((b >> gb) & ((1ull << (W & 63)) - 1ull));
  }
  __device__ u32 gshfl(u32 v, u32 idx) const { return (u32)__shfl((int)v, (int)(gb + idx), 64); }
  __device__ u64 gshfl(u64 v, u32 idx) const {
    u32 lo = gshfl((u32)v, idx), hi = gshfl((u32)(v >> 32), idx);
    return ((u64)hi << 32) | lo;
  }
  static __device__ u32 pop(u64 m) { return (u32)__popcll(m); }
  static __device__ u32 ffs0(u64 m) { return (u32)__ffsll((long long)m) - 1u; }

  static __device__ bool valid(P lo, P hi) { return hi != (P)~(P)0 && hi >= lo; }  // fmindex.h:87-89

  __device__ u32 af_of(u32 src) const {
    u32 ch = find_of(src);
    return ch == 0 ? SIGAX_AF_CHAIN0 : ch == 1 ? SIGAX_AF_CHAIN1 : ch == 2 ? SIGAX_AF_CHAIN2 : SIGAX_AF_CHAIN3;
  }
  // OverlapBlock::index (overlap_builder.cpp:177-179): !TARGETREV ? rfmi : fmi.  Finds 0,1 are !TARGETREV.
  __device__ FmRef ext_index(u32 src) const { return fm_pick(find_of(src) < 2, R, F); }

  __device__ void load_block(E& e, u32 src) const {
    const Cand<WIDE> b = cand_load<WIDE>(slots + slot_of(src));
    e.c0lo = b.c0lo; e.c0hi = b.c0lo + b.d - 1; e.c1lo = b.c1lo; e.c1hi = b.c1lo + b.d - 1;
    e.src = src; e.len = b.len;
  }
  __device__ void pool_put(Ent* dst, const E& e) const {
    ulonglong2* d = reinterpret_cast<ulonglong2*>(dst);
    d[0] = make_ulonglong2(widen(e.c0lo), widen(e.c0hi));
    d[1] = make_ulonglong2(widen(e.c1lo), widen(e.c1hi));
    d[2] = make_ulonglong2((u64)e.src | ((u64)e.len << 32), 0ull);
  }
  __device__ void pool_get(E& e, const Ent* src) const {
    const ulonglong2* d = reinterpret_cast<const ulonglong2*>(src);
    ulonglong2 a = d[0], b = d[1], c = d[2];
    e.c0lo = (P)a.x; e.c0hi = (P)a.y; e.c1lo = (P)b.x; e.c1hi = (P)b.y;
    e.src = (u32)c.x; e.len = (u32)(c.x >> 32);
  }
  __device__ void out_put(u32 i, const E& e) {  // i = position in this group's output
    sh.o0[gb + i] = e.c0lo; sh.o1[gb + i] = e.c0hi; sh.o2[gb + i] = e.c1lo; sh.o3[gb + i] = e.c1hi; sh.osrc[gb + i] = e.src;
  }

  __device__ void sec_add(u32 v) {  // v is the same in all lanes of the group
    if (gl == 0) atomicAdd(&sh.nsec, v);
  }

  static __device__ P sel5(const P v[5], u32 k) { return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : k == 3 ? v[3] : v[4]; }

  // IntervalPair::updateR(b, index) (overlap_builder.cpp:101-106,123-133) from the two rank vectors in registers
  __device__ void apply_updateR(E& e, u32 b, u32 which, const P l[5], const P u[5]) const {
    P d0 = u[0] - l[0], d1 = u[1] - l[1], d2 = u[2] - l[2], d3 = u[3] - l[3];
    P acc = b == 0 ? (P)0 : b == 1 ? d0 : b == 2 ? d0 + d1 : b == 3 ? d0 + d1 + d2 : d0 + d1 + d2 + d3;
    P lb = sel5(l, b), ub = sel5(u, b);
    P pb = (P)tb.C[which][b];
    e.c0lo += acc;
    e.c0hi = e.c0lo + (ub - lb) - 1;
    e.c1lo = pb + lb;
    e.c1hi = pb + ub - 1;
  }

  // ---- IrreducibleBlockListExtractor::extract (overlap_builder.cpp:711-809) ----
  // The group table lives in VGPR lanes of the lane group: lane s holds slot s's alive mask (gAlive), lane i holds
  // the i-th group of the list (gD) / of this pass's incomings (gI).  Every scalar of the algorithm (ng, p, slot,
  // masks) is a per-lane value that is equal across the lane group, so two lane groups of one wave can be at
  // different points of their extractions.
  u32 nslot, ni;  // slots handed out, incomings of the current pass
  u64 gAlive;
  u32 gD, gI;
  bool inreg;     // this item's groups are disjoint lane sets: no pool traffic
  bool toowide;   // body() gave up because the item has more blocks than the group has lanes
  // TEXT: this lane's block on its backward path: the symbols ahead (next one in the top nibble), how many of the window
  // are used up, and from the row table its stretch = Occ('$') at the path's end and the rounds until then
  u64 tw;
  u32 tk, tld, ttt;
  u32 ccor;  // OR of the item's four chain_cnt words (body())
#ifdef SIGAX_FX_PROFILE
  u32 dbg_round;
#endif

  enum { RD_ENDED = 0, RD_UPDATED, RD_BRANCHED, RD_BAIL, RD_XERROR };

  __device__ void to_countdown(E& e) const {
    const FmStrand& xs = find_of(e.src) < 2 ? A.rev : A.fwd;
    if (xs.sa != nullptr) {
      u32 ld, t;
      row_lookup(xs, (u64)e.c1lo, ld, t);
      e.c1lo = (P)ld;
      e.c1hi = (P)t;
      e.len |= FX_COUNTDOWN;
    }
  }

  // A branch of a group of single-row blocks (inreg): every block follows the one symbol at its row, so the branch
  // partitions the lanes by that symbol (complemented for QUERYCOMP blocks, :181-187) and each lane's update is the
  // single-symbol one with ITS symbol -- `v` = C[c] + Occ(c, row), computed by the caller from the line it already holds.
  // Groups go to the incomings in rank order A, C, G, T (:781-787).  Not for '$' below the top level (the caller checks).
  __device__ int branch_inreg(E& e, u64 alive, bool mine, u32 cq, P v) {
    const u32 NSLOT = kNSlot;
    const u64 m1 = gballot(mine && cq == 1u), m2 = gballot(mine && cq == 2u), m3 = gballot(mine && cq == 3u), m4 = gballot(mine && cq == 4u);
    const u32 nb = (m1 != 0) + (m2 != 0) + (m3 != 0) + (m4 != 0);
    if (nslot + nb > NSLOT || ni + nb > NSLOT) return RD_BAIL;
    nocc += 2u * pop(alive);
    if (mine) {
      e.c1lo = v;
      e.c1hi = v;
    }
    bool lonely = false;
#pragma unroll
    for (u32 sy = 1; sy <= 4; ++sy) {
      const u64 m = sy == 1 ? m1 : sy == 2 ? m2 : sy == 3 ? m3 : m4;
      if (!m) continue;
      const u32 ns = nslot++;
      if (gl == ns) gAlive = m;
      if (gl == ni) gI = ns;
      ++ni;
      if (mine && cq == sy && (m & (m - 1ull)) == 0) lonely = true;
    }
    if (lonely) to_countdown(e);
    return RD_BRANCHED;
  }

  // One extension round of one group (the body of the loop at :728-802).  `alive` = the group's blocks, in lane order.
  // RD_UPDATED: the blocks were right-extended in place, *newAlive = those still valid.  RD_ENDED / RD_BRANCHED: the
  // group is finished (top-level blocks emitted, or copies pushed to the incomings).
  __device__ int round(E& e, u64 alive, u64* newAlive) {
    const u32 OUTCAP = kOutCap;
    const u32 NSLOT = kNSlot;
    const bool mine = (alive >> gl) & 1ull;
    const u32 first = ffs0(alive);
    const u32 topLen = gshfl(e.len, first);
    const bool isTop = mine && e.len == topLen;
    const FmRef ix = ext_index(e.src);
    const bool qcomp = (af_of(e.src) & 4u) != 0;
    P l[5] = {0, 0, 0, 0, 0}, u[5] = {0, 0, 0, 0, 0};
    if (mine) {
      fm_rank5p_pair<WIDE>(ix, e.c1lo, (P)(e.c1hi + 1), l, u);
    }
    sec_add(pop(alive) + pop(gballot(mine && ((u64)e.c1lo >> 7) != (((u64)e.c1hi + 1ull) >> 7))));
    // OverlapBlock::ext (overlap_builder.cpp:181-187), complemented for QUERYCOMP blocks
    const bool x0 = mine && (u[0] != l[0]);
    const bool xa = mine && (qcomp ? (u[4] != l[4]) : (u[1] != l[1]));
    const bool xc = mine && (qcomp ? (u[3] != l[3]) : (u[2] != l[2]));
    const bool xg = mine && (qcomp ? (u[2] != l[2]) : (u[3] != l[3]));
    const bool xt = mine && (qcomp ? (u[1] != l[1]) : (u[4] != l[4]));
    const u64 topMask = gballot(isTop);
    if (gballot(isTop && x0)) {
      // the top-level block has ended: emit the top-level blocks in list order (:747-766)
      u64 bad = gballot(isTop && !x0);
      u64 emitMask = topMask;
      if (bad) emitMask &= (1ull << ffs0(bad)) - 1ull;
      nocc += 2u * pop(topMask);
      u32 ne = pop(emitMask);
      if (nout + ne > OUTCAP) return RD_BAIL;
      if ((emitMask >> gl) & 1ull) {
        E br = e;
        apply_updateR(br, 0, ix.which, l, u);
        out_put(nout + pop(emitMask & glt), br);
      }
      nout += ne;
      if (bad) {
        xerror = true;  // "substring read found during overlap computation" (:754-757): extract() returns false
        return RD_XERROR;
      }
      return RD_ENDED;
    }
    nocc += 2u * pop(alive);
    u64 any0 = gballot(x0), any1 = gballot(xa), any2 = gballot(xc), any3 = gballot(xg), any4 = gballot(xt);
    u32 nz = (any0 != 0) + (any1 != 0) + (any2 != 0) + (any3 != 0) + (any4 != 0);
    if (nz == 1) {
      u32 c = any0 ? 0u : any1 ? 1u : any2 ? 2u : any3 ? 3u : 4u;
      u32 b = qcomp ? comp_rank(c) : c;
      if (mine) apply_updateR(e, b, ix.which, l, u);
      bool ok = mine && valid(e.c0lo, e.c0hi) && valid(e.c1lo, e.c1hi);
      *newAlive = gballot(ok);
      return RD_UPDATED;
    }
    const E e0 = e;
    bool lonely = false;
    for (u32 c = 0; c < 5; ++c) {
      u64 ak = c == 0 ? any0 : c == 1 ? any1 : c == 2 ? any2 : c == 3 ? any3 : any4;
      if (!ak) continue;
      if (nslot >= NSLOT || ni >= NSLOT) return RD_BAIL;
      u32 ns = nslot++;
      E br = e0;
      u32 b = qcomp ? comp_rank(c) : c;
      if (mine) apply_updateR(br, b, ix.which, l, u);
      bool ok = mine && valid(br.c0lo, br.c0hi) && valid(br.c1lo, br.c1hi);
      if (ok) {
        // single-row blocks follow exactly one symbol: the branch PARTITIONS the group's lanes and every lane keeps its
        // block in registers; only blocks with wider ranges are copied (to the wave's pool in global memory)
        if (inreg) e = br;
        else pool_put(wpool + ns * 64 + lane, br);
      }
      u64 m = gballot(ok);
      if (ok && (m & (m - 1ull)) == 0) lonely = true;
      if (gl == ns) gAlive = m;
      if (gl == ni) gI = ns;
      ++ni;
    }
    // A group of ONE single-row block (the usual branch: an overlapping read with a substitution right of the overlap) has
    // nothing left to decide: it follows its read to the end and is emitted there.  The row table says after how many
    // rounds and with which final range; from here on the group is a countdown (extract()), not a walk of ~100 lookups.
    if (inreg && lonely) to_countdown(e);
    return RD_BRANCHED;
  }

  // The usual round, cheaply.  When every alive block's capped[1] range lies inside one rank granule, which symbols
  // follow the range is a bit test on the granule's planes (no counting), and when exactly one symbol c follows all
  // blocks, IntervalPair::updateR(b) collapses: every extension of a block is b, so diff[k < b] = 0 and diff[b] = the
  // range size, i.e. capped[0] does not move and capped[1] = C[b] + Occ(b, lower - 1) keeps its size: one symbol's rank
  // at one position.  Anything else (range across granules, a branch, '$' as the single symbol) goes to round().
  __device__ int round_fast(E& e, u64 alive, u64* newAlive, bool allow2) {
    const u32 OUTCAP = kOutCap;
    const bool mine = (alive >> gl) & 1ull;
    const FmRef ix = ext_index(e.src);
    const u64 p0 = (u64)e.c1lo, p1 = (u64)e.c1hi + 1ull;  // [p0, p1) in the extension index
    FXP(0);
    // With the two-step table (fm_layout.h), when every alive block's range lies in one 64-row granule, one 128-byte line
    // per block serves: the end of the top-level block ('$' follows it); with a single group, TWO rounds at once when
    // every range holds one (first, second) symbol pair, the same pair (after complementing) for all
    // blocks and neither of them '$': both rounds are then "usual" rounds (no top-level end, no branch), capped[0] and
    // the range size do not move, and capped[1].lower = C[e] + Occ(e, C[c]) + R2(e, c, lower).  If only the first
    // symbol is common, ONE round from the same line.  Anything else goes on to the one-step forms below.
    if (!ONE_ONLY && t2 != nullptr) {
      const P q0 = (P)p0, q1 = (P)p1;
      const bool in2 = mine && q1 > q0 && ((q1 - 1u) >> 6) == (q0 >> 6) && p1 <= ix.n;
      if (gballot(mine && !in2) != 0) FXP(9);  // a range crosses a 64-row line: no two-step lookup
      if (gballot(mine && !in2) == 0) {
        FXP(8);
#ifdef SIGAX_FX_PROFILE
        {  // how many 128-byte lines do the alive blocks of this round span?
          u32 mn = mine ? (u32)(q0 >> 6) : 0xFFFFFFFFu, mx = mine ? (u32)(q0 >> 6) : 0u;
          for (int off = (W == 64 ? 32 : 16); off > 0; off >>= 1) {
            mn = min(mn, (u32)__shfl_xor((int)mn, off, 64));
            mx = max(mx, (u32)__shfl_xor((int)mx, off, 64));
          }
          const u32 span = mx - mn;
          const u32 rr = dbg_round < 3 ? dbg_round : 3;
          if (gl == 0) atomicAdd(&sh.prof[12 + (span == 0 ? 0 : span == 1 ? 1 : 2)], 1u);
          if (gl == 0 && span == 0) atomicAdd(&sh.prof2[rr], 1u);
          if (gl == 0) atomicAdd(&sh.prof2[4 + rr], 1u);
          ++dbg_round;
        }
#endif
        const u32* gq = (find_of(e.src) < 2 ? A.rev.gran2 : A.fwd.gran2) + (u64)(q0 >> 6) * SIGAX_GRAN2_WORDS;
        // 64-bit positions: the line's counters are relative to its superblock (fm_layout.h)
        const u64* sq = WIDE ? (find_of(e.src) < 2 ? A.rev.super2 : A.fwd.super2) + ((u64)q0 >> SIGAX_SUPER_SHIFT) * 20 : nullptr;
        uint4 a4 = make_uint4(0, 0, 0, 0), a5 = a4, a6 = a4, a7 = a4;
        if (mine) {
          const uint4* pq = reinterpret_cast<const uint4*>(gq + 20);
          a4 = *reinterpret_cast<const uint4*>(gq);  // A, C, G, T before the granule
          a5 = pq[0]; a6 = pq[1]; a7 = pq[2];
        }
        sec_add(2u * pop(alive));
        const u32 r0 = (u32)q0 & 63u, r1 = r0 + (u32)(q1 - q0);  // 0 <= r0 < r1 <= 64
        const u32 lo0 = r0 < 32u ? r0 : 32u, hi0 = r1 < 32u ? r1 : 32u;           // the range inside the low word
        const u32 lo1 = r0 > 32u ? r0 - 32u : 0u, hi1 = r1 > 32u ? r1 - 32u : 0u;  // ... inside the high word
        const u32 w0 = hi0 - lo0, w1 = hi1 - lo1;
        const u32 rm0 = w0 == 32u ? 0xFFFFFFFFu : (((1u << (w0 & 31u)) - 1u) << (lo0 & 31u));
        const u32 rm1 = w1 == 32u ? 0xFFFFFFFFu : (((1u << (w1 & 31u)) - 1u) << (lo1 & 31u));
        const u32 bl0 = lo0 == 32u ? 0xFFFFFFFFu : ((1u << lo0) - 1u);
        const u32 bl1 = (1u << lo1) - 1u;  // lo1 <= 31
        const bool hiw = r0 >= 32u;
        const u32 bit = r0 & 31u;
        // planes: a5 = y1lo y1hi z1lo z1hi, a6 = w1lo w1hi y2lo y2hi, a7 = z2lo z2hi w2lo w2hi
        const u32 fy = 0u - (((hiw ? a5.y : a5.x) >> bit) & 1u), fz = 0u - (((hiw ? a5.w : a5.z) >> bit) & 1u);
        const u32 fw = 0u - (((hiw ? a6.y : a6.x) >> bit) & 1u);
        const u32 gy = 0u - (((hiw ? a6.w : a6.z) >> bit) & 1u), gz = 0u - (((hiw ? a7.y : a7.x) >> bit) & 1u);
        const u32 gw = 0u - (((hiw ? a7.w : a7.z) >> bit) & 1u);
        const u32 d10 = (a5.x ^ fy) | (a5.z ^ fz) | (a6.x ^ fw), d11 = (a5.y ^ fy) | (a5.w ^ fz) | (a6.y ^ fw);
        const u32 d20 = (a6.z ^ gy) | (a7.x ^ gz) | (a7.z ^ gw), d21 = (a6.w ^ gy) | (a7.y ^ gz) | (a7.w ^ gw);
        const u32 first2 = ffs0(alive);
        {
          // the top-level block has ended (:747-766), from the same line: capped.updateR('$') needs Occ('$') at both
          // ends of the range = position minus the A,C,G,T before it
          const u32 ds0 = ~(a5.x | a5.z | a6.x), ds1 = ~(a5.y | a5.w | a6.y);  // rows whose symbol is '$'
          const bool x0 = mine && (((ds0 & rm0) | (ds1 & rm1)) != 0);
          const u32 topLen = gshfl(e.len, first2);
          const bool isTop = mine && e.len == topLen;
          if (gballot(isTop && x0)) {
            const u64 topMask = gballot(isTop);
            const u64 bad = gballot(isTop && !x0);
            u64 emitMask = topMask;
            if (bad) emitMask &= (1ull << ffs0(bad)) - 1ull;
            nocc += 2u * pop(topMask);
            const u32 ne = pop(emitMask);
            if (nout + ne > OUTCAP) return RD_BAIL;
            if ((emitMask >> gl) & 1ull) {
              P acgt = (P)(a4.x + a4.y + a4.z + a4.w + __popc(~ds0 & bl0) + __popc(~ds1 & bl1));
              if (WIDE) acgt += (P)(sq[0] + sq[1] + sq[2] + sq[3]);
              const u32 nd = __popc(ds0 & rm0) + __popc(ds1 & rm1);
              const P ld = q0 - acgt;  // Occ('$', lower - 1); C['$'] = 0
              E br = e;
              br.c0hi = br.c0lo + (P)nd - 1;
              br.c1lo = ld;
              br.c1hi = ld + (P)nd - 1;
              out_put(nout + pop(emitMask & glt), br);
            }
            nout += ne;
            if (bad) {
              xerror = true;
              return RD_XERROR;
            }
            FXP(1);
            return RD_ENDED;
          }
        }
        const bool qcomp2 = (af_of(e.src) & 4u) != 0;
        const u32 c = fw ? 4u : ((fy & 1u) | (fz & 2u));
        const u32 x = gw ? 4u : ((gy & 1u) | (gz & 2u));
        const u32 cq = (qcomp2 && c) ? 5u - c : c, xq = (qcomp2 && x) ? 5u - x : x;
        const u32 cfirst = gshfl(cq, first2);
        const bool diff1 = ((d10 & rm0) | (d11 & rm1)) != 0;
        if (cfirst != 0 && gballot(mine && (diff1 || cq != cfirst)) == 0) {
          const u32 xfirst = gshfl(xq, first2);
          const bool diff2 = ((d20 & rm0) | (d21 & rm1)) != 0;
          const bool two = allow2 && xfirst != 0 && gballot(mine && (diff2 || xq != xfirst)) == 0;
          if (mine) {
            const P size = e.c1hi - e.c1lo;
            if (two) {
              const u32 hi = 4 + (c - 1u) * 4 + (x - 1u);
              const u32 below = __popc(~(d10 | d20) & bl0) + __popc(~(d11 | d21) & bl1);
              P v = (P)tb.C[ix.which][x] + t2->Cc[ix.which][c - 1][x - 1] + (P)(gq[hi] + below);
              if (WIDE) v += (P)sq[hi];
              e.c1lo = v;
            } else {
              const u32 below = __popc(~d10 & bl0) + __popc(~d11 & bl1);
              P v = (P)tb.C[ix.which][c] + (P)(gq[c - 1u] + below);
              if (WIDE) v += (P)sq[c - 1u];
              e.c1lo = v;
            }
            e.c1hi = e.c1lo + size;
          }
          nocc += (two ? 4u : 2u) * pop(alive);
          *newAlive = alive;
          FXP(1);
          if (two) FXP(10);
          return RD_UPDATED;
        }
        if (BR && inreg && gballot(mine && (diff1 || c == 0u)) == 0) {
          P v = 0;
          if (mine) {
            v = (P)tb.C[ix.which][c] + (P)(gq[c - 1u] + __popc(~d10 & bl0) + __popc(~d11 & bl1));
            if (WIDE) v += (P)sq[c - 1u];
          }
          FXP(5);
          return branch_inreg(e, alive, mine, cq, v);
        }
      }
    }
    if (TWO_ONLY) return RD_BAIL;  // not servable from one two-step line
    const u64 g0 = p0 >> 7;
    const bool inside = mine && p1 > p0 && ((p1 - 1) >> 7) == g0 && p1 <= ix.n;
    if (gballot(mine && !inside)) {
      FXP(4);
      if (LEAN) return RD_BAIL;
      return round(e, alive, newAlive);
    }
    const bool qcomp = (af_of(e.src) & 4u) != 0;
    uint4 k[4];
    k[0] = k[1] = k[2] = k[3] = make_uint4(0, 0, 0, 0);
    if (mine) {
      const uint4* q = ix.g + g0 * 4;
      k[0] = q[0]; k[1] = q[1]; k[2] = q[2]; k[3] = q[3];
    }
    sec_add(pop(alive));
    const int r0 = (int)(p0 & 127u), r1 = (int)(p1 - (g0 << 7));  // 0 <= r0 < r1 <= 128
    const u32 first = ffs0(alive);
    // Cheapest form, and the usual one between two read ends: every alive block's range holds ONE symbol, the same
    // (after complementing) for all of them.  "One symbol" = no position of the range differs from the symbol at its
    // first position, a xor/or over the planes; the positions that equal it also give the rank for the update.
    {
      const u32 j0 = (u32)r0 >> 5, bit0 = (u32)r0 & 31u;
      const bool j0b0 = (j0 & 1u) != 0, j0b1 = (j0 & 2u) != 0;
      const u32 ys = j0b1 ? (j0b0 ? k[3].y : k[2].y) : (j0b0 ? k[1].y : k[0].y);
      const u32 zs = j0b1 ? (j0b0 ? k[3].z : k[2].z) : (j0b0 ? k[1].z : k[0].z);
      const u32 ws = j0b1 ? (j0b0 ? k[3].w : k[2].w) : (j0b0 ? k[1].w : k[0].w);
      const u32 fy = 0u - ((ys >> bit0) & 1u), fz = 0u - ((zs >> bit0) & 1u), fw = 0u - ((ws >> bit0) & 1u);
      u32 diff = 0, cntb = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lo = min(max(r0 - 32 * j, 0), 32), hi = min(max(r1 - 32 * j, 0), 32), wd = hi - lo;
        const u32 rm = wd == 32 ? 0xFFFFFFFFu : (((1u << (wd & 31)) - 1u) << (lo & 31));
        const u32 below = lo == 32 ? 0xFFFFFFFFu : ((1u << (lo & 31)) - 1u);
        const u32 d = (k[j].y ^ fy) | (k[j].z ^ fz) | (k[j].w ^ fw);  // positions whose symbol is not the first one's
        diff |= d & rm;
        cntb += __popc(~d & below);
      }
      const u32 c = fw ? 4u : ((fy & 1u) | (fz & 2u));  // planes -> rank: A = p0 only, C = p1 only, G = both, T = p2
      const u32 cq = (qcomp && c) ? 5u - c : c;
      const u32 cfirst = gshfl(cq, first);
      if (cfirst != 0 && gballot(mine && (diff != 0 || cq != cfirst)) == 0) {
        nocc += 2u * pop(alive);
        if (mine) {
          const bool hb0 = ((c - 1u) & 1u) != 0, hb1 = ((c - 1u) & 2u) != 0;
          const u32 hdr = hb1 ? (hb0 ? k[3].x : k[2].x) : (hb0 ? k[1].x : k[0].x);
          P lbp = (P)(hdr + cntb);
          if (WIDE) lbp += (P)ix.super[(p0 >> SIGAX_SUPER_SHIFT) * 4 + (c - 1u)];
          const P size = e.c1hi - e.c1lo;
          e.c1lo = (P)tb.C[ix.which][c] + lbp;
          e.c1hi = e.c1lo + size;
        }
        *newAlive = alive;
        FXP(2);
        return RD_UPDATED;
      }
      if (BR && inreg && gballot(mine && (diff != 0 || c == 0u)) == 0) {
        P v = 0;
        if (mine) {
          const bool hb0 = ((c - 1u) & 1u) != 0, hb1 = ((c - 1u) & 2u) != 0;
          const u32 hdr = hb1 ? (hb0 ? k[3].x : k[2].x) : (hb0 ? k[1].x : k[0].x);
          v = (P)tb.C[ix.which][c] + (P)(hdr + cntb);
          if (WIDE) v += (P)ix.super[(p0 >> SIGAX_SUPER_SHIFT) * 4 + (c - 1u)];
        }
        FXP(5);
        return branch_inreg(e, alive, mine, cq, v);
      }
    }
    u32 lom[4], pa = 0, pc = 0, pg = 0, pt = 0, pd = 0;
    const u32 topLen = gshfl(e.len, first);
    const bool isTop = mine && e.len == topLen;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int t0 = r0 - 32 * j, t1 = r1 - 32 * j;
      u32 m0 = t0 >= 32 ? 0xFFFFFFFFu : (t0 <= 0 ? 0u : ((1u << t0) - 1u));
      u32 m1 = t1 >= 32 ? 0xFFFFFFFFu : (t1 <= 0 ? 0u : ((1u << t1) - 1u));
      u32 rm = m1 & ~m0;
      lom[j] = m0;
      u32 y = k[j].y, z = k[j].z, w = k[j].w;
      pa |= y & ~z & rm;
      pc |= z & ~y & rm;
      pg |= y & z & rm;
      pt |= w & rm;
      pd |= ~(y | z | w) & rm;
    }
    // OverlapBlock::ext > 0 per symbol, complemented for QUERYCOMP blocks (overlap_builder.cpp:181-187)
    const bool x0 = mine && pd != 0;
    const bool xa = mine && (qcomp ? pt : pa) != 0;
    const bool xc = mine && (qcomp ? pg : pc) != 0;
    const bool xg = mine && (qcomp ? pc : pg) != 0;
    const bool xt = mine && (qcomp ? pa : pt) != 0;
    if (gballot(isTop && x0)) {
      // the top-level block has ended (:747-766): capped.updateR('$') needs Occ('$') at both ends of the range
      const u64 topMask = gballot(isTop);
      u64 bad = gballot(isTop && !x0);
      u64 emitMask = topMask;
      if (bad) emitMask &= (1ull << ffs0(bad)) - 1ull;
      nocc += 2u * pop(topMask);
      u32 ne = pop(emitMask);
      if (nout + ne > OUTCAP) return RD_BAIL;
      if ((emitMask >> gl) & 1ull) {
        u32 below = k[0].x + k[1].x + k[2].x + k[3].x, nd = 0;  // A+C+G+T before the granule, then inside up to p0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int t1 = r1 - 32 * j;
          u32 m1 = t1 >= 32 ? 0xFFFFFFFFu : (t1 <= 0 ? 0u : ((1u << t1) - 1u));
          u32 y = k[j].y, z = k[j].z, w = k[j].w;
          below += __popc((y | z | w) & lom[j]);
          nd += __popc(~(y | z | w) & m1 & ~lom[j]);
        }
        P acgt = (P)below;
        if (WIDE) {
          const u64* sb = ix.super + (p0 >> SIGAX_SUPER_SHIFT) * 4;
          acgt += (P)(sb[0] + sb[1] + sb[2] + sb[3]);
        }
        const P ld = (P)p0 - acgt;  // Occ('$', lower - 1); C['$'] = 0
        E br = e;
        br.c0hi = br.c0lo + (P)nd - 1;
        br.c1lo = ld;
        br.c1hi = ld + (P)nd - 1;
        out_put(nout + pop(emitMask & glt), br);
      }
      nout += ne;
      if (bad) {
        xerror = true;
        return RD_XERROR;
      }
      FXP(3);
      return RD_ENDED;
    }
    const u64 any0 = gballot(x0), any1 = gballot(xa), any2 = gballot(xc), any3 = gballot(xg), any4 = gballot(xt);
    const u32 nz = (any0 != 0) + (any1 != 0) + (any2 != 0) + (any3 != 0) + (any4 != 0);
    if (nz != 1 || any0) {
      FXP(5);
      if (LEAN) return RD_BAIL;
      return round(e, alive, newAlive);
    }
    FXP(3);
    nocc += 2u * pop(alive);
    const u32 c = any1 ? 1u : any2 ? 2u : any3 ? 3u : 4u;
    const u32 b = qcomp ? 5u - c : c;
    if (mine) {
      u32 lb = b == 1 ? k[0].x : b == 2 ? k[1].x : b == 3 ? k[2].x : k[3].x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32 y = k[j].y, z = k[j].z, w = k[j].w;
        u32 bits = b == 1 ? (y & ~z) : b == 2 ? (z & ~y) : b == 3 ? (y & z) : w;
        lb += __popc(bits & lom[j]);
      }
      P lbp = (P)lb;
      if (WIDE) lbp += (P)ix.super[(p0 >> SIGAX_SUPER_SHIFT) * 4 + (b - 1)];
      const P size = e.c1hi - e.c1lo;
      e.c1lo = (P)tb.C[ix.which][b] + lbp;
      e.c1hi = e.c1lo + size;
    }
    *newAlive = alive;
    return RD_UPDATED;
  }

  // One extension round of a group of single-row blocks from the reads' text: the symbol that follows a block is the
  // next one on its path, the update is a step along it (nothing to compute: capped[0] keeps still, capped[1] is only
  // needed at the end, where it is the stretch's rank among the '$' rows).  Same cases and return codes as round_fast().
  __device__ int round_text(E& e, u64 alive, u64* newAlive) {
    const u32 OUTCAP = kOutCap;
    const u32 NSLOT = kNSlot;
    const bool mine = (alive >> gl) & 1ull;
    const u32 first = ffs0(alive);
    FXP(0);
    if (gshfl(tk, first) >= (u32)SIGAX_TEXT_WINDOW) {  // the group has used its window (all its lanes step together)
      if (mine) {
        tw = text_window(find_of(e.src) < 2 ? A.rev : A.fwd, tld, ttt);
        tk = 0;
      }
      sec_add(pop(alive));
    }
    const u32 c = ttt ? 1u + (u32)(tw >> 62) : 0u;  // the read's bases hold no rank 0; past its first base everything is
    const bool qcomp = (af_of(e.src) & 4u) != 0;
    const u32 cq = (qcomp && c) ? 5u - c : c;
    const bool x0 = mine && c == 0u;
    const u32 topLen = gshfl(e.len, first);
    const bool isTop = mine && e.len == topLen;
    if (gballot(isTop && x0)) {
      // the top-level block has ended (:747-766)
      const u64 topMask = gballot(isTop);
      const u64 bad = gballot(isTop && !x0);
      u64 emitMask = topMask;
      if (bad) emitMask &= (1ull << ffs0(bad)) - 1ull;
      nocc += 2u * pop(topMask);
      const u32 ne = pop(emitMask);
      if (nout + ne > OUTCAP) return RD_BAIL;
      if ((emitMask >> gl) & 1ull) {
        E br = e;
        br.c0hi = br.c0lo;
        br.c1lo = (P)tld;
        br.c1hi = (P)tld;
        out_put(nout + pop(emitMask & glt), br);
      }
      nout += ne;
      if (bad) {
        xerror = true;
        return RD_XERROR;
      }
      return RD_ENDED;
    }
    if (gballot(x0)) return RD_BAIL;  // '$' below the top level: the generic round's business
    const u32 cfirst = gshfl(cq, first);
    if (gballot(mine && cq != cfirst) == 0) {
      nocc += 2u * pop(alive);
      if (mine) { tw <<= 2; ++tk; --ttt; }
      *newAlive = alive;
      return RD_UPDATED;
    }
    if (!BR) return RD_BAIL;
    // a branch: the lanes part by their next symbol, in rank order A, C, G, T (:781-787)
    const u64 m1 = gballot(mine && cq == 1u), m2 = gballot(mine && cq == 2u), m3 = gballot(mine && cq == 3u), m4 = gballot(mine && cq == 4u);
    const u32 nb = (m1 != 0) + (m2 != 0) + (m3 != 0) + (m4 != 0);
    if (nslot + nb > NSLOT || ni + nb > NSLOT) return RD_BAIL;
    nocc += 2u * pop(alive);
    if (mine) { tw <<= 2; ++tk; --ttt; }
    FXP(5);
#pragma unroll
    for (u32 sy = 1; sy <= 4; ++sy) {
      const u64 m = sy == 1 ? m1 : sy == 2 ? m2 : sy == 3 ? m3 : m4;
      if (!m) continue;
      const u32 ns = nslot++;
      if (gl == ns) gAlive = m;
      if (gl == ni) gI = ns;
      ++ni;
      if (mine && cq == sy && (m & (m - 1ull)) == 0) {  // alone from here on: a countdown group (extract())
        e.c1lo = (P)tld;
        e.c1hi = (P)ttt;
        e.len |= FX_COUNTDOWN;
      }
    }
    return RD_BRANCHED;
  }

  // extract() over the n entries held one per group lane, already sorted by length descending.  Returns false when
  // the item has to be redone by a wider kernel.
  __device__ bool extract(E e, u32 n) {
    if (n == 0) return true;
    const u32 NSLOT = kNSlot;
    nslot = 1;
    ni = 0;
    gAlive = 0;
    gD = 0;
    gI = 0;
    inreg = gballot(gl < n && e.c1hi != e.c1lo) == 0;  // every block a single row (ranges never grow)
    if (TEXT) {
      if (!inreg) return false;
      tw = 0;
      tk = tld = ttt = 0;
      {
        // the entry's own symbols serve the first rounds (a window that is used up after that many); without them the
        // first text window is loaded at once
        const FmStrand& xs = find_of(e.src) < 2 ? A.rev : A.fwd;
        if (xs.xmap != nullptr) {
          // direct: the block names its target (capped[0] = its rank in the other strand's '$' rows) and its length says
          // where in the target the path starts (fm_layout.h); capped[0].lower is below n_strings by construction
          if (gl < n) {
            const u64 m = xs.xmap[(u64)e.c0lo < A.n_map ? (u64)e.c0lo : 0ull];
            tld = (u32)m;
            const u32 tl = (u32)(m >> 32), bl = e.len;
            ttt = tl > bl ? tl - bl : 0u;
            tw = text_window(xs, tld, ttt);
          }
          tk = 0;
          sec_add(2u * n);
        } else {
          const u32 K = (xs.sa_bits - xs.ld_bits - xs.t_bits) >> 1;
          if (gl < n) {
            const u64 syms = row_lookup(xs, (u64)e.c1lo, tld, ttt);
            tw = K ? syms << (64u - 2u * K) : text_window(xs, tld, ttt);
          }
          tk = K ? (u32)SIGAX_TEXT_WINDOW - K : 0u;
          sec_add(K ? n : 2u * n);
        }
      }
    }
#ifdef SIGAX_FX_PROFILE
    dbg_round = 0;
#endif
    // Phase 1: a single group (the usual case: every overlapping read agrees on the next base).  With one group the
    // stride-2 ring walk of :728-802 visits it over and over, so this is a plain loop with the group in registers.
    {
      u64 alive = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
      if (TEXT) {
        // All the rounds up to the one that ends the top-level block, AT ONCE: the top-level block (the first lane) ends after
        // R = its remaining symbols; until then a round only asks whether every block's next symbol (complemented for
        // QUERYCOMP blocks: rank - 1 -> 3 - (rank - 1), the bitwise NOT of its two bits) is the same non-'$' one -- i.e.
        // whether the next R symbols of every lane's window equal the first lane's.  One XOR and a shift instead of R rounds of
        // shuffles and ballots (BASELINE configs[1]: R is 5 on average; the rounds were a quarter of this kernel's
        // instructions).  Counted as the R rounds it replaces.  Anything else -- a lane whose read ends first, a disagreement
        // (a branch), top-level blocks ending at different rounds, a window that runs out first -- is left to the rounds below.
        const bool mine = (alive >> gl) & 1ull;
        const u32 first = ffs0(alive);
        const u32 R = gshfl(ttt, first);
        if (R > 0u && R <= (u32)SIGAX_TEXT_WINDOW) {
          const bool qc = (af_of(e.src) & 4u) != 0;
          const u64 X = qc ? ~tw : tw;
          const u64 Xf = gshfl(X, first);
          const u32 topLen = gshfl(e.len, first);
          const bool isTop = mine && e.len == topLen;
          const bool okl = !mine || (ttt >= R && (u32)SIGAX_TEXT_WINDOW - tk >= R && ((X ^ Xf) >> (64u - 2u * R)) == 0ull && (!isTop || ttt == R));
          if (gballot(!okl) == 0) {
            if (mine) {
              tw <<= 2u * R;
              tk += R;
              ttt -= R;
            }
            nocc += 2u * pop(alive) * R;
          }
        }
      }
      u32 guard = 0;
      while (alive) {
        u64 na = 0;
        int st = TEXT ? round_text(e, alive, &na) : round_fast(e, alive, &na, true);
        if (st == RD_BAIL) return false;
        if (st == RD_XERROR) return true;
        if (st != RD_UPDATED) break;
        alive = na;
        if (++guard > (1u << 20)) return false;
      }
    }
    if (ni == 0) return true;
    if (!BR || (LEAN && !inreg)) return false;
    // Phase 2: the group branched.  General form: groups in a list, walked as the reference's loop walks it.
    u32 cur = 0xFFFFFFFFu, ng = 0;
    for (u32 i = 0; i < ni; ++i) {
      u32 v = gshfl(gI, i);
      if (gl == ng) gD = v;
      ++ng;
    }
    u32 guard = 0;
    while (ng > 0) {
      ni = 0;
      u32 p = 0;
      while (p != ng) {
        if (TEXT && p == 0u && ni == 0u) {
          // Many turns of the ring AT ONCE (one group, or the head of two, is the same walk: place 0 again and again; a countdown
          // group that is alone there is emitted without counting down, and the count comes out the same: 2 t here + 2 there).
          // While nothing is erased and nothing branches, the stride-2 walk visits the groups at
          // the EVEN places of the list, in order, again and again (with an even number of groups a pass ends at the list's end
          // and the next starts at place 0; with an odd number it wraps to place 0 inside the pass), and the groups at the odd
          // places are not touched at all.  A visit to a countdown group with t rounds to go takes one off; a visit to a group
          // of blocks whose next symbols agree is a plain round.  So until the first group at an even place has something to
          // decide -- a countdown at 0, a top-level block at its read's end, a block whose next symbol differs from its
          // group's first, a text window used up -- every turn is the same, and d turns cost what one costs: d = the smallest
          // number of such uneventful visits any even-place group has left.  Reads with sequencing errors spend their time
          // here (a substitution in an overlapping read leaves a countdown group of ~100 rounds behind: at 1 % substitutions
          // the ring took 27 of filter/extract's 31 ms per million reads).  Counted as the visits it replaces.
          bool ing = false, even = false, hole = false;
          u32 myfirst = 0;
          for (u32 q = 0; q < ng; ++q) {
            const u32 sl = gshfl(gD, q);
            const u64 al = gshfl(gAlive, sl);
            if (al == 0ull && (q & 1u) == 0u) hole = true;  // an empty group is erased when it is visited: an event
            if ((al >> gl) & 1ull) {
              ing = true;
              even = (q & 1u) == 0u;
              myfirst = ffs0(al);
            }
          }
          const bool cd = (e.len & FX_COUNTDOWN) != 0u;
          const u64 X = (af_of(e.src) & 4u) ? ~tw : tw;
          const u64 Xf = gshfl(X, myfirst);
          const u32 tf = gshfl(ttt, myfirst);
          u32 d = 0xFFFFFFFFu;
          if (ing && even) {
            if (cd) {
              d = (u32)e.c1hi;
            } else {
              const u64 x = X ^ Xf;
              const u32 a = x ? (u32)__clzll((long long)x) >> 1 : 32u;
              const u32 avail = (u32)SIGAX_TEXT_WINDOW - (tk < (u32)SIGAX_TEXT_WINDOW ? tk : (u32)SIGAX_TEXT_WINDOW);
              d = min(min(a, ttt), min(avail, tf));
            }
          }
#pragma unroll
          for (u32 off = (u32)W / 2u; off > 0u; off >>= 1) d = min(d, gshfl(d, gl ^ off));
          if (!hole && d != 0xFFFFFFFFu && d >= 1u) {
            const bool adv = ing && even;
            if (adv) {
              if (cd) {
                e.c1hi -= (P)d;
              } else {
                tw <<= 2u * d;
                tk += d;
                ttt -= d;
              }
            }
            nocc += 2u * d * pop(gballot(adv));
          }
        }
        const u32 slot = gshfl(gD, p);
        const u64 alive = gshfl(gAlive, slot);
        if (slot != cur && !inreg) {
          if (cur != 0xFFFFFFFFu) {
            const u64 was = gshfl(gAlive, cur);
            if ((was >> gl) & 1ull) pool_put(wpool + cur * 64 + lane, e);
          }
          if ((alive >> gl) & 1ull) pool_get(e, wpool + slot * 64 + lane);
          cur = slot;
        }
        bool eraseGroup = true;
        const u32 owner = alive ? ffs0(alive) : 0u;
        // The loop steps two places on the ring [g0 .. g(ng-1), end] (the body's ++i and the header's): with one group,
        // or at the head of two, the group at hand is visited again and again, and nothing else, until it is erased -- a
        // pass that visits g0 of two ends right after it, and with no incomings to splice the next one starts there again.  Such a group may take two rounds
        // at once, and a countdown group is emitted without counting down.
        const bool alone = ng == 1 || (ng == 2 && p == 0 && ni == 0);
        if (alive && inreg && (gshfl(e.len, owner) & FX_COUNTDOWN)) {
          // a countdown group (round()): t more rounds of one lookup pair each, then the round that finds '$'.  Alone in
          // the ring it runs to its end before anything else happens (the pass does not end while a group stays), so it
          // is emitted at once.
          const u32 t = gshfl((u32)e.c1hi, owner);
          if (t == 0 || alone) {
            nocc += 2u * (t + 1u);
            if (nout + 1 > kOutCap) return false;
            if (gl == owner) {
              E br = e;
              br.c0hi = br.c0lo;  // updateR('$') of a single row: one '$' row, capped[0] stays
              br.c1hi = br.c1lo;
              out_put(nout, br);
            }
            nout += 1;
          } else {
            if (gl == owner) e.c1hi -= 1;
            nocc += 2u;
            eraseGroup = false;
          }
        } else if (alive) {
          u64 na = 0;
          int st = TEXT ? round_text(e, alive, &na) : round_fast(e, alive, &na, alone);  // two rounds at once, as in phase 1
          if (st == RD_BAIL) return false;
          if (st == RD_XERROR) return true;
          if (st == RD_UPDATED) {
            if (gl == slot) gAlive = na;
            eraseGroup = false;
          }
        }
        // body `i = erase(i)` / `++i`, then the loop header's `++i` on the ring [g0..g(k-1), end]
        if (eraseGroup) {
          u32 nxt = gshfl(gD, gl + 1 < (u32)W ? gl + 1 : gl);
          if (gl >= p) gD = nxt;
          --ng;
        } else {
          p = p + 1 > ng ? 0 : p + 1;
        }
        p = p + 1 > ng ? 0 : p + 1;
        if (++guard > (1u << 20)) return false;
      }
      if (ng + ni > NSLOT) return false;
      for (u32 i = 0; i < ni; ++i) {
        u32 v = gshfl(gI, i);
        if (gl == ng) gD = v;
        ++ng;
      }
    }
    return true;
  }

  // one (read, side) item on this lane group; returns false when it must be redone by a wider kernel
  __device__ bool body(u32 r, u32 sd) {
    const u32 OUTCAP = kOutCap;
    nout = 0;
    nocc = 0;
    xerror = false;
    toowide = false;
    slots = reinterpret_cast<const Cand<WIDE>*>(A.arena) + (u64)r * 4 * A.cap;
    const u32 L = (u32)(A.offs[r + 1] - A.offs[r]);
    u32 cc[4];
    {
      uint4 c4 = reinterpret_cast<const uint4*>(A.chain_cnt)[r];
      cc[0] = c4.x; cc[1] = c4.y; cc[2] = c4.z; cc[3] = c4.w;
      ccor = c4.x | c4.y | c4.z | c4.w;  // account() reads the finds' substring flags off it
    }
    if (sd == 0) {  // containfwd, containrev first (:1161-1162)
      u32 ccl = gl == 0 ? cc[0] : gl == 1 ? cc[1] : gl == 2 ? cc[2] : cc[3];
      bool has = gl < 4 && (ccl & SIGAX_CC_CONTAIN);
      u64 m = gballot(has);
      if (has) {
        E e;
        load_block(e, (gl << 30) | (gl * A.cap + (A.cap - 1)));
        out_put(pop(m & glt), e);
      }
      nout = pop(m);
    }
    const u32 chA = sd == 0 ? 0u : 1u, chB = sd == 0 ? 3u : 2u;
    const u32 nA = (sd == 0 ? cc[0] : cc[1]) & SIGAX_CC_COUNT_MASK, nB = (sd == 0 ? cc[3] : cc[2]) & SIGAX_CC_COUNT_MASK;
    const u32 c0 = (cc[0] & SIGAX_CC_CONTAIN) ? 1u : 0u, c1 = (cc[1] & SIGAX_CC_CONTAIN) ? 1u : 0u;
    const u32 c2 = (cc[2] & SIGAX_CC_CONTAIN) ? 1u : 0u, c3 = (cc[3] & SIGAX_CC_CONTAIN) ? 1u : 0u;
    // list X = find A's blocks + containfwd {0,1}; list Y = find B's blocks + containrev {2,3} (:1137-1140)
    const u32 nX = nA + c0 + c1, nY = nB + c2 + c3, T = nX + nY;
    if (T > (u32)W) {
      if (W == 64 && !LEAN && big != nullptr && T <= FX_BIGCAP && A.irreducible) return body_big(L, chA, chB, nA, nB, c0, c2, nX, T);
      toowide = true;
      return false;
    }
    if (T == 0) return true;
    if (LEAN && !A.irreducible) return false;
    FXP(6);
    if (T <= 16) FXP(7);
    const bool active = gl < T;
    const u32 list = gl >= nX ? 1u : 0u;
    const u32 k = list ? gl - nX : gl;
    u32 src = 0;
    if (active) {
      u32 ch;
      if (!list) ch = k < nA ? chA : ((k == nA && c0) ? 0u : 1u);
      else ch = k < nB ? chB : ((k == nB && c2) ? 2u : 3u);
      const u32 inchain = (!list ? k < nA : k < nB) ? k : A.cap - 1;
      src = (ch << 30) | (ch * A.cap + inchain);
    }
    E e;
    e.c0lo = e.c0hi = e.c1lo = e.c1hi = 0; e.src = 0; e.len = 0;
    if (active) load_block(e, src);
    const bool member = active && e.len != L;  // ContainmentBlockRemover (:1094-1111)
    const u32 nm = pop(gballot(member));
    // SubMaximalBlockFilter::filter (:930-953) sorts by capped[0].lower and resolves adjacent intersecting blocks.
    // With the blocks sorted by lower bound, SOME pair intersects iff some ADJACENT pair does (if i precedes j and
    // lower_j <= upper_i, the successor k of i has lower_k <= lower_j <= upper_i), so an any-pair test inside each
    // list decides exactly whether resolve() is needed; if so the item goes to the general kernel.
    u32 rank = 0;  // stable rank by capped[0].lower inside the own list (only the exhaustive output order needs it)
    bool inter = false;
    if (LEAN || A.irreducible) {
      // only the yes/no is needed: each lane walks its own list's bounds, parked in LDS
      if (active) { sh.e0[gb + gl] = e.c0lo; sh.e1[gb + gl] = e.c0hi; }
      wave_lds_sync();
      const u32 jend = active ? (list ? T : nX) : 0u;
      for (u32 j = list ? nX : 0u; j < jend; ++j) {
        const P loj = sh.e0[gb + j], hij = sh.e1[gb + j];
        inter |= (j != gl) & !(e.c0lo > hij || loj > e.c0hi);  // coord.h:37-40
      }
      wave_lds_sync();
    } else
    for (u32 j = 0; j < T; ++j) {
      P loj = gshfl(e.c0lo, j), hij = gshfl(e.c0hi, j);
      u32 lj = j >= nX ? 1u : 0u, kj = lj ? j - nX : j;
      bool same = active && lj == list;
      if (!A.irreducible && same && (loj < e.c0lo || (loj == e.c0lo && kj < k))) ++rank;
      inter |= same & (j != gl) & !(e.c0lo > hij || loj > e.c0hi);  // coord.h:37-40
    }
    if (gballot(inter)) {
      FXP(11);
      return false;
    }
    if (LEAN || A.irreducible) {
      // X += Y; stable sort by length descending (:715-716,1169), ties keep list X first.  Both finds pushed their
      // blocks in increasing length, so the position is a merge rank: blocks after me in my own list, plus the other
      // list's blocks that are longer (or, seen from Y, as long): a binary search over the other list's lanes.
      const u32 obase = list ? 0u : nX, on = list ? nA : nB;
      u32 lo = 0, hi = on;
      for (u32 step = 0; step < 7; ++step) {
        u32 mid = (lo + hi) >> 1;
        u32 lenm = gshfl(e.len, obase + (mid < on ? mid : 0));
        bool right = list ? (lenm < e.len) : (lenm <= e.len);
        if (lo < hi) {
          if (right) lo = mid + 1; else hi = mid;
        }
      }
      const u32 pos = ((list ? nB : nA) - 1u - k) + (on - lo);
      if (member) {
        sh.e0[gb + pos] = e.c0lo; sh.e1[gb + pos] = e.c0hi; sh.e2[gb + pos] = e.c1lo; sh.e3[gb + pos] = e.c1hi;
        sh.esrc[gb + pos] = e.src; sh.elen[gb + pos] = e.len;
      }
      wave_lds_sync();
      E g;
      g.c0lo = g.c0hi = g.c1lo = g.c1hi = 0; g.src = 0; g.len = 0;
      if (gl < nm) {
        g.c0lo = sh.e0[lane]; g.c0hi = sh.e1[lane]; g.c1lo = sh.e2[lane]; g.c1hi = sh.e3[lane];
        g.src = sh.esrc[lane]; g.len = sh.elen[lane];
      }
      wave_lds_sync();
      return extract(g, nm);
    }
    // exhaustive: the filtered lists go out as they are, X then Y, each in capped[0].lower order (:1175-1178)
    u32 pos = 0;
    for (u32 j = 0; j < T; ++j) {
      u32 lenj = gshfl(e.len, j), rj = gshfl(rank, j);
      u32 lj = j >= nX ? 1u : 0u;
      if (lenj != L && (lj < list || (lj == list && rj < rank))) ++pos;
    }
    if (nout + nm > OUTCAP) return false;
    if (member) out_put(nout + pos, e);
    nout += nm;
    return true;
  }

  // An item of 65..256 blocks on the 64-lane group, four blocks per lane (deep coverage: at 54x and 250 bp a side holds
  // 44 blocks on average and a few reads in a thousand pass 64).  Same prologue as body(); the extraction handles the
  // single-group case only -- every block's range inside one granule, one common next symbol per round until the
  // top-level blocks end -- and hands anything else (intersecting blocks, a branch, a range across granules) to the
  // general kernel by returning false.  Without this such items ran on ONE lane each and took longer than the rest of
  // the batch together.
  __device__ bool body_big(u32 L, u32 chA, u32 chB, u32 nA, u32 nB, u32 c0, u32 c2, u32 nX, u32 T) {
    BigSh<WIDE>& bs = *big;
    E e[4];
    bool act[4], member[4];
    u32 nm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 i = lane + 64u * k;
      act[k] = i < T;
      e[k].c0lo = e[k].c0hi = e[k].c1lo = e[k].c1hi = 0; e[k].src = 0; e[k].len = 0;
      if (act[k]) {
        const u32 list = i >= nX ? 1u : 0u, kk = list ? i - nX : i;
        u32 ch;
        if (!list) ch = kk < nA ? chA : ((kk == nA && c0) ? 0u : 1u);
        else ch = kk < nB ? chB : ((kk == nB && c2) ? 2u : 3u);
        const u32 inchain = (!list ? kk < nA : kk < nB) ? kk : A.cap - 1;
        load_block(e[k], (ch << 30) | (ch * A.cap + inchain));
        bs.b0[i] = e[k].c0lo;
        bs.b1[i] = e[k].c0hi;
        bs.blen[i] = e[k].len;
      }
      member[k] = act[k] && e[k].len != L;  // ContainmentBlockRemover (:1094-1111)
      nm += pop(__ballot(member[k]));
    }
    wave_lds_sync();
    // SubMaximalBlockFilter: any intersecting pair inside a list needs resolve() (see body())
    bool inter = false;
    u32 posn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 i = lane + 64u * k;
      posn[k] = 0;
      if (act[k]) {
        const u32 list = i >= nX ? 1u : 0u, kk = list ? i - nX : i;
        const u32 jb = list ? nX : 0u, je = list ? T : nX;
        for (u32 j = jb; j < je; ++j) {
          const P loj = bs.b0[j], hij = bs.b1[j];
          inter |= (j != i) & !(e[k].c0lo > hij || loj > e[k].c0hi);  // coord.h:37-40
        }
        // position after X += Y and the stable sort by length descending (:715-716,1169): a merge rank, see body()
        const u32 obase = list ? 0u : nX, on = list ? nA : nB;
        u32 lo = 0, hi = on;
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          const u32 lenm = bs.blen[obase + mid];
          const bool right = list ? (lenm < e[k].len) : (lenm <= e[k].len);
          if (right) lo = mid + 1; else hi = mid;
        }
        posn[k] = ((list ? nB : nA) - 1u - kk) + (on - lo);
      }
    }
    if (__ballot(inter)) return false;
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (member[k]) {
        const u32 q = posn[k];
        bs.b0[q] = e[k].c0lo; bs.b1[q] = e[k].c0hi; bs.b2[q] = e[k].c1lo; bs.b3[q] = e[k].c1hi;
        bs.bsrc[q] = e[k].src; bs.blen[q] = e[k].len;
      }
    wave_lds_sync();
    bool mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 i = lane + 64u * k;
      mine[k] = i < nm;
      e[k].c0lo = e[k].c0hi = e[k].c1lo = e[k].c1hi = 0; e[k].src = 0; e[k].len = 0;
      if (mine[k]) {
        e[k].c0lo = bs.b0[i]; e[k].c0hi = bs.b1[i]; e[k].c1lo = bs.b2[i]; e[k].c1hi = bs.b3[i];
        e[k].src = bs.bsrc[i]; e[k].len = bs.blen[i];
      }
    }
    wave_lds_sync();
    if (nm == 0) return true;
    // IrreducibleBlockListExtractor::extract (:711-809), one group.  Block 0 (lane 0, k = 0) is the longest.
    const u32 topLen = __builtin_amdgcn_readfirstlane(e[0].len);
    u32 ntop = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) ntop += pop(__ballot(mine[k] && e[k].len == topLen));
    for (u32 guard = 0; guard < (1u << 20); ++guard) {
      bool odd = false, anyEnd = false;  // a block outside the simple case; a top-level block followed by '$'
      u32 cq[4];
      P newlo[4];
      bool x0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cq[k] = 0; newlo[k] = 0; x0[k] = false;
        if (!mine[k]) continue;
        const FmRef ix = ext_index(e[k].src);
        const u64 p0 = (u64)e[k].c1lo, p1 = (u64)e[k].c1hi + 1ull;
        const u64 g0 = p0 >> 7;
        if (!(p1 > p0 && ((p1 - 1) >> 7) == g0 && p1 <= ix.n)) {
          odd = true;
          continue;
        }
        const uint4* q = ix.g + g0 * 4;
        uint4 kq[4];
        kq[0] = q[0]; kq[1] = q[1]; kq[2] = q[2]; kq[3] = q[3];
        const int r0 = (int)(p0 & 127u), r1 = (int)(p1 - (g0 << 7));
        const u32 j0 = (u32)r0 >> 5, bit0 = (u32)r0 & 31u;
        const u32 ys = j0 == 0 ? kq[0].y : j0 == 1 ? kq[1].y : j0 == 2 ? kq[2].y : kq[3].y;
        const u32 zs = j0 == 0 ? kq[0].z : j0 == 1 ? kq[1].z : j0 == 2 ? kq[2].z : kq[3].z;
        const u32 ws = j0 == 0 ? kq[0].w : j0 == 1 ? kq[1].w : j0 == 2 ? kq[2].w : kq[3].w;
        const u32 fy = 0u - ((ys >> bit0) & 1u), fz = 0u - ((zs >> bit0) & 1u), fw = 0u - ((ws >> bit0) & 1u);
        u32 diff = 0, cntb = 0, dol = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int lo = min(max(r0 - 32 * j, 0), 32), hi = min(max(r1 - 32 * j, 0), 32), wd = hi - lo;
          const u32 rm = wd == 32 ? 0xFFFFFFFFu : (((1u << (wd & 31)) - 1u) << (lo & 31));
          const u32 below = lo == 32 ? 0xFFFFFFFFu : ((1u << (lo & 31)) - 1u);
          const u32 d = (kq[j].y ^ fy) | (kq[j].z ^ fz) | (kq[j].w ^ fw);
          diff |= d & rm;
          cntb += __popc(~d & below);
          dol |= ~(kq[j].y | kq[j].z | kq[j].w) & rm;
        }
        const u32 c = fw ? 4u : ((fy & 1u) | (fz & 2u));
        const bool qcomp = (af_of(e[k].src) & 4u) != 0;
        cq[k] = (qcomp && c) ? 5u - c : c;
        x0[k] = dol != 0;
        if (diff != 0) odd = true;  // more than one symbol follows this block
        if (c != 0) {
          const u32 hdr = c == 1 ? kq[0].x : c == 2 ? kq[1].x : c == 3 ? kq[2].x : kq[3].x;
          P lbp = (P)(hdr + cntb);
          if (WIDE) lbp += (P)ix.super[(p0 >> SIGAX_SUPER_SHIFT) * 4 + (c - 1u)];
          newlo[k] = (P)tb.C[ix.which][c] + lbp;
        }
        if (e[k].len == topLen && x0[k]) anyEnd = true;
      }
      sec_add(nm);
      if (__ballot(anyEnd)) {
        // the top-level blocks have ended (:747-766): they are blocks 0 .. ntop-1; the first one without '$' is the
        // "substring read found" error and ends the emission
        u32 fb = ntop;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u64 m = __ballot(mine[k] && e[k].len == topLen && !x0[k]);
          if (m) fb = min(fb, ffs0(m) + 64u * k);
        }
        nocc += 2u * ntop;
        if (nout + fb > (u32)FX_OUTCAP) return false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 i = lane + 64u * k;
          if (mine[k] && i < fb) {
            // capped.updateR('$'): Occ('$') at both ends of the range = position minus the A,C,G,T before it
            const FmRef ix = ext_index(e[k].src);
            P l[5], u[5];
            fm_rank5p_pair<WIDE>(ix, e[k].c1lo, (P)(e[k].c1hi + 1), l, u);
            E br = e[k];
            apply_updateR(br, 0, ix.which, l, u);
            out_put(nout + i, br);
          }
        }
        nout += fb;
        if (fb < ntop) xerror = true;
        return true;
      }
      nocc += 2u * nm;
      const u32 cfirst = __builtin_amdgcn_readfirstlane(cq[0]);
      bool differ = odd;
#pragma unroll
      for (int k = 0; k < 4; ++k) differ = differ || (mine[k] && cq[k] != cfirst);
      if (cfirst == 0 || __ballot(differ)) return false;  // a branch, '$' below the top level, a range across granules
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (mine[k]) {
          const P size = e[k].c1hi - e[k].c1lo;
          e[k].c1lo = newlo[k];
          e[k].c1hi = newlo[k] + size;
        }
    }
    return false;
  }

  // Both lane groups of the wave together: run the items, then flush their blocks into the wave's chunk of the
  // unordered arena (one atomic per FX_FIN_CHUNK blocks).  `has` = this group has an item; returns done.
  __device__ bool run(bool has, u32 r, u32 sd) {
    bool done = false;
    if (has) done = body(r, sd);
    wave_lds_sync();
    const u32 mine_n = (has && done) ? nout : 0u;
    u32 tot = 0, before = 0;  // blocks of all the wave's groups; of the groups below this lane's
#pragma unroll
    for (u32 g = 0; g < 64u; g += (u32)W) {
      const u32 ng = __shfl((int)mine_n, (int)g, 64);
      before += g < gb ? ng : 0u;
      tot += ng;
    }
    if (tot == 0) return done;
    if (fin_cur + tot > fin_end) {
      u64 b0 = 0;
      if (lane == 0) b0 = atomicAdd(&A.dstat[DS_FIN_TOP], (u64)FX_FIN_CHUNK);
      fin_cur = readlane64(b0, 0);
      fin_end = fin_cur + FX_FIN_CHUNK;
    }
    const u64 base = fin_cur + before;
    fin_cur += tot;
    if (has && done) {
      if (gl == 0) A.item_base[2ull * r + sd] = base;
      if (gl < nout && base + gl < A.fin_cap) {
        // capped pair + where the candidate record is: the ordered scatter (a streaming kernel) fetches the raw pair,
        // length and flags from it, which keeps that memory round trip off this item's dependent chain
        const u32 src = sh.osrc[lane];
        ulonglong2* d = reinterpret_cast<ulonglong2*>(A.fin + base + gl);
        d[0] = make_ulonglong2(widen(sh.o0[lane]), widen(sh.o1[lane]));
        d[1] = make_ulonglong2(widen(sh.o2[lane]), widen(sh.o3[lane]));
        d[4] = make_ulonglong2(0ull, 0x8000000000000000ull | (u64)src);
      }
    }
    return done;
  }

  // per-item bookkeeping by the group's first lane
  __device__ void account(bool has, bool done, u64 item, u64& nocc_total, u64& nerr, u64& nsub) {
    if (!has || gl != 0 || !done) return;
    const u32 r = (u32)(item >> 1), sd = (u32)(item & 1);
    u32 word = nocc & OCC_SIDE_MASK;
    nocc_total += nocc;
    if (xerror) { word |= OCC_SIDE_ERR; ++nerr; }
    if (sd == 0) {
      u32 sub = ccor & SIGAX_CC_SUBSTRING;
      A.substring[r] = sub ? 1 : 0;
      if (sub) { word |= OCC_SIDE_SUB; ++nsub; }
    }
    A.fin_cnt[item] = nout;
    A.occ_side[item] = word;
  }
};

// W == 32: two (read, side) items per wave, one per half; items that do not fit (more than 32 blocks, branching beyond
// the half's slots, output beyond its share) are queued for the next launch (launch_filter_extract_fast), the last of
// which queues what it cannot finish for the general kernel.
// Which (read, side) items of the sub-batch have more blocks than a 32-lane group holds?  They skip the 32-lane launches:
// queued here for the 64-lane one, one atomic per wave of 64 items.  Same count as GFx::body(): the find's blocks plus the
// containment copies of the list's two chains.
__global__ __launch_bounds__(256) void k_fx_route(FxArgs A) {
  const u64 first = 2ull * A.read_begin, last = 2ull * A.read_end;
  const u64 item = first + (u64)blockIdx.x * 256 + threadIdx.x;
  bool wide = false;
  if (item < last) {
    const u32 r = (u32)(item >> 1), sd = (u32)(item & 1);
    const uint4 c4 = reinterpret_cast<const uint4*>(A.chain_cnt)[r];
    const u32 nA = (sd == 0 ? c4.x : c4.y) & SIGAX_CC_COUNT_MASK, nB = (sd == 0 ? c4.w : c4.z) & SIGAX_CC_COUNT_MASK;
    const u32 nc = ((c4.x & SIGAX_CC_CONTAIN) ? 1u : 0u) + ((c4.y & SIGAX_CC_CONTAIN) ? 1u : 0u) + ((c4.z & SIGAX_CC_CONTAIN) ? 1u : 0u) +
                   ((c4.w & SIGAX_CC_CONTAIN) ? 1u : 0u);
    wide = nA + nB + nc > 32u;
  }
  const u64 m = __ballot(wide);
  if (!m) return;
  const u32 lane = threadIdx.x & 63u;
  u64 b = 0;
  if (lane == 0) b = atomicAdd(A.q_wide_n, (u64)__popcll(m));
  b = readlane64(b, 0);
  if (wide) A.q_wide[b + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)item;
}

#ifndef SIGAX_FX_LEAN_WAVES
#define SIGAX_FX_LEAN_WAVES 4  // register budget of the lean launch as waves per SIMD (4: up to 128, it takes 89; 6: 80 with 6 spilled)
#endif
template <bool WIDE, int W, int LEAN = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE ? (LEAN ? 2 : 1) : (LEAN ? SIGAX_FX_LEAN_WAVES : 4)))) void k_filter_extract_fast(FxArgs A) {
  __shared__ FmTables tb;
  __shared__ Find2TablesT<WIDE> t2;
  __shared__ SideSh<WIDE> shm[4];
  const bool have2 = A.fwd.gran2 != nullptr && A.rev.gran2 != nullptr;
  if (have2) find2_tables_load<WIDE>(t2, A.fwd, A.rev);
  fm_tables_load(tb, A.fwd, A.rev);
  const u32 wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const u64 wave = (u64)blockIdx.x * 4 + wid, nwaves = (u64)gridDim.x * 4;
  if (lane == 0) shm[wid].nsec = 0;
#ifdef SIGAX_FX_PROFILE
  if (lane < 16) shm[wid].prof[lane] = 0;
  if (lane < 8) shm[wid].prof2[lane] = 0;
#endif
  wave_lds_sync();
  GFx<WIDE, W, LEAN> fx(A, tb, shm[wid], A.wpool + wave * FX_WPOOL, have2 ? &t2 : nullptr);
  if constexpr (W == 64 && LEAN == 0) {
    __shared__ BigSh<WIDE> bigsh[4];
    fx.big = &bigsh[wid];
  }
  u64 nocc_total = 0, nerr = 0, nsub = 0;
  // Items come from this sub-batch's read range (the first launch) or from the queue the launch before filled; what a
  // launch cannot finish goes to a later one's queue -- items with more blocks than the group has lanes to q_wide, the
  // rest to q_out -- and from the last launch to the general kernel (by read).  Queue entries are collected per wave in
  // LDS and appended 28 or more at a time: one atomic on the queue's counter per item serialises at ~6 ns each.
  // The strict lean launch is always the first one (range input) and appends with one atomic per item: its register
  // budget is what the error-free step time hangs on (89 VGPRs; 99 and scratch with the batching below: 9 % on the step),
  // and its atomics hide behind its other work.
  constexpr bool STRICT = LEAN == 1 || LEAN == 2 || LEAN == 5;
  const u64 first = 2ull * A.read_begin;
  if constexpr (STRICT) {
    // Items with more blocks than a 32-lane group holds were queued for the 64-lane launch by k_fx_route before this
    // launch: one atomic per such item on the queue's counter serialises at 6-16 ns, and at the C5 shape, where most items
    // are that wide, it alone took 12.8 ms of a 17.8 ms chain.  (Batching appends through LDS here costs the launch 10
    // VGPRs and scratch whichever way it is written; the few branching items' atomics hide behind the other work.)
    const u64 last = 2ull * A.read_end;
    for (u64 w0 = first + wave * 2; w0 < last; w0 += nwaves * 2) {
      const u64 item = w0 + (lane >> 5);
      const bool has = item < last;
      const bool done = fx.run(has, (u32)(item >> 1), (u32)(item & 1));
      if (has && !done && fx.gl == 0) {
        A.fin_cnt[item] = 0;
        A.occ_side[item] = 0;
        if (!fx.toowide) A.q_out[atomicAdd(A.q_out_n, 1ull)] = (u32)item;
      }
      fx.account(has, done, item, nocc_total, nerr, nsub);
    }
  } else {
    const u32* qin = A.q_in;
    const u64 nitems = qin ? *A.q_in_n : 2ull * A.read_end - first;
    const u64 per = 64u / (u32)W;  // items per wave
    __shared__ u32 qbuf[4][2][32];  // items a wave hands on, waiting to be appended to the later launches' queues
    u32 qn[2] = {0, 0};
    auto flush = [&](int which) {
      const u32 n = qn[which];
      if (n == 0) return;
      u64 b = 0;
      if (lane == 0) b = atomicAdd(which ? A.q_wide_n : A.q_out_n, (u64)n);
      b = readlane64(b, 0);
      if (lane < n) (which ? A.q_wide : A.q_out)[b + lane] = qbuf[wid][which][lane];
      wave_lds_sync();
      qn[which] = 0;
    };
    auto push = [&](int which, bool p, u32 it) {
      const u64 m = __ballot(p);
      if (!m) return;
      if (p) qbuf[wid][which][qn[which] + (u32)__popcll(m & ((1ull << lane) - 1ull))] = it;
      wave_lds_sync();
      qn[which] += (u32)__popcll(m);
      if (qn[which] >= 28u) flush(which);  // up to four more (16-lane groups) must still fit the 32 entries
    };
    for (u64 i0 = wave * per; i0 < nitems; i0 += nwaves * per) {
      const u64 idx = i0 + lane / (u32)W;
      const bool has = idx < nitems;
      const u64 item = !has ? 0ull : (qin ? (u64)qin[idx] : first + idx);
      const bool done = fx.run(has, (u32)(item >> 1), (u32)(item & 1));
      const bool bail = has && !done && fx.gl == 0;
      if (bail) {
        A.fin_cnt[item] = 0;
        A.occ_side[item] = 0;
      }
      const bool wide_q = A.q_wide != nullptr && fx.toowide;
      if (A.q_wide != nullptr) push(1, bail && wide_q, (u32)item);
      if (A.q_out != nullptr) push(0, bail && !wide_q, (u32)item);
      else if (bail && !wide_q) {
        const u32 r = (u32)(item >> 1);
        if (atomicExch(&A.slow_flag[r], 1u) == 0u) A.work_out[atomicAdd(A.slow_counter, 1ull)] = r;
      }
      fx.account(has, done, item, nocc_total, nerr, nsub);
    }
    flush(0);
    flush(1);
  }
  nocc_total = wave_sum(nocc_total); nerr = wave_sum(nerr); nsub = wave_sum(nsub);
  wave_lds_sync();
#ifdef SIGAX_FX_PROFILE
  if (lane < 16 && shm[wid].prof[lane]) atomicAdd(&A.dstat[DS_PROF_BASE + (W == 64 ? 16 : 0) + lane], (u64)shm[wid].prof[lane]);
  if (W == 32 && lane < 8 && shm[wid].prof2[lane]) atomicAdd(&A.dstat[DS_PROF_BASE + 24 + lane], (u64)shm[wid].prof2[lane]);
#endif
  if (lane == 0) {
    if (shm[wid].nsec) atomicAdd(&A.dstat[DS_SEC_EXTRACT], (u64)shm[wid].nsec);
    if (nocc_total) atomicAdd(&A.dstat[DS_OCC_EXTRACT], nocc_total);
    if (nerr) atomicAdd(&A.dstat[DS_EXTRACT_ERRORS], nerr);
    if (nsub) atomicAdd(&A.dstat[DS_SUBSTRING], nsub);
  }
}

// -------------------------------------------------------------------------------------------------------
// k_correct: KmerCorrector::process (src/correct_processor.cpp:81-229), one wave per read.
// Lanes are k-mer windows: every window's FMIndex::Interval::occurrences (src/fmindex.h:67-86) is an independent
// chain of k-1 dependent rank pairs, so a round is the same memory-bound gather as the block finder.  Only windows
// covering a corrected base are recounted (the reference's per-read kmerCache has the same effect).  Candidate
// corrections of up to eight unsolid bases are evaluated at once (2 covering k-mers x 4 bases each); the one the
// reference's left-to-right scan would take first is applied.
// -------------------------------------------------------------------------------------------------------
#define CORRECT_LMAX 1024
#ifndef SIGAX_CORRECT_WAVES
#define SIGAX_CORRECT_WAVES 6  // waves per SIMD the small form is held to at least: 80 registers, no scratch (4: 85 registers, five waves, 38.5 M reads/s; 6: 41.0 M)
#endif

// One wave's read in LDS.  LMAX = the longest read the instantiation takes: 10 KB per wave at 1024 is what holds the
// kernel to three workgroups per CU; reads of up to 512 bases run in a 5 KB form (four per CU with the registers held to
// 128), longer ones are left to a second launch of the 1024 form (launch_correct).
template <int LMAX>
struct CorrectShT {
  unsigned char seq[LMAX];    // current sequence (bytes as read)
  unsigned char score[LMAX];  // phred per base (DNASeq::score, src/kseq.h:34-40)
  unsigned char minph[LMAX];  // min phred of the window starting here (src/correct_processor.cpp:95-103)
  unsigned char redo[LMAX];   // window must be recounted
  unsigned short pref[LMAX + 1];  // solid windows before this one
  u32 cnt[LMAX];              // occurrences of the window's k-mer (saturated)
  // the read's bases as 2-bit codes (rank - 1), LAST base first: symbol j at bits 2 (j & 31) of word j / 32, so the k-mer
  // starting at s, in the order the backward search consumes it (= the k-mer table's key), is the 2 k bits from bit
  // 2 (n - s - k) on; two words of zeros behind for the unaligned reads
  u64 rv[LMAX / 32 + 3];
};

// Interval::occurrences of the k-mer starting at `s` in sh.seq, with base `ovpos` replaced by rank `ovrank`.
// With the two-step table (t2 != NULL) two backward steps at a time come from the two positions of the first one
// (fm_layout.h: Occ(e, C[c] + Occ(c, p)) = Occ(e, C[c]) + R2(e, c, p)): half the dependent lookups and half the lines.  A
// pair containing a non-ACGT base, and the last step of an odd count, take the one-step form.  An interval that the first
// step of a pair would have emptied comes out empty after the pair (R2 over no rows), so the reference's early exit
// (fmindex.h:67-86) and this give the same count: 0.
// Prefix table of the k-mer lookups (CorrectArgs::ptab, pk = 13 by default): entry [code] = (lower, size) of the interval of
// the pk-mer whose symbols, first one in the highest two bits, are `code` -- what Interval::get holds after its first pk
// symbols (the LAST pk of the k-mer: the search runs backwards).  A lookup whose last pk bases are all ACGT starts there:
// at pk = 12 one gather from a 134 MB table instead of eleven dependent rank steps (5.5 two-step lines).
template <bool WIDE>
__global__ __launch_bounds__(256) void k_prefix_build(FmStrand s, void* tab, u32 pk) {
  typedef typename PosOf<WIDE>::type P;
  const u32 code = blockIdx.x * 256 + threadIdx.x;
  if (code >= (1u << (2 * pk))) return;
  const FmRef f = fm_ref(s, 0);
  u32 r = 1u + (code & 3u);  // the last symbol first (src/fmindex.h:67-79)
  P lo = (P)s.C[r], hi = lo + (P)s.total[r] - 1;
  for (u32 i = 1; i < pk && hi != (P)~(P)0 && hi >= lo; ++i) {
    r = 1u + ((code >> (2 * i)) & 3u);
    P l[5], u[5];
    fm_rank5p<WIDE>(f, lo, l);
    fm_rank5p<WIDE>(f, (P)(hi + 1), u);
    lo = (P)s.C[r] + l[r];
    hi = (P)s.C[r] + u[r] - 1;
  }
  const bool ok = hi != (P)~(P)0 && hi >= lo;
  const P cnt = ok ? (P)(hi - lo + 1) : (P)0;
  if (WIDE) reinterpret_cast<ulonglong2*>(tab)[code] = make_ulonglong2((u64)lo, (u64)cnt);
  else reinterpret_cast<uint2*>(tab)[code] = make_uint2((u32)lo, (u32)cnt);
}

template <bool WIDE>
__device__ __forceinline__ u32 kmer_occ(const FmRef& f, const FmTables& tb, const Find2TablesT<WIDE>* t2, const uint32_t* gran2,
                                        const u64* super2, const void* ptab, u32 pk, const unsigned char* seq, u32 s, u32 k, u32 ovpos,
                                        u32 ovrank, u32& nsec, const void* ktab = nullptr, u64 ktab_slots = 0, const u64* rv = nullptr,
                                        u32 n = 0) {
  typedef typename PosOf<WIDE>::type P;
  if (ktab != nullptr && rv != nullptr) {
    // ACGT-only read with its reverse-packed codes in LDS (CorrectShT::rv): the key is two shifted words, the candidate
    // substitution two bits of it -- a dozen instructions instead of k byte reads and rank conversions
    const u32 o = 2u * (n - s - k), w = o >> 6, sh = o & 63u;
    const u64 a = rv[w], b = rv[w + 1], c = rv[w + 2];
    u64 k0 = sh ? (a >> sh) | (b << (64u - sh)) : a;
    u64 k1 = sh ? (b >> sh) | (c << (64u - sh)) : b;
    if (k <= 32u) {
      if (k < 32u) k0 &= (1ull << (2u * k)) - 1ull;
      k1 = 0;
    } else {
      k1 &= (1ull << (2u * (k - 32u))) - 1ull;
    }
    if (ovpos >= s && ovpos < s + k) {
      const u32 i = s + k - 1u - ovpos;
      const u64 cc = (u64)((ovrank - 1u) & 3u);
      if (i < 32u) k0 = (k0 & ~(3ull << (2u * i))) | (cc << (2u * i));
      else k1 = (k1 & ~(3ull << (2u * (i - 32u)))) | (cc << (2u * (i - 32u)));
    }
    k1 |= (u64)(0x8000u | k) << 48;
    P a0 = 0, a1 = 0, az = 0;
    nsec += 1u;
    if (!deep_lookup<WIDE>(ktab, ktab_slots, k0, k1, a0, a1, az)) return 0u;
    return (u64)az > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)az;
  }
  if (ktab != nullptr) {
    // the table holds every distinct k-mer of the indexed reads with its number of occurrences (fm_layout.h: the deep start
    // table with K = k): one lookup, and a k-mer that is not there does not occur.  Key = the k-mer in the order the
    // backward search consumes it (last base first).  k-mers with a non-ACGT base walk as before.
    u64 k0 = 0, k1 = 0;
    bool acgt = true;
    for (u32 i = 0; i < k; ++i) {
      const u32 p = s + k - 1u - i;
      const u32 r = p == ovpos ? ovrank : base_rank(seq[p]);
      acgt = acgt && r != 0u;
      const u64 c = (u64)((r - 1u) & 3u);
      if (i < 32u) k0 |= c << (2u * i);
      else k1 |= c << (2u * (i - 32u));
    }
    if (acgt) {
      k1 |= (u64)(0x8000u | k) << 48;
      P a0 = 0, a1 = 0, az = 0;
      nsec += 1u;
      if (!deep_lookup<WIDE>(ktab, ktab_slots, k0, k1, a0, a1, az)) return 0u;
      return (u64)az > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)az;
    }
  }
  u32 j = k;
  P lo, hi;
  bool started = false;
  if (ptab != nullptr && k >= pk) {
    u32 code = 0;
    bool acgt = true;
    for (u32 i = 0; i < pk; ++i) {
      const u32 p = s + k - pk + i;
      const u32 r = p == ovpos ? ovrank : base_rank(seq[p]);
      acgt = acgt && r != 0u;
      code = (code << 2) | ((r - 1u) & 3u);
    }
    if (acgt) {
      u64 cnt;
      if (WIDE) {
        const ulonglong2 e = reinterpret_cast<const ulonglong2*>(ptab)[code];
        lo = (P)e.x;
        cnt = e.y;
      } else {
        const uint2 e = reinterpret_cast<const uint2*>(ptab)[code];
        lo = (P)e.x;
        cnt = e.y;
      }
      nsec += 1u;
      if (cnt == 0) return 0u;  // the reference stops updating an empty interval and reports no occurrence
      hi = lo + (P)cnt - 1;
      j = k - pk;
      started = true;
    }
  }
  if (!started) {
    const u32 r0 = (s + j - 1 == ovpos) ? ovrank : base_rank(seq[s + j - 1]);
    lo = (P)tb.C[f.which][r0];
    hi = lo + (P)tb.T[f.which][r0] - 1;  // Interval::init (src/fmindex.h:90-93)
    --j;  // j = steps left; the next symbol is seq[s + j - 1]
  }
  u32 r;
  while (j > 0 && hi != (P)~(P)0 && hi >= lo) {
    r = (s + j - 1 == ovpos) ? ovrank : base_rank(seq[s + j - 1]);
    if (t2 != nullptr && j >= 2 && r != 0) {
      const u32 e = (s + j - 2 == ovpos) ? ovrank : base_rank(seq[s + j - 2]);
      const u64 pl = (u64)lo > f.n ? f.n : (u64)lo, pu0 = (u64)hi + 1ull, pu = pu0 > f.n ? f.n : pu0;
      if (e != 0) {
        const uint4* ql = reinterpret_cast<const uint4*>(gran2 + (pl >> 6) * SIGAX_GRAN2_WORDS);
        const uint4* qu = reinterpret_cast<const uint4*>(gran2 + (pu >> 6) * SIGAX_GRAN2_WORDS);
        auto ld = [](const uint4& v) { v4u w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; return w; };
        Gran2 ga, gb;
        ga.s = ld(ql[0]); ga.pc = ld(ql[r]); ga.p5 = ld(ql[5]); ga.p6 = ld(ql[6]); ga.p7 = ld(ql[7]);
        // the upper position's line only for the lanes where it is another one (find_step2_loads_masked says why)
        if ((pl >> 6) != (pu >> 6)) {
          gb.s = ld(qu[0]); gb.pc = ld(qu[r]); gb.p5 = ld(qu[5]); gb.p6 = ld(qu[6]); gb.p7 = ld(qu[7]);
        } else {
          gb = ga;
        }
        nsec += (pl >> 6) != (pu >> 6) ? 4u : 2u;
        const Rank2 rl = rank2_from(ga, (u32)pl & 63u, r), ru = rank2_from(gb, (u32)pu & 63u, r);
        P l2 = (P)(e == 1 ? rl.pa : e == 2 ? rl.pc : e == 3 ? rl.pg : rl.pt);
        P u2 = (P)(e == 1 ? ru.pa : e == 2 ? ru.pc : e == 3 ? ru.pg : ru.pt);
        if (WIDE) {
          const u32 col = 4u + (r - 1u) * 4u + (e - 1u);
          l2 += (P)super2[(pl >> SIGAX_SUPER_SHIFT) * 20 + col];
          u2 += (P)super2[(pu >> SIGAX_SUPER_SHIFT) * 20 + col];
        }
        const P pb = (P)tb.C[f.which][e] + t2->Cc[f.which][r - 1][e - 1];
        lo = pb + l2;
        hi = pb + u2 - 1;
        j -= 2;
        continue;
      }
    }
    P l[5], u[5];
    fm_rank5p_pair<WIDE>(f, lo, (P)(hi + 1), l, u);  // getOcc(c, lower - 1), getOcc(c, upper)
    nsec += ((u64)lo >> 7) != (((u64)hi + 1ull) >> 7) ? 2u : 1u;
    P lr = r == 0 ? l[0] : r == 1 ? l[1] : r == 2 ? l[2] : r == 3 ? l[3] : l[4];
    P ur = r == 0 ? u[0] : r == 1 ? u[1] : r == 2 ? u[2] : r == 3 ? u[3] : u[4];
    P pb = (P)tb.C[f.which][r];
    lo = pb + lr;
    hi = pb + ur - 1;
    --j;
  }
  if (!(hi != (P)~(P)0 && hi >= lo)) return 0u;
  u64 c = (u64)(hi - lo) + 1ull;
  return c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)c;
}

template <bool WIDE, int LMAX>
// (64-bit positions: held to four, which gives 96 registers and five waves; six would spill)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LMAX <= 512 ? (WIDE ? 4 : SIGAX_CORRECT_WAVES) : 3))) void k_correct(CorrectArgs A) {
  typedef CorrectShT<LMAX> CorrectSh;
  __shared__ FmTables tb;
  __shared__ CorrectSh shm[4];
  __shared__ Find2TablesT<WIDE> t2s;
  const bool have2 = A.fwd.gran2 != nullptr && (!WIDE || A.fwd.super2 != nullptr);
  if (have2) find2_tables_load<WIDE>(t2s, A.fwd, A.fwd);
  const Find2TablesT<WIDE>* t2 = have2 ? &t2s : nullptr;
  fm_tables_load(tb, A.fwd, A.fwd);
  const u32 wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const u64 lt = (1ull << lane) - 1ull;
  CorrectSh& sh = shm[wid];
  const FmRef F = fm_ref(A.fwd, 0);
  const u64 wave = (u64)blockIdx.x * 4 + wid, nwaves = (u64)gridDim.x * 4;
  const u32 k = A.k;
  u32 nsec = 0, nlook = 0;  // per lane: distinct 64-byte sectors asked for, k-mer lookups made
  for (u64 rd = wave; rd < A.n_reads; rd += nwaves) {
    const u64 b0 = A.offs[rd];
    const u32 n = uni((u32)(A.offs[rd + 1] - b0));
    u32 valid = 0;
    if (A.only_deferred && A.valid[rd] != 3) continue;  // the second launch: what the first one left
    if (n > (u32)LMAX) {
      // too long for this form: for the 1024 form when another launch follows, for good otherwise
      if (lane == 0) {
        if (LMAX < CORRECT_LMAX) {
          A.valid[rd] = 3;
        } else {
          A.valid[rd] = 2;
          atomicAdd(&A.dstat[0], 1ull);
        }
      }
      continue;
    }
    for (u32 i = lane; i < n; i += 64) {
      unsigned char c = A.seqs[b0 + i];
      sh.seq[i] = c;
      A.out[b0 + i] = c;  // r.seq = item.read.seq unless the read becomes all-solid
      sh.score[i] = A.quals ? (unsigned char)(A.quals[b0 + i] - 33) : (unsigned char)15;
    }
    wave_lds_sync();
    if (n >= k) {  // src/correct_processor.cpp:85-89
      const u32 nw = n - k + 1;
      // the reverse-packed codes for the k-mer table's keys (only reads of A, C, G, T: the others keep the byte-wise keys)
      bool clean = false;
      if (A.ktab != nullptr) {
        bool bad = false;
        for (u32 w = lane; w < n / 32u + 3u; w += 64) {
          u64 v = 0;
          for (u32 j = 0; j < 32u; ++j) {
            const u32 t = 32u * w + j;
            if (t < n) {
              const u32 r = base_rank(sh.seq[n - 1u - t]);
              bad = bad || r == 0u;
              v |= (u64)((r - 1u) & 3u) << (2u * j);
            }
          }
          sh.rv[w] = v;
        }
        clean = __ballot(bad) == 0ull;
      }
      const u64* rv = clean ? sh.rv : nullptr;
      for (u32 s = lane; s < nw; s += 64) {
        u32 m = 15;  // no qualities (FASTA): every base scores 15
        if (A.quals != nullptr) {
          m = 255;
          for (u32 j = 0; j < k; ++j) m = min(m, (u32)sh.score[s + j]);
        }
        sh.minph[s] = (unsigned char)m;
        sh.redo[s] = 1;
      }
      wave_lds_sync();
      u32 rounds = 0;
      while (true) {
        // k-mer counts of the windows that changed, solid windows, prefix counts of solid windows
        u32 base = 0;
        for (u32 s0 = 0; s0 < nw; s0 += 64) {
          const u32 s = s0 + lane;
          bool good = false;
          if (s < nw) {
            if (sh.redo[s]) {
              sh.cnt[s] = kmer_occ<WIDE>(F, tb, t2, A.fwd.gran2, A.fwd.super2, A.ptab, A.pk, sh.seq, s, k, 0xFFFFFFFFu, 0u, nsec, A.ktab, A.ktab_slots, rv, n);
              ++nlook;
              sh.redo[s] = 0;
            }
            good = sh.cnt[s] >= (sh.minph[s] >= A.cutoff ? A.high : A.low);  // CorrectThreshold::requiredSupport
          }
          u64 gm = __ballot(good);
          if (s < nw) sh.pref[s] = (unsigned short)(base + (u32)__popcll(gm & lt));
          base += (u32)__popcll(gm);
        }
        if (lane == 0) sh.pref[nw] = (unsigned short)base;
        wave_lds_sync();
        // a base is solid if some solid window covers it (:131-135)
        bool allSolid = true;
        for (u32 i0 = 0; i0 < n; i0 += 64) {
          const u32 i = i0 + lane;
          bool unsolid = false;
          if (i < n) {
            u32 lo = i + 1 >= k ? i + 1 - k : 0, hi = i < nw - 1 ? i : nw - 1;
            unsolid = sh.pref[hi + 1] == sh.pref[lo];
          }
          if (__ballot(unsolid)) allSolid = false;
        }
        if (allSolid) { valid = 1; break; }
        if (++rounds > A.rounds) break;
        // leftmost base that can be corrected, trying its leftmost then rightmost covering k-mer (:154-173)
        bool corrected = false;
        for (u32 i0 = 0; i0 < n && !corrected; i0 += 64) {
          const u32 i = i0 + lane;
          bool unsolid = false;
          if (i < n) {
            u32 lo = i + 1 >= k ? i + 1 - k : 0, hi = i < nw - 1 ? i : nw - 1;
            unsolid = sh.pref[hi + 1] == sh.pref[lo];
          }
          u64 m = __ballot(unsolid);
          while (m && !corrected) {
            // lane = (candidate q = lane / 8, side = (lane / 4) & 1, base rank = 1 + lane % 4)
            const u32 q = lane >> 3, side = (lane >> 2) & 1u, brank = 1u + (lane & 3u);
            u64 mm = m;
            u32 pos = 0xFFFFFFFFu;
            for (u32 t = 0; t < 8; ++t) {
              u32 bit = mm ? (u32)__ffsll((long long)mm) - 1u : 0xFFFFFFFFu;
              if (t == q && mm) pos = i0 + bit;
              if (mm) mm &= mm - 1;
            }
            bool cand = false;
            if (pos != 0xFFFFFFFFu) {
              const u32 cur = base_rank(sh.seq[pos]);
              if (brank != cur) {  // c != currBase (:204-206)
                const u32 kidx = side ? (pos < n - k ? pos : n - k) : (pos + 1 >= k ? pos + 1 - k : 0u);
                const u32 thr = sh.score[pos] >= A.cutoff ? A.high : A.low;
                const u32 minCount = A.offset > thr ? A.offset : thr;  // max(countVector[..] (always 0) + offset, threshold)
                cand = kmer_occ<WIDE>(F, tb, t2, A.fwd.gran2, A.fwd.super2, A.ptab, A.pk, sh.seq, kidx, k, pos, brank, nsec, A.ktab, A.ktab_slots, rv, n) >= minCount;
                ++nlook;
              }
            }
            // try2Correct succeeds iff exactly one alternative base reaches minCount (:207-223)
            const u64 cm = __ballot(cand);
            const u32 grp = (u32)((cm >> (lane & ~3u)) & 0xFull);
            const bool win = cand && __popc(grp) == 1;
            const u64 wm = __ballot(win);
            if (wm) {
              const u32 wl = (u32)__ffsll((long long)wm) - 1u;  // lowest lane = smallest position, left before right
              const u32 wpos = __builtin_amdgcn_readlane(pos, wl);
              const u32 wr = 1u + (wl & 3u);
              if (lane == 0) {
                sh.seq[wpos] = wr == 1 ? 'A' : wr == 2 ? 'C' : wr == 3 ? 'G' : 'T';
                if (rv != nullptr) {  // ... and in the packed copy
                  const u32 t = n - 1u - wpos;
                  sh.rv[t >> 5] = (sh.rv[t >> 5] & ~(3ull << (2u * (t & 31u)))) | ((u64)(wr - 1u) << (2u * (t & 31u)));
                }
              }
              const u32 wlo = wpos + 1 >= k ? wpos + 1 - k : 0, whi = wpos < nw - 1 ? wpos : nw - 1;
              for (u32 s = wlo + lane; s <= whi; s += 64) sh.redo[s] = 1;
              corrected = true;
            }
            m = mm;  // the eight candidates just tried are done
          }
        }
        wave_lds_sync();
        if (!corrected) break;
      }
    }
    if (valid) {
      for (u32 i = lane; i < n; i += 64) A.out[b0 + i] = sh.seq[i];
    }
    if (lane == 0) A.valid[rd] = (unsigned char)valid;
    wave_lds_sync();
  }
  const u64 tsec = wave_sum((u64)nsec), tlook = wave_sum((u64)nlook);
  if (lane == 0) {
    if (tsec) atomicAdd(&A.dstat[1], tsec);
    if (tlook) atomicAdd(&A.dstat[2], tlook);
  }
}

// -------------------------------------------------------------------------------------------------------
// exclusive scan of u32 counts into u64 offsets (three small kernels), ordered scatter of the final blocks
// -------------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 2048  // per workgroup of 256 threads

__device__ __forceinline__ u64 block_exclusive_scan(u64 v, u64* total) {
  __shared__ u64 wsum[4];
  u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  u64 x = v;
  for (int off = 1; off < 64; off <<= 1) {
    u64 y = __shfl_up(x, off, 64);
    if (lane >= (u32)off) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  u64 base = 0;
  for (u32 w = 0; w < wid; ++w) base += wsum[w];
  u64 tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  *total = tot;
  return base + x - v;
}

__global__ __launch_bounds__(256) void k_scan_partials(const u32* cnt, u64 n, u64* partial) {
  u64 base = (u64)blockIdx.x * SCAN_ITEMS + (u64)threadIdx.x * 8;
  u64 s = 0;
  for (int k = 0; k < 8; ++k)
    if (base + k < n) s += cnt[base + k];
  u64 tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_top(u64* partial, u64 nparts, u64* total_out) {
  u64 carry = 0;
  for (u64 base = 0; base < nparts; base += 256) {
    u64 i = base + threadIdx.x;
    u64 v = i < nparts ? partial[i] : 0;
    u64 tot;
    u64 ex = block_exclusive_scan(v, &tot);
    if (i < nparts) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const u32* cnt, u64 n, const u64* partial, u64* offs) {
  u64 base = (u64)blockIdx.x * SCAN_ITEMS + (u64)threadIdx.x * 8;
  u64 v[8], s = 0;
  for (int k = 0; k < 8; ++k) {
    v[k] = (base + k < n) ? cnt[base + k] : 0;
    s += v[k];
  }
  u64 tot;
  u64 ex = block_exclusive_scan(s, &tot) + partial[blockIdx.x];
  for (int k = 0; k < 8; ++k) {
    if (base + k < n) offs[base + k] = ex;
    ex += v[k];
  }
  if (base <= n && n < base + 8) offs[n] = ex;  // the thread owning position n writes the grand total
}

// block_offs[r] = offs2[2r]: per-read offsets from the per-(read, side) scan
__global__ __launch_bounds__(256) void k_pick_read_offsets(const u64* offs2, u64 n_reads, u64* block_offs) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i <= n_reads) block_offs[i] = offs2[2 * i];
}

// one lane per (read, side) item: copy its blocks from the unordered arena to their place in the ordered output; blocks
// written by the lane-group kernels get their raw pair, length and flags from the candidate record here
template <bool WIDE>
__device__ __forceinline__ void order_fill_from_cand(const OrderArgs& A, u64 item, u32 src, ulonglong2& v2, ulonglong2& v3, ulonglong2& v4) {
  const Cand<WIDE>* rec = reinterpret_cast<const Cand<WIDE>*>(A.arena) + (item >> 1) * 4 * A.cap + (src & 0x3FFFFFFFu);
  const Cand<WIDE> b = cand_load<WIDE>(rec);
  v2 = make_ulonglong2((u64)b.r0lo, (u64)b.r0lo + b.sz - 1);
  v3 = make_ulonglong2((u64)b.r1lo, (u64)b.r1lo + b.sz - 1);
  v4 = make_ulonglong2((u64)b.len | ((u64)b.af << 32), 0ull);
}
// Hit2OverlapConverter::convert's test for one (query, target) pair (src/overlap_builder.cpp:358,365)
__device__ __forceinline__ bool edge_kept(const uint32_t* name_rank, const uint32_t* read_len, u32 q, u32 t, u32 len, u32 af) {
  u32 nq = name_rank[q], nt = name_rank[t];
  if (nq == nt) return false;                       // query.name != target.name (:358)
  bool contained = (len == read_len[q]) || (len == read_len[t]);  // Match::isContainment (coord.h:150-152)
  if (nq < nt || (contained && (af & 1u))) return false;              // dedup rule (:365)
  return true;
}
// While a block is on its way to its place, its edge records are counted (the block and its .sai range are in registers;
// the item knows its read): the item's total goes to item_edges, whose scan gives k_edges_fill the item's place in the
// edge list.  Round 2 counted per block in a pass of its own that found each block's read by binary search.
__global__ __launch_bounds__(256) void k_order_scatter(OrderArgs A) {
  u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
  if (w >= A.n_items) return;
  u32 cnt = A.fin_cnt[w];
  u32 kept = 0;
  u32 q = A.read_base + (u32)(w >> 1);
  bool known = true;  // the read is one of the index's
  if (A.read_ids != nullptr) {
    q = A.read_ids[w >> 1];
    known = q < A.n_index_reads;
    if (!known) atomicAdd(A.bad_ids, 1ull);
  }
  if (cnt) {
    u64 srcb = A.item_base[w], dstb = A.offs2[w];
    for (u32 i = 0; i < cnt; ++i) {
      if (srcb + i >= A.fin_cap || dstb + i >= A.out_cap) break;  // only after an overflow, whose results the host discards
      const ulonglong2* s = reinterpret_cast<const ulonglong2*>(A.fin + srcb + i);
      ulonglong2* d = reinterpret_cast<ulonglong2*>(A.out + dstb + i);
      ulonglong2 v0 = s[0], v1 = s[1], v4 = s[4];
      ulonglong2 v2, v3;
      if (v4.y >> 63) {
        if (A.wide) order_fill_from_cand<true>(A, w, (u32)v4.y, v2, v3, v4);
        else order_fill_from_cand<false>(A, w, (u32)v4.y, v2, v3, v4);
      } else {
        v2 = s[2]; v3 = s[3];
      }
      d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3; d[4] = v4;
      if (A.item_edges != nullptr && known) {
        const u32 len = (u32)v4.x, af = (u32)(v4.x >> 32);
        const u32* sa = (af & 2u) ? A.rsai : A.sai;
        for (u64 j = v0.x; j <= v0.y && j < A.n_sai; ++j)
          if (edge_kept(A.name_rank, A.read_len, q, sa[j], len, af)) ++kept;
      }
    }
  }
  if (A.item_edges != nullptr) A.item_edges[w] = kept;
}

// -------------------------------------------------------------------------------------------------------
// edges: Hit2OverlapConverter::convert (overlap_builder.cpp:345-375), one lane per (read, side) item: its blocks are
// consecutive in the ordered output, its records consecutive in the edge list (hits order = the ED order at -t 1)
// -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edges_fill(EdgeArgs A) {
  const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
  if (w >= A.n_items) return;
  const u32 cnt = A.fin_cnt[w];
  if (!cnt) return;
  u64 e = A.edge_offs[w];
  if (A.edge_offs[w + 1] == e) return;  // no record from this item (most items: the self-containment blocks)
  const u64 first = A.offs2[w];
  const u32 q = A.read_ids != nullptr ? A.read_ids[w >> 1] : A.read_base + (u32)(w >> 1);
  for (u32 i = 0; i < cnt; ++i) {
    if (first + i >= A.blocks_cap) return;  // only after an overflow, whose results the host discards
    const ulonglong2* s = reinterpret_cast<const ulonglong2*>(A.blocks + first + i);
    const ulonglong2 v0 = s[0], v4 = s[4];
    const u32 len = (u32)v4.x, af = (u32)(v4.x >> 32);
    const u32* sa = (af & 2u) ? A.rsai : A.sai;
    for (u64 j = v0.x; j <= v0.y && j < A.n_sai; ++j) {
      const u32 t = sa[j];
      if (edge_kept(A.name_rank, A.read_len, q, t, len, af)) {
        if (e < A.edge_cap) {
          sigax_edge rec;
          rec.query = q; rec.target = t; rec.length = len; rec.af = af;
          A.edges[e] = rec;
        }
        ++e;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------------
// launch wrappers
// -------------------------------------------------------------------------------------------------------
static inline unsigned nblk(u64 n, u64 per) { return (unsigned)((n + per - 1) / per); }

void launch_occ_batch(const FmStrand& s, bool wide, const u64* pos, u64 n, u64* out, hipStream_t st) {
  if (n == 0) return;
  if (wide) hipLaunchKernelGGL(k_occ_batch<true>, dim3(nblk(n, 256)), dim3(256), 0, st, s, pos, n, out);
  else hipLaunchKernelGGL(k_occ_batch<false>, dim3(nblk(n, 256)), dim3(256), 0, st, s, pos, n, out);
}

void launch_kmer_count(const FmStrand& s, bool wide, const unsigned char* kmers, u32 k, u64 n, u64* out, hipStream_t st) {
  if (n == 0) return;
  if (wide) hipLaunchKernelGGL(k_kmer_count<true>, dim3(nblk(n, 256)), dim3(256), 0, st, s, kmers, k, n, out);
  else hipLaunchKernelGGL(k_kmer_count<false>, dim3(nblk(n, 256)), dim3(256), 0, st, s, kmers, k, n, out);
}

unsigned long long prefix_table_bytes(bool wide, u32 pk) { return (1ull << (2 * pk)) * (wide ? 16u : 8u); }
void launch_prefix_build(const FmStrand& s, bool wide, void* tab, u32 pk, hipStream_t st) {
  const unsigned g = (1u << (2 * pk)) / 256u;
  if (wide) hipLaunchKernelGGL(k_prefix_build<true>, dim3(g), dim3(256), 0, st, s, tab, pk);
  else hipLaunchKernelGGL(k_prefix_build<false>, dim3(g), dim3(256), 0, st, s, tab, pk);
}

unsigned long long start_table_bytes(bool wide) { return (1ull << (2 * SIGAX_START_K)) * (wide ? 32u : 16u); }
void launch_start_build(const FmStrand& prim, const FmStrand& other, bool wide, void* tab, hipStream_t st) {
  const unsigned g = (1u << (2 * SIGAX_START_K)) / 256u;
  if (wide) hipLaunchKernelGGL(k_start_build<true>, dim3(g), dim3(256), 0, st, prim, other, tab);
  else hipLaunchKernelGGL(k_start_build<false>, dim3(g), dim3(256), 0, st, prim, other, tab);
}

unsigned long long deep_entry_bytes() { return 32; }
void launch_deep_scan(const FmStrand& s, const u32* slen, u64 n_stretch, u32 K, u64* count, u64* list, u64 list_cap, hipStream_t st) {
  if (s.n == 0) return;
  hipLaunchKernelGGL(k_deep_scan, dim3((unsigned)std::min<u64>((s.n + 255) / 256, 1u << 22)), dim3(256), 0, st, s, slen, n_stretch, K, count, list, list_cap);
}
void launch_deep_fill(const FmStrand& prim, const FmStrand& other, bool wide, const u32* slen, u64 n_stretch, u32 K, const u64* list, u64 n_list,
                      void* tab, u64 nslots, u64* err, hipStream_t st) {
  if (n_list == 0) return;
  if (wide) hipLaunchKernelGGL(k_deep_fill<true>, dim3(nblk(n_list, 256)), dim3(256), 0, st, prim, other, slen, n_stretch, K, list, n_list, (u64*)tab, nslots, err);
  else hipLaunchKernelGGL(k_deep_fill<false>, dim3(nblk(n_list, 256)), dim3(256), 0, st, prim, other, slen, n_stretch, K, list, n_list, (u64*)tab, nslots, err);
}
void launch_suffix_order_check(const FmStrand& s, const u32* sai, u32* isai_tmp, const u32* read_len, u64 n_strings, u64* bad3, hipStream_t st) {
  if (n_strings == 0 || s.n < 2) return;
  hipLaunchKernelGGL(k_isai, dim3(nblk(n_strings, 256)), dim3(256), 0, st, sai, n_strings, isai_tmp);
  hipLaunchKernelGGL(k_suffix_order_check, dim3(nblk(s.n - 1, 256)), dim3(256), 0, st, s, sai, (const u32*)isai_tmp, read_len, n_strings, bad3);
}

void launch_correct(const CorrectArgs& a0, bool wide, hipStream_t st) {
  if (a0.n_reads == 0) return;
  unsigned g = (unsigned)((a0.n_reads + 3) / 4);
  if (g > 4096) g = 4096;
  CorrectArgs a = a0;
  a.only_deferred = 0;
  // reads of up to 512 bases in the small form; it marks longer ones (valid = 3) for the 1024 form, which is launched
  // behind it unless the caller knows there are none (max_len: an upper bound of the batch's read lengths, 0 = unknown)
  if (a0.max_len == 0 || a0.max_len <= 512) {
    if (wide) hipLaunchKernelGGL((k_correct<true, 512>), dim3(g), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_correct<false, 512>), dim3(g), dim3(256), 0, st, a);
    if (a0.max_len != 0) return;
    a.only_deferred = 1;
  }
  if (wide) hipLaunchKernelGGL((k_correct<true, CORRECT_LMAX>), dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_correct<false, CORRECT_LMAX>), dim3(g), dim3(256), 0, st, a);
}

// LDS budget of a per-lane finder workgroup (its residency cap and its read-staging buffer; SIGAX_FIND_LDS overrides).
// Round 4: 45000 = THREE workgroups (twelve waves) per CU and two filter/extract workgroups beside them, where rounds 1-3 ran
// two and three: since the deep start table and the extractor's fast-forward took a quarter of each kernel's work, the
// finder is what the step waits for, and it follows its residency (tools/sweep_env.sh, BASELINE configs[1], same box: 60000
// 145.7 M reads/s, 50000 143.6, 47000 149.6, 45000 153.4-153.8 -- with a filter/extract grid of 2.5 per CU 155.9 --, 43000
// and below slower again: the finder's workgroups then leave filter/extract no LDS at all).  Reads too long for 128 of them
// to fit the smaller staging buffer get the budget they need, up to the old 60000.
static unsigned find_lds_budget(unsigned max_len) {
  static const char* env = getenv("SIGAX_FIND_LDS");
  if (env) return (unsigned)atoi(env);
  const unsigned need = (unsigned)sizeof(FindStage) + 128u * max_len + 64u;
  return need <= 45000u ? 45000u : (need <= 60000u ? need : 60000u);
}
// bytes of a workgroup's reads that can be staged in LDS when the batch's longest read has max_len bases (see launch_find)
unsigned long long find_stage_capacity(unsigned max_len) {
  const unsigned lds = find_lds_budget(max_len);
  return lds > (unsigned)sizeof(FindStage) ? lds - (unsigned)sizeof(FindStage) : 0u;
}

void launch_find(const FindArgs& a, bool wide, hipStream_t st) {
  if (a.read_end <= a.read_begin) return;
  const unsigned bs = 256u;
  unsigned g = nblk((u64)(a.read_end - a.read_begin), 256u / a.chains_per_wg);  // 64 reads x 4 chains (or 128 x 2) per workgroup
  // Unused dynamic LDS caps the finder's residency (it saturates the memory request rate with few waves), leaving
  // wave slots and registers for the filter/extract kernel that runs beside it on the other stream.
  // Measured on MI355X at C2: 28 resident waves/CU 15.7 ms, 12 waves 14.5 ms, 8 waves 13.4 ms, 4 waves 14.9 ms.
  // 60 KB per workgroup = two workgroups (8 waves) per CU and 40 KB of LDS left for filter/extract workgroups.
  unsigned lds = (unsigned)find_stage_capacity(a.max_len);  // the record staging rows are part of the budget
  FindArgs b = a;
  b.stage_bytes = lds;  // the residency cap doubles as the staging buffer for the workgroup's reads
  if (a.coop && a.two_step && a.fwd.gran2 && a.rev.gran2 && a.chains_per_wg == 2) {
    // 64 reads per workgroup; the dynamic LDS holds exactly their bases (no residency padding: LDS is what limits it)
    const unsigned gc = nblk((u64)(a.read_end - a.read_begin), 64u);
    b.stage_bytes = a.coop_stage_bytes;
    // 2 KB of unused LDS per workgroup: five of them per CU instead of six, and 20 KB of a CU's LDS left for the kernels
    // that run beside the finder (with six the filter/extract workgroups of the batches in flight found no LDS on a CU
    // until finder workgroups retired: BASELINE configs[2] shape 88.3 -> 90.3 M reads/s; SIGAX_FIND_COOP_PAD=0 for six)
    static const char* env_pad = getenv("SIGAX_FIND_COOP_PAD");
    const unsigned dyn = a.coop_stage_bytes + (env_pad ? (unsigned)atoi(env_pad) : 2048u);
    const unsigned gp = a.coop_grid ? std::min(gc, a.coop_grid) : gc;
    if (wide) hipLaunchKernelGGL(k_find_c2w, dim3(gp), dim3(128), dyn, st, b);
    else hipLaunchKernelGGL(k_find_c2, dim3(gp), dim3(128), dyn, st, b);
    return;
  }
  if (wide) hipLaunchKernelGGL(k_find_w, dim3(g), dim3(bs), lds, st, b);
  else if (a.two_step && a.fwd.gran2 && a.rev.gran2) hipLaunchKernelGGL(k_find_n2, dim3(g), dim3(bs), lds, st, b);
  else hipLaunchKernelGGL(k_find_n, dim3(g), dim3(bs), lds, st, b);
}

template <bool WIDE>
static void launch_fx_stages(const FxArgs& a, unsigned grid32, unsigned grid64, const u64* qhint, hipStream_t st) {
  static const bool no_lean = getenv("SIGAX_FX_NO_LEAN") != nullptr;  // A/B aid
  const bool lean = a.irreducible && !no_lean;
  const bool have2 = a.fwd.gran2 && a.rev.gran2;
  // Irreducible mode, a chain of launches, each taking what the one before could not finish:
  //   strict lean 32 lanes (single-group rounds only; skipped when a.no_lean says most items branch)
  //   -> branching lean 32 lanes -> branching lean 64 lanes (items of 33..64 blocks) -> full 64 lanes (everything else the
  //   lane-group form can do, items of up to 256 blocks) -> general kernel (launched by the caller).
  // Exhaustive mode: full 32 lanes -> full 64 lanes.
  u32* q[4] = {a.work64, a.work64b, a.work64c, a.work64d};
  u64* qn[4] = {a.w64_counter, a.w64b_counter, a.w64c_counter, a.w64d_counter};
  // stage(kernel, grid, in, out, wide): queue indices, -1 = the read range as input / the general kernel as output /
  // no separate queue for items wider than the lane group
  auto stage = [&](auto kernel, unsigned grid, int in, int out, int widei, unsigned items_per_wg = 0) {
    FxArgs x = a;
    if (in >= 0 && qhint && qhint[in] != ~0ull) {
      // a queue that held few items in this batch object's previous run gets a small grid (any grid is correct: the
      // waves loop over the queue; an empty 768-workgroup launch costs ~25 us beside the finder)
      const u64 per_wg = items_per_wg ? items_per_wg : grid == grid32 ? 8 : 4;
      const u64 want = (qhint[in] + qhint[in] / 2 + per_wg - 1) / per_wg + 4;
      if (want < grid) grid = (unsigned)want;
    }
    x.q_in = in < 0 ? nullptr : q[in];
    x.q_in_n = in < 0 ? nullptr : qn[in];
    x.q_out = out < 0 ? nullptr : q[out];
    x.q_out_n = out < 0 ? nullptr : qn[out];
    x.q_wide = widei < 0 ? nullptr : q[widei];
    x.q_wide_n = widei < 0 ? nullptr : qn[widei];
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, x);
  };
  auto route = [&]() {  // wide items of the range straight into queue 1
    FxArgs x = a;
    x.q_wide = q[1];
    x.q_wide_n = qn[1];
    hipLaunchKernelGGL(k_fx_route, dim3(nblk(2ull * (a.read_end - a.read_begin), 256)), dim3(256), 0, st, x);
  };
  if (!lean) {
    stage(k_filter_extract_fast<WIDE, 32, 0>, grid32, -1, 2, -1);
    stage(k_filter_extract_fast<WIDE, 64, 0>, grid64, 2, -1, -1);
    return;
  }
  // queue 0: items the strict launch handed on; queue 1: for the branching 64-lane launch; queue 2: for the full one;
  // queue 3 (text launches): what the 16-lane launch leaves to the branching 32-lane one
  static const bool no_text = getenv("SIGAX_FX_NO_TEXT") != nullptr;  // A/B aid
  static const bool use_16 = getenv("SIGAX_FX_16") != nullptr;  // see below
  if (a.fwd.text && a.rev.text && ((a.fwd.sa && a.rev.sa) || (a.fwd.xmap && a.rev.xmap)) && !no_text) {
    if (!a.no_lean) { route(); stage(k_filter_extract_fast<WIDE, 32, 5>, grid32, -1, 0, -1); }
    // SIGAX_FX_16=1: branching items of at most 16 blocks -- what reads with substitutions mostly make: errors cost them
    // overlaps -- four to a wave in a launch of their own; wider ones, and whatever outgrows a quarter wave's slots, go on
    // to the 32-lane launch.  Built on the round-2 expectation that twice the items in flight would halve the time of reads
    // with errors; measured, it is +7 % at 1 % substitutions (31.6 -> 33.7 M reads/s) and -5 % at 0.3 % (60.4 -> 57.6): four
    // items that branch differently share a wave's instruction stream, and the branching rounds are bound by instruction
    // issue, not by latency.  Off by default.
    if (use_16) {
      stage(k_filter_extract_fast<WIDE, 16, 6>, grid32, a.no_lean ? -1 : 0, 3, 3, 16);
      stage(k_filter_extract_fast<WIDE, 32, 6>, grid32, 3, 1, -1);
    } else {
      stage(k_filter_extract_fast<WIDE, 32, 6>, grid32, a.no_lean ? -1 : 0, 1, -1);
    }
    stage(k_filter_extract_fast<WIDE, 64, 6>, grid64, 1, 2, -1);
  } else if (have2) {
    if (!a.no_lean) { route(); stage(k_filter_extract_fast<WIDE, 32, 1>, grid32, -1, 0, -1); }
    stage(k_filter_extract_fast<WIDE, 32, 3>, grid32, a.no_lean ? -1 : 0, 1, -1);
    stage(k_filter_extract_fast<WIDE, 64, 3>, grid64, 1, 2, -1);
  } else {
    if (!a.no_lean) { route(); stage(k_filter_extract_fast<WIDE, 32, 2>, grid32, -1, 0, -1); }
    stage(k_filter_extract_fast<WIDE, 32, 4>, grid32, a.no_lean ? -1 : 0, 1, -1);
    stage(k_filter_extract_fast<WIDE, 64, 4>, grid64, 1, 2, -1);
  }
  stage(k_filter_extract_fast<WIDE, 64, 0>, grid64, 2, -1, -1);
}

void launch_filter_extract_fast(const FxArgs& a, bool wide, unsigned grid32, unsigned grid64, const u64* qhint, hipStream_t st) {
  if (a.read_end <= a.read_begin) return;
  if (wide) launch_fx_stages<true>(a, grid32, grid64, qhint, st);
  else launch_fx_stages<false>(a, grid32, grid64, qhint, st);
}

unsigned long long fast_pool_entries_per_wave() { return FX_WPOOL; }
unsigned long long fast_fin_chunk() { return FX_FIN_CHUNK; }
unsigned long long cand_bytes(bool wide) { return wide ? sizeof(Cand<true>) : sizeof(Cand<false>); }

void launch_filter_extract(const FxArgs& a, bool wide, unsigned grid, hipStream_t st) {
  if (a.n_work == 0 && !a.n_work_ptr) return;
  if (wide) hipLaunchKernelGGL(k_filter_extract<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_filter_extract<false>, dim3(grid), dim3(256), 0, st, a);
}

void launch_scan(const u32* cnt, u64 n, u64* partial, u64* offs, u64* total_out, hipStream_t st) {
  // offs has n+1 entries; offs[n] = total
  unsigned g = nblk(n + 1, SCAN_ITEMS);
  hipLaunchKernelGGL(k_scan_partials, dim3(g), dim3(256), 0, st, cnt, n, partial);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, st, partial, (u64)g, total_out);
  hipLaunchKernelGGL(k_scan_apply, dim3(g), dim3(256), 0, st, cnt, n, (const u64*)partial, offs);
}

void launch_build2(const FmStrand& s, bool wide, u32* gran2, u64* super2, u32* cnt, u64* offs, u64* partial, u64* total, hipStream_t st) {
  const u64 ng2 = s.n / SIGAX_GRAN2_SYMS + 1;
  const unsigned g = (unsigned)std::min<u64>(nblk(ng2, 4), 65536);
  if (wide) hipLaunchKernelGGL(k_build2_a<true>, dim3(g), dim3(256), 0, st, s, gran2, cnt, ng2);
  else hipLaunchKernelGGL(k_build2_a<false>, dim3(g), dim3(256), 0, st, s, gran2, cnt, ng2);
  for (u32 col = 0; col < 20; ++col) {
    launch_scan(cnt + (u64)col * ng2, ng2, partial, offs, total, st);
    hipLaunchKernelGGL(k_build2_b, dim3(nblk(ng2, 256)), dim3(256), 0, st, (const u64*)offs, gran2, col, ng2, wide ? super2 : (u64*)nullptr);
  }
}

void launch_stretch_scan(const FmStrand& s, bool wide, u64 n_stretch, u64* info, u32* maxlen, hipStream_t st) {
  if (n_stretch == 0) return;
  if (wide) hipLaunchKernelGGL(k_stretch_scan<true>, dim3(nblk(n_stretch, 256)), dim3(256), 0, st, s, n_stretch, info, maxlen);
  else hipLaunchKernelGGL(k_stretch_scan<false>, dim3(nblk(n_stretch, 256)), dim3(256), 0, st, s, n_stretch, info, maxlen);
}
void launch_rows_fill(const FmStrand& s, bool wide, u64 n_stretch, const u64* info, unsigned char* sa, u32 sa_bits, u32 ld_bits, u32 t_bits,
                      unsigned char* text, u32 text_stride, u32* slen, hipStream_t st) {
  if (n_stretch == 0) return;
  if (wide) hipLaunchKernelGGL(k_rows_fill<true>, dim3(nblk(n_stretch, 256)), dim3(256), 0, st, s, n_stretch, info, sa, sa_bits, ld_bits, t_bits, text, text_stride, slen);
  else hipLaunchKernelGGL(k_rows_fill<false>, dim3(nblk(n_stretch, 256)), dim3(256), 0, st, s, n_stretch, info, sa, sa_bits, ld_bits, t_bits, text, text_stride, slen);
}
void launch_xmap(const u32* sai_y, const u32* sai_x, u32* isai_tmp, const u32* slen_x, u64 n, u64* xmap, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_isai, dim3(nblk(n, 256)), dim3(256), 0, st, sai_x, n, isai_tmp);
  hipLaunchKernelGGL(k_xmap, dim3(nblk(n, 256)), dim3(256), 0, st, sai_y, (const u32*)isai_tmp, slen_x, n, xmap);
}

u64 scan_partials_needed(u64 n) { return (n + 1 + SCAN_ITEMS - 1) / SCAN_ITEMS + 1; }

void launch_pick_read_offsets(const u64* offs2, u64 n_reads, u64* block_offs, hipStream_t st) {
  hipLaunchKernelGGL(k_pick_read_offsets, dim3(nblk(n_reads + 1, 256)), dim3(256), 0, st, offs2, n_reads, block_offs);
}

void launch_order_scatter(const OrderArgs& a, hipStream_t st) {
  if (a.n_items == 0) return;
  hipLaunchKernelGGL(k_order_scatter, dim3(nblk(a.n_items, 256)), dim3(256), 0, st, a);
}

void launch_edges_fill(const EdgeArgs& a, hipStream_t st) {
  if (a.n_items == 0) return;
  hipLaunchKernelGGL(k_edges_fill, dim3(nblk(a.n_items, 256)), dim3(256), 0, st, a);
}
