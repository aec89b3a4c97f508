// Raw deflate (RFC 1951) for line-structured text of sequence records, written for the ASQG writer (src/asqg.cpp:
// VT / ED lines; src/utils.cpp:92-126 puts a gzip filter on names that end with ".gz").  Host side of SURVEY 8 f4.
//
// zlib spends its time hashing every position of four-letter text in which a 32 KiB window (200 reads) holds next to
// nothing to match; what there IS to match in such text sits one line up: the tag, the running name prefix, the
// trailing fields, the numbers of an ED line that repeat.  So the match finder here looks in exactly one place -- field
// j of the line above, for every field j of a line -- and everything else goes out as literals under one dynamic
// Huffman code per block (pairs of literals per table lookup).  Any inflate reads the result.  Measured per thread
// (tools/deflate_probe.cpp): VT lines 650 MB/s at 0.270 of the text (zlib -6: 10 MB/s at 0.295, -4: 60 MB/s at 0.308),
// ED lines 300 MB/s at 0.191 (zlib -4: 95 MB/s at 0.187).
#pragma once
#include <immintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace sigah {
namespace ldef {

static const int kMinMatch = 6;  // a match costs ~11-13 bits (length code, distance code + 6-7 extra bits); a base 2.25
static const int kMaxMatch = 258;

struct Match {
  uint32_t pos, len, dist;
};

// Code lengths of an optimal prefix code limited to maxbits, complete (Kraft sum exactly 1) whenever two or more
// symbols are in use; callers make sure of two.
inline void code_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* lens) {
  struct Leaf { uint32_t f; int s; };
  std::vector<Leaf> leaf;
  for (int i = 0; i < n; ++i) {
    lens[i] = 0;
    if (freq[i]) leaf.push_back({freq[i], i});
  }
  const int m = (int)leaf.size();
  if (m == 0) return;
  if (m == 1) { lens[leaf[0].s] = 1; return; }
  std::sort(leaf.begin(), leaf.end(), [](const Leaf& a, const Leaf& b) { return a.f != b.f ? a.f < b.f : a.s < b.s; });
  // two-queue Huffman: nodes 0..m-1 leaves (ascending), m..2m-2 internal in order of creation (ascending too)
  std::vector<uint64_t> w(2 * m - 1);
  std::vector<int> parent(2 * m - 1, -1);
  for (int i = 0; i < m; ++i) w[i] = leaf[i].f;
  int a = 0, b = m, next = m;
  auto take = [&]() -> int {
    if (a < m && (b >= next || w[a] <= w[b])) return a++;
    return b++;
  };
  while (next < 2 * m - 1) {
    const int x = take(), y = take();
    w[next] = w[x] + w[y];
    parent[x] = parent[y] = next;
    ++next;
  }
  std::vector<int> depth(2 * m - 1, 0);
  for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
  // how many leaves per length with the deep ones held at maxbits, then one unit of the Kraft sum back per move: a leaf
  // at the longest length below maxbits becomes the parent of itself and of one leaf taken from maxbits
  std::vector<int> cnt(maxbits + 2, 0);
  for (int i = 0; i < m; ++i) cnt[std::min(depth[i], maxbits)]++;
  uint64_t kraft = 0;
  for (int l = 1; l <= maxbits; ++l) kraft += (uint64_t)cnt[l] << (maxbits - l);
  for (uint64_t over = kraft - ((uint64_t)1 << maxbits); over > 0; --over) {
    int bits = maxbits - 1;
    while (cnt[bits] == 0) --bits;
    cnt[bits]--;
    cnt[bits + 1] += 2;
    cnt[maxbits]--;
  }
  int i = 0;  // rarest symbols take the longest codes
  for (int l = maxbits; l >= 1; --l)
    for (int c = cnt[l]; c > 0; --c) lens[leaf[i++].s] = (uint8_t)l;
}

// canonical codes (RFC 1951 3.2.2), bit-reversed: deflate sends Huffman codes most significant bit first in a stream
// that fills bytes from the least significant bit
inline void canonical_codes(const uint8_t* lens, int n, uint16_t* codes) {
  int cnt[16] = {0};
  for (int i = 0; i < n; ++i) cnt[lens[i]]++;
  cnt[0] = 0;
  unsigned next[16], code = 0;
  for (int l = 1; l <= 15; ++l) {
    code = (code + (unsigned)cnt[l - 1]) << 1;
    next[l] = code;
  }
  for (int i = 0; i < n; ++i) {
    const int l = lens[i];
    if (!l) { codes[i] = 0; continue; }
    unsigned c = next[l]++, r = 0;
    for (int k = 0; k < l; ++k) r |= ((c >> k) & 1u) << (l - 1 - k);
    codes[i] = (uint16_t)r;
  }
}

struct BitSink {
  std::string* out;
  uint64_t acc;
  int n;
  size_t at;
  explicit BitSink(std::string* o) : out(o), acc(0), n(0), at(0) {}
  void room(size_t more) {
    if (out->size() < at + more) out->resize(std::max(out->size() * 2, at + more + 4096));
  }
  inline void put(uint64_t v, int bits) {  // bits <= 32 with fewer than 32 pending
    acc |= v << n;
    n += bits;
    if (n >= 32) {
      memcpy(&(*out)[at], &acc, 4);  // little endian hosts only (x86-64, as the rest of this code base)
      at += 4;
      acc >>= 32;
      n -= 32;
    }
  }
  void align() {
    while (n > 0) {
      (*out)[at++] = (char)(acc & 0xFF);
      acc >>= 8;
      n -= 8;
    }
    acc = 0;
    n = 0;
  }
};

// length 3..258 -> symbol 257..285 with its extra bits; distance 1..32768 -> symbol 0..29 (RFC 1951 3.2.5)
struct LenCode { uint16_t sym; uint8_t extra_bits, extra; };
inline const LenCode* length_table() {
  static const LenCode* tab = [] {
    static LenCode t[259];
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t eb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int len = 3; len <= 258; ++len) {
      int s = 28;
      while (base[s] > len) --s;
      t[len] = {(uint16_t)(257 + s), eb[s], (uint8_t)(len - base[s])};
    }
    return t;
  }();
  return tab;
}
inline int length_symbol(int len, int* extra_bits, int* extra) {
  const LenCode& c = length_table()[len];
  *extra_bits = c.extra_bits;
  *extra = c.extra;
  return c.sym;
}
inline int dist_symbol(int dist, int* extra_bits, int* extra) {
  const unsigned v = (unsigned)dist - 1;
  if (v < 4) {
    *extra_bits = 0;
    *extra = 0;
    return (int)v;
  }
  const int nb = 31 - __builtin_clz(v);  // v in [2^nb, 2^(nb+1)): two symbols per power of two
  *extra_bits = nb - 1;
  *extra = (int)(v & ((1u << (nb - 1)) - 1));
  return 2 * nb + (int)((v >> (nb - 1)) & 1u);
}

// Matches against the previous line, field by field.  A line is cut at its tabs and blanks; field j of a line is compared
// with field j of the line above, from the separator on, and what is equal -- on into the following fields, over the line
// end and into the next line, as far as it goes -- is one match.  Aligned columns (VT lines: tag and running name prefix,
// the trailing tags) and columns that shift because a number before them grew a digit (ED lines: "... 149 150 0 ") are
// found alike; the bases of a read are never searched, only compared once with the read above.  Separators and line
// ends are located 16 bytes at a time (SSE2).
static const int kMaxFields = 24;
struct Fields {
  uint32_t at[kMaxFields];  // at[0] = start of the line, at[k] = position of its k-th separator
  int n;
};
// fields of the line that starts at `from`; returns the position of its line end (or n)
inline size_t split_line(const unsigned char* in, size_t n, size_t from, Fields* f) {
  f->at[0] = (uint32_t)from;
  f->n = 1;
  const __m128i nl = _mm_set1_epi8('\n'), tab = _mm_set1_epi8('\t'), blank = _mm_set1_epi8(' ');
  size_t p = from;
  for (; p + 16 <= n; p += 16) {
    const __m128i x = _mm_loadu_si128((const __m128i*)(in + p));
    const unsigned e = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(x, nl));
    unsigned sep = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(x, tab), _mm_cmpeq_epi8(x, blank)));
    if (e) sep &= (e & (0u - e)) - 1;  // separators before the line end only
    for (; sep && f->n < kMaxFields; sep &= sep - 1) f->at[f->n++] = (uint32_t)(p + (size_t)__builtin_ctz(sep));
    if (e) return p + (size_t)__builtin_ctz(e);
  }
  for (; p < n; ++p) {
    if (in[p] == '\n') return p;
    if ((in[p] == '\t' || in[p] == ' ') && f->n < kMaxFields) f->at[f->n++] = (uint32_t)p;
  }
  return n;
}
inline size_t common_run(const unsigned char* in, size_t n, size_t s, size_t src) {
  size_t run = 0;
  while (run < (size_t)kMaxMatch && s + run + 8 <= n) {
    uint64_t x, y;
    memcpy(&x, in + s + run, 8);
    memcpy(&y, in + src + run, 8);
    if (x != y) {
      run += (size_t)(__builtin_ctzll(x ^ y) >> 3);
      return std::min(run, (size_t)kMaxMatch);
    }
    run += 8;
  }
  while (run < (size_t)kMaxMatch && s + run < n && in[s + run] == in[src + run]) ++run;
  return std::min(run, (size_t)kMaxMatch);
}
// is a match of `run` bytes at s cheaper than its bytes as literals?  (a match: ~13 bits; a base 2.25, anything else ~4)
inline bool match_pays(const unsigned char* in, size_t s, size_t run) {
  if (run >= 7) return true;
  if (run < 4) return false;
  unsigned quarter_bits = 0;
  for (size_t k = 0; k < run; ++k) {
    const unsigned char c = in[s + k];
    quarter_bits += (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? 9u : 16u;
  }
  return quarter_bits > 52u;
}
// The line a line is compared with: the nearest of the four before it that starts with the same byte (the record above in
// FASTA -- header against header, two lines up -- and FASTQ, four lines up; in ASQG text every line starts like the one
// before), else the line before.
inline void find_matches(const unsigned char* in, size_t n, std::vector<Match>* ms) {
  ms->clear();
  if (n >= ((size_t)1 << 32)) return;  // positions are kept in 32 bits (the writer's blocks are 1 MiB): literals only
  Fields ring[5];
  int have = 0;   // lines before this one that are in the ring (at most 4)
  int cur = 0;    // ring slot of the current line
  size_t covered = 0;  // text below this position is inside a match already
  for (size_t ls = 0; ls < n;) {
    Fields& c = ring[cur];
    const size_t le = split_line(in, n, ls, &c);
    const Fields* ref = nullptr;
    for (int back = 1; back <= have; ++back) {
      const Fields& f = ring[(cur + 5 - back) % 5];
      if (in[f.at[0]] == in[ls]) {
        ref = &f;
        break;
      }
    }
    if (!ref && have) ref = &ring[(cur + 4) % 5];
    const int nf = ref ? std::min(c.n, ref->n) : 0;
    for (int j = 0; j < nf; ++j) {
      const size_t s = c.at[j];
      if (s < covered) continue;
      const size_t src = ref->at[j], d = s - src;
      if (d > 32768) break;
      const size_t run = common_run(in, n, s, src);
      if (!match_pays(in, s, run)) continue;
      ms->push_back({(uint32_t)s, (uint32_t)run, (uint32_t)d});
      covered = s + run;
    }
    cur = (cur + 1) % 5;
    if (have < 4) ++have;
    ls = le + 1;
  }
}

// One block of raw deflate for in[0, n): a dynamic-Huffman block, then -- unless `last` -- an empty stored block that
// brings the stream to a byte boundary (what zlib's Z_SYNC_FLUSH emits), so that blocks made apart concatenate.
inline void deflate_lines(const unsigned char* in, size_t n, bool last, std::string* out) {
  BitSink bs(out);
  bs.room(1024);
  if (n == 0) {
    if (last) {
      bs.put(1, 1);  // BFINAL, fixed codes, end of block
      bs.put(1, 2);
      bs.put(0, 7);
    } else {
      bs.put(0, 3);
      bs.align();
      bs.room(8);
      bs.put(0x0000, 16);
      bs.put(0xFFFF, 16);
    }
    bs.align();
    out->resize(bs.at);
    return;
  }
  std::vector<Match> ms;
  find_matches(in, n, &ms);
  // symbol counts: all bytes, minus the bytes the matches cover, plus the matches' symbols
  uint32_t h4[4][256];
  memset(h4, 0, sizeof(h4));
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    h4[0][in[i]]++;
    h4[1][in[i + 1]]++;
    h4[2][in[i + 2]]++;
    h4[3][in[i + 3]]++;
  }
  for (; i < n; ++i) h4[0][in[i]]++;
  uint32_t lf[286], df[30];
  memset(lf, 0, sizeof(lf));
  memset(df, 0, sizeof(df));
  for (int c = 0; c < 256; ++c) lf[c] = h4[0][c] + h4[1][c] + h4[2][c] + h4[3][c];
  for (const Match& m : ms) {
    for (uint32_t k = 0; k < m.len; ++k) lf[in[m.pos + k]]--;
    int eb, ex;
    lf[length_symbol((int)m.len, &eb, &ex)]++;
    df[dist_symbol((int)m.dist, &eb, &ex)]++;
  }
  lf[256] = 1;
  {  // two codes at least on either side (RFC 1951 3.2.7 allows one; two keeps the codes complete)
    int used = 0;
    for (int c = 0; c < 286; ++c) used += lf[c] != 0;
    if (used < 2) lf[lf[0] ? 1 : 0] = 1;
    used = 0;
    for (int c = 0; c < 30; ++c) used += df[c] != 0;
    for (int c = 0; used < 2 && c < 30; ++c)
      if (!df[c]) { df[c] = 1; ++used; }
  }
  uint8_t ll[286], dl[30];
  uint16_t lc[286], dc[30];
  code_lengths(lf, 286, 15, ll);
  code_lengths(df, 30, 15, dl);
  canonical_codes(ll, 286, lc);
  canonical_codes(dl, 30, dc);
  // header: code lengths of both alphabets, run-length coded (RFC 1951 3.2.7)
  int nl = 286, nd = 30;
  while (nl > 257 && ll[nl - 1] == 0) --nl;
  while (nd > 1 && dl[nd - 1] == 0) --nd;
  uint8_t seq[316];
  for (int c = 0; c < nl; ++c) seq[c] = ll[c];
  for (int c = 0; c < nd; ++c) seq[nl + c] = dl[c];
  const int ns = nl + nd;
  struct Cl { uint8_t sym, extra; };
  std::vector<Cl> cl;
  for (int p = 0; p < ns;) {
    int r = 1;
    while (p + r < ns && seq[p + r] == seq[p]) ++r;
    if (seq[p] == 0) {
      int left = r;
      while (left >= 11) { const int t = std::min(left, 138); cl.push_back({18, (uint8_t)(t - 11)}); left -= t; }
      if (left >= 3) { cl.push_back({17, (uint8_t)(left - 3)}); left = 0; }
      while (left-- > 0) cl.push_back({0, 0});
    } else {
      cl.push_back({seq[p], 0});
      int left = r - 1;
      while (left >= 3) { const int t = std::min(left, 6); cl.push_back({16, (uint8_t)(t - 3)}); left -= t; }
      while (left-- > 0) cl.push_back({seq[p], 0});
    }
    p += r;
  }
  uint32_t cf[19];
  memset(cf, 0, sizeof(cf));
  for (const Cl& c : cl) cf[c.sym]++;
  {
    int used = 0;
    for (int c = 0; c < 19; ++c) used += cf[c] != 0;
    for (int c = 0; used < 2 && c < 19; ++c)
      if (!cf[c]) { cf[c] = 1; ++used; }
  }
  uint8_t cll[19];
  uint16_t clc[19];
  code_lengths(cf, 19, 7, cll);
  canonical_codes(cll, 19, clc);
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int ncl = 19;
  while (ncl > 4 && cll[order[ncl - 1]] == 0) --ncl;
  bs.room(512);
  bs.put(last ? 1 : 0, 1);
  bs.put(2, 2);  // dynamic codes
  bs.put((unsigned)(nl - 257), 5);
  bs.put((unsigned)(nd - 1), 5);
  bs.put((unsigned)(ncl - 4), 4);
  for (int c = 0; c < ncl; ++c) bs.put(cll[order[c]], 3);
  for (const Cl& c : cl) {
    bs.put(clc[c.sym], cll[c.sym]);
    if (c.sym == 16) bs.put(c.extra, 2);
    else if (c.sym == 17) bs.put(c.extra, 3);
    else if (c.sym == 18) bs.put(c.extra, 7);
  }
  // pairs of literals per lookup: (code of a, then code of b) for the byte values in use
  struct Pair { uint32_t code; uint32_t bits; };
  static thread_local std::vector<Pair> pair;  // 65536 entries, the used ones rewritten for every block
  if (pair.empty()) pair.resize(65536);
  int present[256], np = 0;
  for (int c = 0; c < 256; ++c)
    if (ll[c]) present[np++] = c;
  for (int x = 0; x < np; ++x)
    for (int y = 0; y < np; ++y) {
      const int a = present[x], b = present[y];  // a is the first byte in memory = the low byte of the pair
      pair[(size_t)a | ((size_t)b << 8)] = {(uint32_t)lc[a] | ((uint32_t)lc[b] << ll[a]), (uint32_t)ll[a] + ll[b]};
    }
  {  // what the block's symbols will take is known to the bit: room for it once, no checks from here on
    uint64_t bits = 0;
    for (int c = 0; c < 286; ++c) bits += (uint64_t)lf[c] * ll[c];
    for (int c = 0; c < 30; ++c) bits += (uint64_t)df[c] * dl[c];
    bits += 18ull * ms.size();  // extra bits of a length (<= 5) and a distance (<= 13)
    bs.room((size_t)(bits / 8) + 1024);
  }
  auto literals = [&](size_t from, size_t to) {
    size_t p = from;
    if (to - from >= 16) {
      // the sink's state in locals for the loop: byte stores may alias anything, members would be reloaded per symbol
      uint64_t acc = bs.acc;
      int nb = bs.n;
      char* dst = &(*bs.out)[0] + bs.at;
      const Pair* __restrict tab = pair.data();
      for (; p + 4 <= to; p += 4) {  // two pairs (<= 60 bits), whole bytes flushed after each
        uint16_t v0, v1;
        memcpy(&v0, in + p, 2);
        memcpy(&v1, in + p + 2, 2);
        const Pair e0 = tab[v0], e1 = tab[v1];
        acc |= (uint64_t)e0.code << nb;
        nb += (int)e0.bits;
        memcpy(dst, &acc, 8);
        dst += nb >> 3;
        acc >>= (nb & ~7);
        nb &= 7;
        acc |= (uint64_t)e1.code << nb;
        nb += (int)e1.bits;
        memcpy(dst, &acc, 8);
        dst += nb >> 3;
        acc >>= (nb & ~7);
        nb &= 7;
      }
      bs.acc = acc;
      bs.n = nb;
      bs.at = (size_t)(dst - &(*bs.out)[0]);
    }
    for (; p < to; ++p) bs.put(lc[in[p]], ll[in[p]]);
  };
  size_t at = 0;
  for (const Match& m : ms) {
    literals(at, m.pos);
    int eb, ex;
    const int ls = length_symbol((int)m.len, &eb, &ex);
    bs.put(lc[ls], ll[ls]);
    if (eb) bs.put((unsigned)ex, eb);
    const int ds = dist_symbol((int)m.dist, &eb, &ex);
    bs.put(dc[ds], dl[ds]);
    if (eb) bs.put((unsigned)ex, eb);
    at = (size_t)m.pos + m.len;
  }
  literals(at, n);
  bs.put(lc[256], ll[256]);
  if (!last) {
    bs.put(0, 3);  // empty stored block: to the byte boundary, LEN 0, NLEN ~0
    bs.align();
    bs.put(0x0000, 16);
    bs.put(0xFFFF, 16);
  }
  bs.align();
  out->resize(bs.at);
}


// CRC-32 of the gzip trailer (RFC 1952), folded 64 bytes at a time with carry-less multiplies (the scheme of Intel's
// "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", constants for the reflected polynomial 0xEDB88320):
// zlib 1.2.11's table walk makes 1 GB/s per thread, which is what the line coder above makes, too.  `tail32` finishes
// (and does everything on a CPU without PCLMULQDQ): the caller passes zlib's crc32.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(const unsigned char* buf, size_t len, uint32_t state) {
  // len: a multiple of 16, at least 64; state: the running remainder (complement of the gzip value)
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
  const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
  __m128i x1 = _mm_loadu_si128((const __m128i*)(buf + 0)), x2 = _mm_loadu_si128((const __m128i*)(buf + 16));
  __m128i x3 = _mm_loadu_si128((const __m128i*)(buf + 32)), x4 = _mm_loadu_si128((const __m128i*)(buf + 48));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
  buf += 64;
  len -= 64;
  while (len >= 64) {
    const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
    const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
    x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
    x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
    x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i*)(buf + 0)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i*)(buf + 16)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i*)(buf + 32)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i*)(buf + 48)));
    buf += 64;
    len -= 64;
  }
  // four lanes into one, then whatever whole 16 bytes are left (a plain loop: lambdas do not inherit the target attribute)
  const __m128i rest[3] = {x2, x3, x4};
  for (int k = 0; k < 3 || len >= 16; ++k) {
    __m128i next;
    if (k < 3) {
      next = rest[k];
    } else {
      next = _mm_loadu_si128((const __m128i*)buf);
      buf += 16;
      len -= 16;
    }
    const __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, next), a);
  }
  // 128 -> 64 -> 32 bits (Barrett)
  const __m128i mask = _mm_setr_epi32(~0, 0, ~0, 0);
  __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
  t = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, mask);
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), t);
  t = _mm_and_si128(x1, mask);
  t = _mm_clmulepi64_si128(t, poly, 0x10);
  t = _mm_and_si128(t, mask);
  t = _mm_clmulepi64_si128(t, poly, 0x00);
  x1 = _mm_xor_si128(x1, t);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
// gzip CRC-32 of buf[0, n) continued from `crc` (0 to start); tail32(crc, p, k) = zlib's crc32
template <class Tail>
inline uint32_t crc32_fast(uint32_t crc, const unsigned char* buf, size_t n, Tail tail32) {
  static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  if (clmul && n >= 64) {
    const size_t body = n & ~(size_t)15;
    crc = ~crc32_fold(buf, body, ~crc);
    buf += body;
    n -= body;
  }
  return n ? tail32(crc, buf, n) : crc;
}

}  // namespace ldef
}  // namespace sigah
