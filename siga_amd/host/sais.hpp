// siga_amd/host/sais.hpp -- suffix array by induced sorting (Nong, Zhang, Chan 2009), own implementation.
//
// Used by `siga index` (reference: src/indexer.cpp:41-47, src/suffix_array_builder.cpp:472-674, default
// algorithm "sais2").  The order the reference's default builder was observed to produce is the plain suffix
// array of T = r0 $ r1 $ ... r(n-1) $ with one shared smallest sentinel, comparisons running on past the
// sentinels and end-of-text smallest (SURVEY.md App. C "model B"); that is exactly what a textbook suffix
// array of T followed by a unique terminator gives, so no sentinel-specific comparator is needed.
#ifndef SIGA_AMD_HOST_SAIS_HPP_
#define SIGA_AMD_HOST_SAIS_HPP_

#include <cstdint>
#include <vector>

namespace sigah {

template <typename I>
struct SaisBits {
  std::vector<uint64_t> w;
  explicit SaisBits(I n) : w(((uint64_t)n + 63) / 64, 0) {}
  bool get(I i) const { return (w[(uint64_t)i >> 6] >> ((uint64_t)i & 63)) & 1u; }
  void set(I i, bool v) {
    uint64_t m = 1ull << ((uint64_t)i & 63);
    if (v) w[(uint64_t)i >> 6] |= m;
    else w[(uint64_t)i >> 6] &= ~m;
  }
};

template <typename Ch, typename I>
static void sais_buckets(const Ch* s, std::vector<I>& bkt, I n, I K, bool end) {
  for (I i = 0; i < K; ++i) bkt[i] = 0;
  for (I i = 0; i < n; ++i) ++bkt[s[i]];
  I sum = 0;
  for (I i = 0; i < K; ++i) {
    sum += bkt[i];
    bkt[i] = end ? sum : sum - bkt[i];
  }
}

template <typename Ch, typename I>
static void sais_induce(const SaisBits<I>& t, I* SA, const Ch* s, std::vector<I>& bkt, I n, I K) {
  // L-type suffixes, left to right
  sais_buckets(s, bkt, n, K, false);
  for (I i = 0; i < n; ++i) {
    I j = SA[i];
    if (j > 0 && !t.get(j - 1)) SA[bkt[s[j - 1]]++] = j - 1;
  }
  // S-type suffixes, right to left
  sais_buckets(s, bkt, n, K, true);
  for (I i = n - 1; i >= 0; --i) {
    I j = SA[i];
    if (j > 0 && t.get(j - 1)) SA[--bkt[s[j - 1]]] = j - 1;
  }
}

// s[0..n-1] over alphabet [0,K), s[n-1] == 0 the unique smallest character.  I must be signed.
template <typename Ch, typename I>
void sais(const Ch* s, I* SA, I n, I K) {
  if (n == 1) {
    SA[0] = 0;
    return;
  }
  SaisBits<I> t(n);  // 1 = S-type
  t.set(n - 1, true);
  t.set(n - 2, false);
  for (I i = n - 3; i >= 0; --i) t.set(i, s[i] < s[i + 1] || (s[i] == s[i + 1] && t.get(i + 1)));
  auto isLMS = [&](I i) { return i > 0 && t.get(i) && !t.get(i - 1); };

  std::vector<I> bkt((size_t)K);
  // stage 1: sort the LMS substrings
  sais_buckets(s, bkt, n, K, true);
  for (I i = 0; i < n; ++i) SA[i] = -1;
  for (I i = 1; i < n; ++i)
    if (isLMS(i)) SA[--bkt[s[i]]] = i;
  sais_induce(t, SA, s, bkt, n, K);

  I n1 = 0;
  for (I i = 0; i < n; ++i)
    if (isLMS(SA[i])) SA[n1++] = SA[i];
  for (I i = n1; i < n; ++i) SA[i] = -1;
  I name = 0, prev = -1;
  for (I i = 0; i < n1; ++i) {
    I pos = SA[i];
    bool diff = false;
    for (I d = 0; d < n; ++d) {
      if (prev == -1 || s[pos + d] != s[prev + d] || t.get(pos + d) != t.get(prev + d)) {
        diff = true;
        break;
      } else if (d > 0 && (isLMS(pos + d) || isLMS(prev + d))) {
        break;
      }
    }
    if (diff) {
      ++name;
      prev = pos;
    }
    SA[n1 + pos / 2] = name - 1;
  }
  for (I i = n - 1, j = n - 1; i >= n1; --i)
    if (SA[i] >= 0) SA[j--] = SA[i];

  // stage 2: the reduced problem
  I* SA1 = SA;
  I* s1 = SA + n - n1;
  if (name < n1) {
    sais<I, I>(s1, SA1, n1, name);
  } else {
    for (I i = 0; i < n1; ++i) SA1[s1[i]] = i;
  }

  // stage 3: induce the result
  sais_buckets(s, bkt, n, K, true);
  for (I i = 1, j = 0; i < n; ++i)
    if (isLMS(i)) s1[j++] = i;
  for (I i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
  for (I i = n1; i < n; ++i) SA[i] = -1;
  for (I i = n1 - 1; i >= 0; --i) {
    I j = SA[i];
    SA[i] = -1;
    SA[--bkt[s[j]]] = j;
  }
  sais_induce(t, SA, s, bkt, n, K);
}

}  // namespace sigah

#endif
