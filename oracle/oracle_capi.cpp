// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE (see siga_oracle.hpp header).  C entry points so that
// tests/, smoke() and bench.py's cpu_baseline leg can drive the CPU restatement through ctypes.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "siga_oracle.hpp"

using namespace oracle;

struct OrcIndex {
  BuiltIndex b;
};

static std::vector<std::string> unpack(const char* seqs, const uint64_t* offs, uint64_t n) {
  std::vector<std::string> r(n);
  for (uint64_t i = 0; i < n; ++i) r[i].assign(seqs + offs[i], seqs + offs[i + 1]);
  return r;
}

extern "C" {

// Build an index (.bwt + .sai content) from reads; reverse != 0 reverses every read first (indexer.cpp:60-64).
void* orc_index_build(const char* seqs, const uint64_t* offs, uint64_t n, int reverse) {
  std::vector<std::string> reads = unpack(seqs, offs, n);
  if (reverse)
    for (auto& r : reads) std::reverse(r.begin(), r.end());
  OrcIndex* ix = new OrcIndex();
  buildIndex(reads, &ix->b);
  return ix;
}

void* orc_index_load(const char* bwt_path, const char* sai_path) {
  std::unique_ptr<OrcIndex> ix(new OrcIndex());
  if (!ix->b.fm.load(bwt_path)) return nullptr;
  if (sai_path && sai_path[0]) {
    std::ifstream in(sai_path);
    if (!in || !loadSai(in, &ix->b.sai)) return nullptr;
  }
  return ix.release();
}

int orc_index_save(void* h, const char* bwt_path, const char* sai_path) {
  OrcIndex* ix = (OrcIndex*)h;
  {
    std::ofstream out(bwt_path, std::ios::binary);
    if (!out) return -1;
    ix->b.fm.save(out);
  }
  {
    std::ofstream out(sai_path);
    if (!out) return -1;
    out << saiText(ix->b.sai);
  }
  return 0;
}

void orc_index_free(void* h) { delete (OrcIndex*)h; }
uint64_t orc_length(void* h) { return ((OrcIndex*)h)->b.fm.length(); }
uint64_t orc_nruns(void* h) { return ((OrcIndex*)h)->b.fm.runs.size(); }
uint64_t orc_nstrings(void* h) { return ((OrcIndex*)h)->b.fm.nstrings; }
const uint8_t* orc_runs(void* h) { return ((OrcIndex*)h)->b.fm.runs.data(); }
uint64_t orc_sai_size(void* h) { return ((OrcIndex*)h)->b.sai.size(); }
const uint32_t* orc_sai(void* h) { return ((OrcIndex*)h)->b.sai.data(); }

void orc_occ(void* h, uint64_t i, uint64_t* out5) {
  AlphaCount c = ((OrcIndex*)h)->b.fm.getOcc(i);
  memcpy(out5, c.v, sizeof(c.v));
}
void orc_pred(void* h, uint64_t* out5) {
  const FMIndex& fm = ((OrcIndex*)h)->b.fm;
  for (int k = 0; k < 5; ++k) out5[k] = fm.getPC(tochar(k));
}
int orc_getchar(void* h, uint64_t i) { return ((OrcIndex*)h)->b.fm.getChar(i); }
uint64_t orc_occurrences(void* h, const char* w, uint64_t len) {
  return Interval::occurrences(std::string(w, len), &((OrcIndex*)h)->b.fm);
}

// Blocks are returned as 10 x u64: capped0 lo/hi, capped1 lo/hi, raw0 lo/hi, raw1 lo/hi, length, af bits
// (hits order, overlap_builder.cpp:198-201).  Returns the number of blocks (may exceed cap; only cap are written).
int64_t orc_overlap(void* hf, void* hr, const char* seq, uint64_t len, uint64_t min_overlap, int irreducible, int rc,
                    uint64_t* out, uint64_t cap, int* substring, uint64_t* stats2) {
  OrcIndex* f = (OrcIndex*)hf;
  OrcIndex* r = (OrcIndex*)hr;
  OccStats st;
  f->b.fm.stats = &st;
  r->b.fm.stats = &st;
  OverlapBuilder builder(&f->b.fm, &r->b.fm, irreducible != 0, rc != 0, &st);
  OverlapBlockList blocks;
  OverlapResult res = builder.overlap(std::string(seq, len), min_overlap, &blocks);
  f->b.fm.stats = nullptr;
  r->b.fm.stats = nullptr;
  if (substring) *substring = res.substring ? 1 : 0;
  if (stats2) { stats2[0] = st.calls; stats2[1] = st.nmin; }
  uint64_t k = 0;
  for (auto& b : blocks) {
    if (k < cap) {
      uint64_t* o = out + 10 * k;
      o[0] = b.capped[0].lower; o[1] = b.capped[0].upper; o[2] = b.capped[1].lower; o[3] = b.capped[1].upper;
      o[4] = b.raw[0].lower; o[5] = b.raw[0].upper; o[6] = b.raw[1].lower; o[7] = b.raw[1].upper;
      o[8] = b.length; o[9] = b.af.bits;
    }
    ++k;
  }
  return (int64_t)k;
}

// Whole `siga overlap` at -t 1 (overlap_builder.cpp:423-509): reads file -> ASQG text (+ hits text).
// stats3 = {occ calls, N_occ_min, blocks}.
int orc_build_asqg(void* hf, void* hr, const char* reads_path, uint64_t min_overlap, int irreducible, int rc,
                   const char* asqg_path, const char* hits_path, uint64_t* stats3) {
  OrcIndex* f = (OrcIndex*)hf;
  OrcIndex* r = (OrcIndex*)hr;
  std::ifstream in(reads_path);
  if (!in) return -1;
  std::vector<DNASeq> reads;
  if (!readSequences(in, &reads)) return -2;
  std::ofstream asqg(asqg_path);
  if (!asqg) return -3;
  std::unique_ptr<std::ofstream> hits;
  if (hits_path && hits_path[0]) hits.reset(new std::ofstream(hits_path));
  OccStats st;
  u64 nb = 0;
  buildASQG(reads, f->b.fm, r->b.fm, f->b.sai, r->b.sai, min_overlap, irreducible != 0, rc != 0, asqg, hits.get(), &st, &nb);
  if (stats3) { stats3[0] = st.calls; stats3[1] = st.nmin; stats3[2] = nb; }
  return 0;
}

// Whole `siga overlap` at -t N as the reference runs it (overlap_builder.cpp:439-483): overlap() of all reads under OpenMP
// (parallel_framework.h:38), then the serial post-processing in input order: VT lines, hit conversion, ED lines.  Plain
// text out (the reference's gzip filter is left out).  secs3 = {read + parse, overlap (parallel), VT + ED text}; the ED
// order equals the -t 1 order here because hits are converted per read in input order.
int orc_build_asqg_mt(void* hf, void* hr, const char* reads_path, uint64_t min_overlap, int irreducible, int rc,
                      const char* asqg_path, int threads, double* secs3) {
  OrcIndex* f = (OrcIndex*)hf;
  OrcIndex* r = (OrcIndex*)hr;
  auto t0 = std::chrono::steady_clock::now();
  std::ifstream in(reads_path);
  if (!in) return -1;
  std::vector<DNASeq> reads;
  if (!readSequences(in, &reads)) return -2;
  auto t1 = std::chrono::steady_clock::now();
  std::vector<OverlapBlockList> all(reads.size());
  std::vector<uint8_t> sub(reads.size(), 0);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 64)
#endif
  for (int64_t i = 0; i < (int64_t)reads.size(); ++i) {
    OverlapBuilder builder(&f->b.fm, &r->b.fm, irreducible != 0, rc != 0, nullptr);
    sub[i] = builder.overlap(reads[i].seq, min_overlap, &all[i]).substring ? 1 : 0;
  }
  auto t2 = std::chrono::steady_clock::now();
  std::ofstream asqg(asqg_path);
  if (!asqg) return -3;
  asqg << "HT\tVN:i:1\tOL:i:" << (int)min_overlap << "\tCN:i:1\n";
  std::vector<ReadInfo> info(reads.size());
  for (size_t i = 0; i < reads.size(); ++i) {
    writeVertex(asqg, reads[i], sub[i] != 0);
    info[i].name = reads[i].name;
    info[i].length = reads[i].seq.length();
  }
  for (size_t i = 0; i < reads.size(); ++i) convertHit(i, all[i], info, f->b.sai, r->b.sai, &asqg);
  asqg.close();
  auto t3 = std::chrono::steady_clock::now();
  if (secs3) {
    secs3[0] = std::chrono::duration<double>(t1 - t0).count();
    secs3[1] = std::chrono::duration<double>(t2 - t1).count();
    secs3[2] = std::chrono::duration<double>(t3 - t2).count();
  }
  return 0;
}

// `siga rmdup` at -t 1 (overlap_builder.cpp:562-704)
int orc_rmdup(void* hf, void* hr, const char* reads_path, const char* fasta_path, const char* dups_path) {
  OrcIndex* f = (OrcIndex*)hf;
  OrcIndex* r = (OrcIndex*)hr;
  std::ifstream in(reads_path);
  if (!in) return -1;
  std::vector<DNASeq> reads;
  if (!readSequences(in, &reads)) return -2;
  std::ofstream fa(fasta_path), du(dups_path);
  if (!fa || !du) return -3;
  rmdupText(reads, f->b.fm, r->b.fm, f->b.sai, r->b.sai, fa, du);
  return 0;
}

// `siga correct` (k-mer algorithm) at -t 1: correct_processor.cpp:270-317.  stats2 = {reads written, reads changed}
int orc_correct(void* hf, const char* reads_path, const char* out_path, uint64_t k, int threshold, uint64_t rounds,
                uint64_t offset, uint64_t* stats2) {
  OrcIndex* f = (OrcIndex*)hf;
  std::ifstream in(reads_path);
  if (!in) return -1;
  SeqReader reader(in);
  if (!reader.ok()) return -2;
  std::ofstream out(out_path);
  if (!out) return -3;
  CorrectParams P;
  P.kmerSize = k; P.minSupport = threshold; P.maxAttempts = rounds; P.countOffset = offset;
  DNASeq rd;  // one object reused across reads, as DNASeqWorkItemGenerator does (kseq.h:176-183)
  uint64_t written = 0, changed = 0;
  while (reader.read(rd)) {
    std::string seq;
    if (correctRead(f->b.fm, P, rd, &seq)) {
      if (seq != rd.seq) ++changed;
      DNASeq w = rd;
      w.seq = seq;
      writeSeq(out, w);
      ++written;
    }
  }
  if (stats2) { stats2[0] = written; stats2[1] = changed; }
  return 0;
}

// CPU baseline leg: OverlapBuilder::overlap over a batch of reads, OpenMP over reads like
// parallel_framework.h:38.  Returns seconds; out3 = {blocks, substring reads, N_occ_min}.
double orc_overlap_batch_timed(void* hf, void* hr, const char* seqs, const uint64_t* offs, uint64_t n,
                               uint64_t min_overlap, int irreducible, int rc, int threads, uint64_t* out3) {
  OrcIndex* f = (OrcIndex*)hf;
  OrcIndex* r = (OrcIndex*)hr;
  uint64_t blocks = 0, subs = 0, nmin = 0;
  auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : blocks, subs, nmin)
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    OccStats st;
    OverlapBuilder builder(&f->b.fm, &r->b.fm, irreducible != 0, rc != 0, &st);
    OverlapBlockList bl;
    OverlapResult res = builder.overlap(std::string(seqs + offs[i], seqs + offs[i + 1]), min_overlap, &bl);
    blocks += bl.size();
    subs += res.substring ? 1 : 0;
    nmin += st.nmin;
  }
  auto t1 = std::chrono::steady_clock::now();
  if (out3) { out3[0] = blocks; out3[1] = subs; out3[2] = nmin; }
  return std::chrono::duration<double>(t1 - t0).count();
}

// OverlapBuilder::overlap (mode 0) or OverlapBuilder::duplicate (mode 1, overlap_builder.cpp:1184-1195) for a batch of
// reads, OpenMP over reads, all blocks returned: block_offs[n+1]; blocks[10 * k ..] as orc_overlap lays them out, up to
// `cap` blocks (the return value is the number needed); substring[n]; *nmin = N_occ_min summed.
int64_t orc_overlap_batch(void* hf, void* hr, const char* seqs, const uint64_t* offs, uint64_t n, uint64_t min_overlap,
                          int irreducible, int rc, int mode, int threads, uint64_t* block_offs, uint64_t* blocks, uint64_t cap,
                          uint8_t* substring, uint64_t* nmin_out) {
  OrcIndex* f = (OrcIndex*)hf;
  OrcIndex* r = (OrcIndex*)hr;
  std::vector<std::vector<uint64_t>> per(n);
  uint64_t nmin = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : nmin)
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    OccStats st;
    OverlapBuilder builder(&f->b.fm, &r->b.fm, irreducible != 0, rc != 0, &st);
    OverlapBlockList bl;
    std::string seq(seqs + offs[i], seqs + offs[i + 1]);
    OverlapResult res = mode ? builder.duplicate(seq, &bl) : builder.overlap(seq, min_overlap, &bl);
    substring[i] = res.substring ? 1 : 0;
    nmin += st.nmin;
    std::vector<uint64_t>& o = per[i];
    o.reserve(bl.size() * 10);
    for (auto& b : bl) {
      o.push_back(b.capped[0].lower); o.push_back(b.capped[0].upper); o.push_back(b.capped[1].lower); o.push_back(b.capped[1].upper);
      o.push_back(b.raw[0].lower); o.push_back(b.raw[0].upper); o.push_back(b.raw[1].lower); o.push_back(b.raw[1].upper);
      o.push_back(b.length); o.push_back(b.af.bits);
    }
  }
  uint64_t k = 0;
  for (uint64_t i = 0; i < n; ++i) {
    block_offs[i] = k;
    const uint64_t nb = per[i].size() / 10;
    if (k + nb <= cap) memcpy(blocks + 10 * k, per[i].data(), per[i].size() * 8);
    k += nb;
  }
  block_offs[n] = k;
  if (nmin_out) *nmin_out = nmin;
  return (int64_t)k;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// KAT helpers (test/index_test.cpp:33-84, test/utils_test.cpp:32-36, test/preprocess_test.cpp:30-43)
uint64_t orc_rl_encode(const char* s, uint64_t len, uint8_t* out, uint64_t cap) {
  std::vector<uint8_t> runs;
  RLEncoder enc(&runs);
  for (uint64_t i = 0; i < len; ++i) enc.push(s[i]);
  enc.finish();
  for (uint64_t i = 0; i < runs.size() && i < cap; ++i) out[i] = runs[i];
  return runs.size();
}
int orc_torank(int c) { return torank((char)c); }
int orc_tochar(int r) { return tochar((size_t)r); }
void orc_stem(const char* path, char* out, uint64_t cap) {
  std::string s = stem(path);
  snprintf(out, cap, "%s", s.c_str());
}
void orc_revcomp(const char* s, uint64_t len, int mode, char* out) {  // 0 reverse, 1 complement, 2 revcomp
  std::string in(s, len), r;
  r = mode == 0 ? reverse_copy(in) : mode == 1 ? complement_copy(in) : reverse_complement_copy(in);
  memcpy(out, r.data(), len);
}
// asqg.h:31-80: returns 1 if parsed; writes tostring(key) of the parsed value
int orc_tag_roundtrip(const char* text, int type, const char* key, char* out, uint64_t cap) {
  std::string s;
  bool ok = false;
  if (type == 'i') { TagValue<int> t; ok = t.fromstring(text); if (ok) s = t.tostring(key); }
  else if (type == 'f') { TagValue<float> t; ok = t.fromstring(text); if (ok) s = t.tostring(key); }
  else { TagValue<std::string> t; ok = t.fromstring(text); if (ok) s = t.tostring(key); }
  snprintf(out, cap, "%s", s.c_str());
  return ok ? 1 : 0;
}

}  // extern "C"
