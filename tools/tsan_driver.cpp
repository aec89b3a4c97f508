// tools/tsan_driver.cpp -- the host library's threaded paths under ThreadSanitizer without a Python in between (a preloaded
// libtsan under CPython hangs): built by tools/sanitize_host.sh against the -fsanitize=thread libsiga_host.so.  Writes a reads
// file with names of mixed lengths and comments, then runs: the parallel loader (modes 0, 2: chunk parse + join, sample-sort
// name ranks), the text side of OverlapBuilder::build (VT lines, raw-pointer ED formatter, block-parallel gzip writer with the
// line deflate coder), the gz writer in pieces, the host index builder (SA-IS on 1-2 threads, bucket sort on 4, `-a sais`
// order) -- each on several threads; results are compared between thread counts.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../include/sigax.h"
extern "C" {
int64_t sigah_parse_file(const char* path, int mode, const char* out_path, int threads);
int64_t sigah_format_asqg(const char* path, const uint8_t* substring, const sigax_edge* edges, uint64_t n_edges, uint64_t min_overlap,
                          const char* out_path, int threads);
int sigah_write_file(const char* path, const char* data, uint64_t n, uint64_t pieces);
int sigah_index_file(const char* reads_path, const char* prefix, int threads, char* err, uint64_t errcap);
int sigah_index_file_sais(const char* reads_path, const char* prefix, int threads, int do_fwd, int do_rev, char* err, uint64_t errcap);
}
static std::string slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  std::mt19937_64 rng(7);
  const int n = 60000;
  std::string fa;
  std::vector<uint32_t> len(n);
  for (int i = 0; i < n; ++i) {
    fa += i % 3 ? ">r" + std::to_string(i) : ">a_much_longer_read_name_" + std::to_string(i) + std::string(rng() % 40, 'x');
    if (i % 5 == 0) fa += " CR:i:" + std::to_string(rng() % 90) + " BX:Z:ACGT-1";
    fa += "\n";
    len[i] = 40 + rng() % 120;
    for (uint32_t k = 0; k < len[i]; ++k) fa += "ACGT"[rng() & 3];
    fa += "\n";
  }
  const std::string reads = dir + "/tsan_reads.fa";
  { std::ofstream f(reads, std::ios::binary); f << fa; }
  int bad = 0;
  auto check = [&](bool ok, const char* what) { if (!ok) { fprintf(stderr, "FAILED: %s\n", what); ++bad; } };
  for (int mode : {0, 2}) {
    std::string ref;
    for (int t : {1, 3, 8}) {
      const std::string out = dir + "/tsan_parse_" + std::to_string(mode) + "_" + std::to_string(t);
      check(sigah_parse_file(reads.c_str(), mode, out.c_str(), t) == n, "parse count");
      const std::string got = slurp(out);
      if (t == 1) ref = got; else check(got == ref, "parse output independent of threads");
    }
  }
  std::vector<sigax_edge> ed(200000);
  for (auto& e : ed) {
    e.query = (uint32_t)(rng() % n);
    e.target = (uint32_t)(rng() % n);
    e.length = 1 + (uint32_t)(rng() % std::min(len[e.query], len[e.target]));
    e.af = (uint32_t)(rng() & 7);
  }
  std::vector<uint8_t> sub(n);
  for (auto& s : sub) s = rng() % 10 == 0;
  std::string ref;
  for (int t : {1, 4, 8}) {
    for (const char* ext : {".asqg", ".asqg.gz"}) {
      const std::string out = dir + "/tsan_fmt_" + std::to_string(t) + ext;
      check(sigah_format_asqg(reads.c_str(), sub.data(), ed.data(), ed.size(), 45, out.c_str(), t) == n, "format_asqg");
      if (std::string(ext) == ".asqg") { const std::string got = slurp(out); if (t == 1) ref = got; else check(got == ref, "ASQG text independent of threads"); }
    }
  }
  {  // VT lines ahead of their flags (its own threads + the writer's thread) and the caller-written file: the same bytes
    const std::string base = slurp(dir + "/tsan_fmt_4.asqg.gz");
    auto again = [&](int t, const char* what) {
      const std::string out = dir + "/tsan_ahead.asqg.gz";
      check(sigah_format_asqg(reads.c_str(), sub.data(), ed.data(), ed.size(), 45, out.c_str(), t) == n, what);
      check(slurp(out) == base, what);
    };
    setenv("SIGA_VT_AHEAD", "1", 1);
    setenv("SIGA_BATCH_READS", "3001", 1);
    again(1, "VT lines ahead, 1 thread");
    again(4, "VT lines ahead, 4 threads");
    setenv("SIGA_VT_AHEAD_BYTES", "1", 1);
    again(3, "VT lines ahead, one wave at a time");
    unsetenv("SIGA_VT_AHEAD_BYTES");
    unsetenv("SIGA_VT_AHEAD");
    unsetenv("SIGA_BATCH_READS");
    setenv("SIGA_SYNC_WRITE", "1", 1);
    again(4, "file written by the caller");
    unsetenv("SIGA_SYNC_WRITE");
  }
  check(sigah_write_file((dir + "/tsan_w.gz").c_str(), fa.data(), fa.size(), 37) == 0, "gz writer in pieces");
  char err[512];
  std::string ib[4];
  int k = 0;
  for (int t : {1, 4}) {
    const std::string pre = dir + "/tsan_ix" + std::to_string(t);
    check(sigah_index_file(reads.c_str(), pre.c_str(), t, err, sizeof err) == 0, "index_file");
    ib[k++] = slurp(pre + ".bwt") + slurp(pre + ".rsai");
  }
  check(ib[0] == ib[1], "SA-IS and the threaded bucket sort write the same index");
  check(sigah_index_file_sais(reads.c_str(), (dir + "/tsan_sais").c_str(), 4, 1, 1, err, sizeof err) == 0, "index -a sais");
  printf("tsan_driver: %s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}
