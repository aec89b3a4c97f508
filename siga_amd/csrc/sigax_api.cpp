// siga_amd/csrc/sigax_api.cpp -- implementation of include/sigax.h: index lifetime, device workspaces and the
// launch sequence of the overlap path.  Compiled with hipcc together with sigax_kernels.hip into libsigax.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sigax_kernels.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// ------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// the same for the library's other translation units (sigax_index_build.hip)
int sigax_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(SIGAX_E_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" const char* sigax_last_error(void) { return g_err; }

extern "C" int sigax_stream_create(int device, void** stream) {
  if (!stream) return fail(SIGAX_E_ARG, "NULL argument");
  *stream = nullptr;
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return SIGAX_OK;
}

extern "C" void sigax_stream_destroy(int device, void* stream) {
  if (!stream) return;
  if (hipSetDevice(device) == hipSuccess) hipStreamDestroy((hipStream_t)stream);
}

extern "C" int sigax_device_count(int* n) {
  if (!n) return fail(SIGAX_E_ARG, "n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return fail(SIGAX_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n = c;
  return SIGAX_OK;
}

// ------------------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------------------
struct sigax_index {
  int device;
  bool wide;
  FmStrand st[2];  // 0 forward (.bwt), 1 reverse (.rbwt); pointers are device pointers
  void* d_gran[2];
  void* d_gran2[2];  // two-step tables (fm_layout.h) or NULL
  void* d_super2[2]; // ... their superblock bases (64-bit positions) or NULL
  void* d_sa[2];     // row tables (fm_layout.h) or NULL
  void* d_text[2];   // stretch texts (fm_layout.h) or NULL
  void* d_xmap[2];   // direct maps (fm_layout.h) or NULL
  u64 sa_alloc[2], text_alloc[2];  // bytes allocated for them
  // The row tables of an index of 2^26 symbols and more are built by a side thread while the caller goes on (at BASELINE
  // configs[1] 0.1 s: more than the whole one-batch `siga overlap` spends on the GPU); runs enqueued before they are ready
  // use the forms without them -- same bytes out.  0 none / published, 1 being built, 2 built: tab_st waits for publishing.
  // Nor are they started before the index has been asked for as many reads as it holds (build_rowend): one pass of the
  // CLI would pay for tables that cost it more than they save.  tab_plan = bytes of the tables still to be allocated (sigax_batch_size_hint leaves them free).
  std::thread* tab_thread;
  std::atomic<int>* tab_state;
  FmStrand tab_st[2];
  u64 tab_bytes, tab_plan, reads_asked;
  bool tab_tried;  // the build has been started once (it is not tried again when memory was short)
  bool tab_text;
  bool tab_direct;   // direct maps instead of row tables (.sai tables present, ACGT-only reads)
  u32 tab_syms;      // symbols a row-table entry carries (plan)
  u32 max_read_len;  // longest read of sigax_index_set_reads (0: not told yet), an upper bound of the longest stretch
  void* d_super[2];
  void* d_start[2];  // start tables of the block finder (fm_layout.h) or NULL
  // Deep start tables (fm_layout.h): built by sigax_index_prepare_overlap, or on a side thread once the index is being
  // reused, for the min-overlap of the run at hand; deep_state 0 none / published, 1 being built, 2 built (deep_new waits
  // for publishing under enqueue_mu).  d_slen = the stretches' lengths by '$' rank (kept from the row tables' build).
  void* d_deep[2];
  u64 deep_slots[2], deep_bytes;
  uint32_t deep_k;
  uint32_t* d_slen[2];
  std::thread* deep_thread;
  std::atomic<int>* deep_state;
  void* deep_new[2];
  u64 deep_new_slots[2], deep_new_bytes;
  uint32_t deep_new_k;
  bool deep_tried;
  uint32_t ptab_k;
  void* d_ptab;      // intervals of all 12-mers of the forward index: `siga correct`'s k-mer lookups start there (built by the
  bool ptab_tried;   // first correction call; SIGAX_KMER_PREFIX=0: never)
  hipEvent_t ptab_ev;  // recorded behind the table's build on the first call's stream
  // `siga correct`'s k-mer table: the deep start table of the forward strand for K = the corrector's k (fm_layout.h) -- every
  // distinct k-mer of the reads with its number of occurrences, so FMIndex::Interval::occurrences (src/fmindex.h:80-86) of a
  // k-mer is ONE lookup, and a k-mer that is not in the table does not occur.  Built by the first correction call with that k
  // (ensure_kmer_table), from a forward row table + text of its own when the index has none.
  void* d_ktab;
  u64 ktab_slots, ktab_bytes;
  uint32_t ktab_k, ktab_tried_k;
  uint32_t csa_bits, cld_bits, ct_bits, ctext_stride;
  void *d_csa, *d_ctext;  // forward row table + stretch text built for it (fwd_only indexes, or before the extractor's exist)
  uint32_t* d_cslen;
  uint32_t* d_sai[2];
  u64 n_sai;
  uint32_t* d_read_len;
  uint32_t* d_name_rank;
  u64 n_meta;
  u64 n_symbols, n_strings, device_bytes;
  // The internal pipeline streams belong to the index, not to a batch: every batch on this index queues its finder
  // launches on s_find and its filter/extract launches on s_fx, so with two batches in flight batch B's first finder
  // launch runs beside batch A's last filter/extract launch and finder launches never run beside each other.
  hipStream_t s_find, s_fx, s_tail;
  hipStream_t s_ord;  // the locality ordering of a batch (a key kernel + some twenty launches of the radix sort, 1 ms of work
                      // per 2.5 M reads): high priority -- queued on the caller's stream beside the long kernels of the
                      // batches in flight it took 17 ms at the BASELINE configs[2] shape, all of it on the batch's own chain
  std::mutex* enqueue_mu;
  int n_cu;  // compute units of the device
  // The longest chain of candidate blocks any run on this index has produced so far.  The candidate arena gives every chain
  // that many slots plus headroom instead of the worst case (one per overlap length): BASELINE configs[1] 11 records per
  // chain on average, 30-odd at most, 106 in the worst case.  A run whose chains outgrow their slots is repeated with what
  // it reported (sigax_batch_finish).
  std::atomic<uint32_t>* cap_seen;
  bool split_strands;  // two-step tables too large to gather from both at once: one finder launch per strand
  bool fwd_only;       // opened without the reverse strand (what `siga correct` needs: src/correct.cpp loads <prefix>.bwt alone)
};

// RL units (src/rlstring.h:10-63) -> 64-byte rank granules (fm_layout.h), decoded on the device (sigax_index_build.hip)
int sigax_decode_strand(const uint8_t* runs, u64 n_runs, u64 nsym, bool wide, void** d_gran, u64* gran_bytes, void** d_super,
                        u64* super_bytes, u64 C[5], u64 total[5]);

// frees device buffers on every exit path of the one-shot calls below
struct DevGuard {
  std::vector<void*> ptrs;
  ~DevGuard() {
    for (void* p : ptrs)
      if (p) hipFree(p);
  }
  hipError_t alloc(void** out, size_t bytes) {
    *out = nullptr;
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e == hipSuccess) ptrs.push_back(*out);
    return e;
  }
};

// Whole file into memory; files of 64 MiB and more in slices on several threads (pread): one thread copies out of the
// page cache at 2 GB/s, and BASELINE configs[2]'s four index files are 3 GB.
static int read_file(const char* path, std::vector<uint8_t>* out) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(SIGAX_E_IO, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return fail(SIGAX_E_IO, "cannot stat %s", path);
  }
  const size_t n = st.st_size > 0 ? (size_t)st.st_size : 0;
  try {
    out->resize(n);
  } catch (...) {  // no exception crosses the C boundary
    close(fd);
    return fail(SIGAX_E_IO, "%s: no memory for its %zu bytes", path, n);
  }
  const unsigned nt = n >= (64u << 20) ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u) : 1u;
  std::vector<int> shortread(nt, 0);
  auto slice = [&](unsigned k) {
    size_t at = n * k / nt;
    const size_t end = n * (k + 1) / nt;
    while (at < end) {
      const ssize_t got = pread(fd, out->data() + at, std::min<size_t>(end - at, (size_t)1 << 30), (off_t)at);
      if (got <= 0) {
        if (got < 0 && errno == EINTR) continue;
        shortread[k] = 1;
        return;
      }
      at += (size_t)got;
    }
  };
  std::vector<std::thread> th;
  for (unsigned k = 1; k < nt; ++k) th.emplace_back(slice, k);
  slice(0);
  for (auto& t : th) t.join();
  close(fd);
  for (unsigned k = 0; k < nt; ++k)
    if (shortread[k]) return fail(SIGAX_E_IO, "short read on %s", path);
  return SIGAX_OK;
}

// src/bwt.cpp:59-98: u16 magic 0xCACA, u64 nStrings, u64 nSymbols, u64 nRuns, i32 flag, then the RL units
static int parse_bwt(const std::vector<uint8_t>& buf, const char* path, u64* nstrings, u64* nsym, const uint8_t** runs,
                     u64* nruns) {
  if (buf.size() < 30) return fail(SIGAX_E_IO, "%s: truncated .bwt header", path);
  uint16_t magic;
  memcpy(&magic, buf.data(), 2);
  if (magic != 0xCACA) return fail(SIGAX_E_IO, "%s: bad .bwt magic", path);
  memcpy(nstrings, buf.data() + 2, 8);
  memcpy(nsym, buf.data() + 10, 8);
  memcpy(nruns, buf.data() + 18, 8);
  if (buf.size() < 30 + *nruns) return fail(SIGAX_E_IO, "%s: truncated .bwt payload", path);
  *runs = buf.data() + 30;
  return SIGAX_OK;
}

// src/suffix_array.cpp:57-95: "51914\n<strings>\n<elems>\n" then elems lines "<readIdx> <j>".  Tables of a million rows and
// more are parsed in chunks on the host's threads (BASELINE configs[2]: 2 x 20 M lines were 3 of the 3.9 s of `siga overlap`'s
// index load): chunks cut at line ends, lines counted, then every chunk parsed to its place.
static int parse_sai(const std::vector<uint8_t>& buf, const char* path, std::vector<uint32_t>* out) {
  const char* p = (const char*)buf.data();
  const char* e = p + buf.size();
  auto next = [&](const char*& q, const char* end, u64* v) -> bool {
    while (q < end && (*q < '0' || *q > '9')) ++q;
    if (q >= end) return false;
    u64 x = 0;
    while (q < end && *q >= '0' && *q <= '9') x = x * 10 + (u64)(*q++ - '0');
    *v = x;
    return true;
  };
  u64 magic = 0, strings = 0, elems = 0;
  if (!next(p, e, &magic) || magic != 0xCACA) return fail(SIGAX_E_IO, "%s: bad .sai magic", path);
  if (!next(p, e, &strings) || !next(p, e, &elems)) return fail(SIGAX_E_IO, "%s: truncated .sai header", path);
  if (elems > (u64)(e - p)) return fail(SIGAX_E_IO, "%s: truncated .sai body", path);  // every line takes bytes
  try {
    out->resize(elems);
  } catch (...) {
    return fail(SIGAX_E_IO, "%s: no memory for %llu rows", path, elems);
  }
  // one chunk, or as many as there are threads: [cut[k], cut[k+1]) starts right after a line end
  unsigned nt = elems >= (1u << 18) ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1u;
  if (p < e && *p == '\n') ++p;  // the header's own line end
  std::vector<const char*> cut(nt + 1, e);
  cut[0] = p;
  for (unsigned k = 1; k < nt; ++k) {
    const char* q = p + (u64)(e - p) * k / nt;
    q = (const char*)memchr(q, '\n', (size_t)(e - q));
    cut[k] = q ? q + 1 : e;
    if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
  }
  std::vector<u64> lines(nt + 1, 0);
  auto count = [&](unsigned k) {
    u64 c = 0;
    for (const char* q = cut[k]; q < cut[k + 1];) {  // a line = something up to '\n' (or the end) holding a digit
      const char* nl = (const char*)memchr(q, '\n', (size_t)(cut[k + 1] - q));
      const char* le = nl ? nl : cut[k + 1];
      bool digit = false;
      for (const char* t = q; t < le && !digit; ++t) digit = *t >= '0' && *t <= '9';
      c += digit ? 1 : 0;
      q = le + 1;
    }
    lines[k + 1] = c;
  };
  std::vector<int> bad(nt, 0);
  std::vector<u64> badrow(nt, 0), badid(nt, 0);
  auto parse = [&](unsigned k) {
    const char* q = cut[k];
    for (u64 i = lines[k]; i < lines[k + 1] && i < elems; ++i) {
      u64 a = 0, b2 = 0;
      if (!next(q, cut[k + 1], &a) || !next(q, cut[k + 1], &b2)) { bad[k] = 1; badrow[k] = i; return; }
      if (a >= strings) { bad[k] = 2; badrow[k] = i; badid[k] = a; return; }
      (*out)[i] = (uint32_t)a;
    }
    // one pair per line is what `siga index` writes; a chunk with numbers left over is some other layout: parse serially
    u64 extra = 0;
    if (nt > 1 && !bad[k] && next(q, cut[k + 1], &extra)) bad[k] = 3;
  };
  auto run = [&](auto fn) {
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nt; ++k) th.emplace_back(fn, k);
    fn(0u);
    for (auto& t : th) t.join();
  };
  for (;;) {
    if (nt > 1) {
      run(count);
      for (unsigned k = 0; k < nt; ++k) lines[k + 1] += lines[k];
    } else {
      lines[1] = elems;  // one chunk: the token stream as it comes, whatever the line layout
    }
    bool irregular = nt > 1 && lines[nt] != elems;
    if (!irregular) {
      run(parse);
      // numbers left over in a chunk, or a chunk that ran dry (pairs split across lines with the line count intact): some
      // other layout of a token stream that operator>> (src/suffix_array.cpp:57-95) may still accept -- the serial parse decides
      for (unsigned k = 0; k < nt; ++k) irregular = irregular || bad[k] == 3 || (nt > 1 && bad[k] == 1);
    }
    if (!irregular) break;
    nt = 1;  // once more, serially
    cut.assign(2, e);
    cut[0] = p;
    lines.assign(2, 0);
    bad.assign(1, 0);
    badrow.assign(1, 0);
    badid.assign(1, 0);
  }
  for (unsigned k = 0; k < nt; ++k) {
    if (bad[k] == 1) return fail(SIGAX_E_IO, "%s: truncated .sai body", path);
    if (bad[k] == 2) return fail(SIGAX_E_IO, "%s: read id %llu at row %llu, the table declares %llu strings", path, badid[k], badrow[k], strings);
  }
  return SIGAX_OK;
}

// Binary image of a parsed .sai beside the text file (<path>.bin: magic, size and mtime (ns) of the text, a checksum of its
// first and last 64 KiB, count, ids): parsing 5e7 decimal lines takes seconds, reading 200 MB does not.  A .sai is a
// permutation of 0..n-1, so every read set of n reads gives a text of the same size: the checksum is what tells a re-indexed
// prefix from the one the image was made of.  Stale or unreadable images are ignored and rewritten.
static const u64 SAI_IMAGE_MAGIC = 0x5349474153414932ull;  // "SIGASAI2"
static bool sai_text_stamp(const char* path, u64 stamp[3]) {
  struct stat st;
  if (stat(path, &st) != 0) return false;
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  u64 h = 1469598103934665603ull;  // FNV-1a over the head and the tail
  std::vector<unsigned char> buf(65536);
  auto eat = [&](size_t n) {
    for (size_t i = 0; i < n; ++i) h = (h ^ buf[i]) * 1099511628211ull;
  };
  eat(fread(buf.data(), 1, buf.size(), f));
  if ((u64)st.st_size > buf.size() && fseek(f, -(long)std::min<u64>(buf.size(), (u64)st.st_size - buf.size()), SEEK_END) == 0)
    eat(fread(buf.data(), 1, buf.size(), f));
  fclose(f);
  stamp[0] = (u64)st.st_size;
  stamp[1] = (u64)st.st_mtim.tv_sec * 1000000000ull + (u64)st.st_mtim.tv_nsec;
  stamp[2] = h;
  return true;
}
static bool sai_cache_load(const char* path, std::vector<uint32_t>* out) {
  u64 stamp[3];
  if (!sai_text_stamp(path, stamp)) return false;
  std::string cp = std::string(path) + ".bin";
  FILE* f = fopen(cp.c_str(), "rb");
  if (!f) return false;
  struct stat ist;
  u64 hdr[5];
  bool ok = fstat(fileno(f), &ist) == 0 && fread(hdr, 8, 5, f) == 5 && hdr[0] == SAI_IMAGE_MAGIC && hdr[1] == stamp[0] && hdr[2] == stamp[1] &&
            hdr[3] == stamp[2];
  // the count must be what the image file holds (a corrupt header must not size a vector) and a text of that size can hold
  // (every line is at least "0 0\n")
  ok = ok && hdr[4] <= 0xFFFFFFFFull && (u64)ist.st_size == 40 + 4 * hdr[4] && 4 * hdr[4] <= stamp[0];
  if (ok) {
    try {
      out->resize(hdr[4]);
      ok = hdr[4] == 0 || fread(out->data(), 4, hdr[4], f) == hdr[4];
    } catch (...) {
      ok = false;
    }
  }
  fclose(f);
  if (!ok) out->clear();
  return ok;
}
static void sai_cache_store(const char* path, const std::vector<uint32_t>& ids) {
  if (getenv("SIGAX_NO_SAI_CACHE")) return;
  u64 stamp[3];
  if (!sai_text_stamp(path, stamp)) return;
  // a temporary name of this process and thread: ranks of one job, or two runs on one prefix, write their own file and the
  // rename puts a complete one in place
  char uniq[64];
  snprintf(uniq, sizeof(uniq), ".tmp.%ld.%zx", (long)getpid(), std::hash<std::thread::id>()(std::this_thread::get_id()));
  std::string cp = std::string(path) + ".bin", tmp = cp + uniq;
  FILE* f = fopen(tmp.c_str(), "wbx");
  if (!f) return;  // read-only directory (or a leftover of this very name): no cache
  u64 hdr[5] = {SAI_IMAGE_MAGIC, stamp[0], stamp[1], stamp[2], (u64)ids.size()};
  bool ok = fwrite(hdr, 8, 5, f) == 5 && (ids.empty() || fwrite(ids.data(), 4, ids.size(), f) == ids.size());
  ok = fclose(f) == 0 && ok;
  if (ok) ok = rename(tmp.c_str(), cp.c_str()) == 0;
  if (!ok) remove(tmp.c_str());
}
static int load_sai(const char* path, std::vector<uint32_t>* out) {
  if (sai_cache_load(path, out)) return SIGAX_OK;
  std::vector<uint8_t> buf;
  int rc = read_file(path, &buf);
  if (rc == SIGAX_OK) rc = parse_sai(buf, path, out);
  if (rc == SIGAX_OK && out->size() >= (1u << 20)) sai_cache_store(path, *out);
  return rc;
}

static int upload(const void* src, size_t bytes, void** dst, u64* acct) {
  *dst = nullptr;
  size_t alloc = bytes ? bytes : 16;
  HIP_TRY(hipMalloc(dst, alloc));
  if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  if (acct) *acct += alloc;
  return SIGAX_OK;
}

extern "C" void sigax_index_close(sigax_index* ix) {
  if (!ix) return;
  hipSetDevice(ix->device);
  if (ix->tab_thread) {
    ix->tab_thread->join();
    delete ix->tab_thread;
  }
  delete ix->tab_state;
  if (ix->deep_thread) {
    ix->deep_thread->join();
    delete ix->deep_thread;
  }
  delete ix->deep_state;
  for (int s = 0; s < 2; ++s) {
    if (ix->d_deep[s]) hipFree(ix->d_deep[s]);
    if (ix->deep_new[s]) hipFree(ix->deep_new[s]);
    if (ix->d_slen[s]) hipFree(ix->d_slen[s]);
    if (ix->d_gran[s]) hipFree(ix->d_gran[s]);
    if (ix->d_gran2[s]) hipFree(ix->d_gran2[s]);
    if (ix->d_super2[s]) hipFree(ix->d_super2[s]);
    if (ix->d_sa[s]) hipFree(ix->d_sa[s]);
    if (ix->d_text[s]) hipFree(ix->d_text[s]);
    if (ix->d_xmap[s]) hipFree(ix->d_xmap[s]);
    if (ix->d_start[s]) hipFree(ix->d_start[s]);
    if (ix->d_super[s]) hipFree(ix->d_super[s]);
    if (ix->d_sai[s]) hipFree(ix->d_sai[s]);
  }
  if (ix->d_read_len) hipFree(ix->d_read_len);
  if (ix->d_name_rank) hipFree(ix->d_name_rank);
  if (ix->d_ptab) hipFree(ix->d_ptab);
  if (ix->ptab_ev) hipEventDestroy(ix->ptab_ev);
  if (ix->d_ktab) hipFree(ix->d_ktab);
  if (ix->d_csa) hipFree(ix->d_csa);
  if (ix->d_ctext) hipFree(ix->d_ctext);
  if (ix->d_cslen) hipFree(ix->d_cslen);
  if (ix->s_find) hipStreamDestroy(ix->s_find);
  if (ix->s_fx) hipStreamDestroy(ix->s_fx);
  if (ix->s_tail) hipStreamDestroy(ix->s_tail);
  if (ix->s_ord) hipStreamDestroy(ix->s_ord);
  delete ix->enqueue_mu;
  delete ix->cap_seen;
  delete ix;
}

// Row tables for the irreducible extractor (fm_layout.h: the suffix array as (stretch, offset), bit-packed, plus the
// stretches' text): a single-row block's extension rounds are read off its read's text instead of computed from rank
// lines, and a branch that leaves ONE single-row block in a group -- what a substitution in an overlapping read does -- is
// resolved by one lookup instead of a walk to the end of that read (~100 dependent rounds).  An accelerator like the
// two-step tables: skipped when memory is short or SIGAX_ROWEND=0 (SIGAX_LOOKAHEAD=0: no text, countdowns only), and the
// extractor then walks.  Built on the index's own device (a clone builds its own: 2 n LF steps on the spot beat copying
// the tables between GPUs).
static u32 bits_for(u64 maxval) {  // bits that hold 0 .. maxval
  u32 b = 1;
  while (b < 64 && (maxval >> b) != 0) ++b;
  return b;
}
struct RowTabGeom {
  u32 sa_bits, ld_bits, t_bits, text_stride;
  u64 sa_bytes, text_bytes;  // per strand
};
// syms = symbols an entry carries at most (as many as keep it within the 57 bits one unaligned 8-byte load delivers)
static RowTabGeom row_tab_geom(const sigax_index* ix, u32 maxlen, u32 syms) {
  RowTabGeom g;
  const u64 n_stretch = ix->st[0].C[1];
  g.ld_bits = bits_for(n_stretch ? n_stretch - 1 : 0);
  g.t_bits = bits_for(maxlen);
  g.sa_bits = g.ld_bits + g.t_bits;
  if (g.sa_bits < 57) g.sa_bits += 2 * std::min<u32>(syms, std::min<u32>(14u, (57 - g.sa_bits) / 2));
  g.text_stride = ((2 * maxlen + 7) / 8 + 8 + 7) & ~7u;  // 2 bits per symbol; the build ORs whole 8-byte words in
  g.sa_bytes = ((ix->n_symbols * g.sa_bits + 63) / 64) * 8 + 16;
  g.text_bytes = n_stretch * (u64)g.text_stride + 16;
  return g;
}
// the longest stretch this index can hold, as far as the host knows: no stretch is longer than the longest read
static u32 maxlen_bound(const sigax_index* ix) {
  if (ix->max_read_len) return ix->max_read_len;
  const u64 n_stretch = std::max<u64>(ix->st[0].C[1], 1);
  const u64 avg = ix->n_symbols / n_stretch;
  return (u32)std::min<u64>(std::max<u64>(2 * avg, avg + 64), (1u << 28) - 1);
}
// Which tables does this index get?  Decided from the free memory of that moment.
static void plan_row_tables(sigax_index* ix) {
  const char* envr = getenv("SIGAX_ROWEND");
  const char* envl = getenv("SIGAX_LOOKAHEAD");
  ix->tab_plan = 0;
  if ((envr && envr[0] == '0') || ix->st[0].C[1] >= 0xFFFFFFFFull || ix->n_symbols == 0) return;
  size_t mfree = 0, mtotal = 0;
  (void)hipMemGetInfo(&mfree, &mtotal);
  ix->tab_text = !(envl && envl[0] == '0');
  // Row table with as many of its entries' first symbols (14 at most) as fit half of the free memory (one lookup then
  // serves an item's first rounds: at BASELINE configs[1] 14 symbols, 56 bits per row); bare entries when they fit 70 %;
  // else -- when the .sai tables are there and every stretch is a read (no non-ACGT bases) -- the DIRECT MAPS (fm_layout.h):
  // text + 8 bytes per read and strand, no table per BWT symbol.  Measured (gpurun_out/r3u/): one lookup in the big table
  // beats two in small ones -- configs[1] 119.9 M reads/s on the row table, 108.8 M on direct maps (whose 112 MB compete
  // with the finder's table for the Infinity Cache: the finder goes from 8.3 to 9.1 ms), configs[2] shape 86.4 vs 77.4 M,
  // configs[4] (bare entries: two lookups either way) 38.4 vs 37.3 M with 188 vs 79 GB on the device -- so the direct maps
  // are what an index too big for a row table gets instead of nothing.  SIGAX_XMAP=1 forces them, =0 forbids them.
  const char* envx = getenv("SIGAX_XMAP");
  const bool can_direct = ix->tab_text && !(envx && envx[0] == '0') && ix->d_sai[0] && ix->d_sai[1] && ix->n_sai == ix->n_strings &&
                          ix->st[0].C[1] == ix->n_strings && ix->st[1].C[1] == ix->n_strings;
  auto plan_direct = [&]() -> bool {
    const RowTabGeom g = row_tab_geom(ix, maxlen_bound(ix), 0);
    const u64 want = 2 * (g.text_bytes + 8 * ix->n_strings);
    if (!can_direct || want >= mfree / 10 * 7) return false;
    ix->tab_direct = true;
    ix->tab_plan = want;
    ix->tab_syms = 0;
    return true;
  };
  ix->tab_direct = false;
  if (envx && envx[0] == '1' && plan_direct()) return;
  static const char* envk = getenv("SIGAX_ROW_SYMS");
  for (u32 syms = ix->tab_text ? (envk ? (u32)atoi(envk) : 14u) : 0u;; --syms) {
    const RowTabGeom g = row_tab_geom(ix, maxlen_bound(ix), syms);
    if (g.sa_bits > 57) break;
    const u64 want = 2 * (g.sa_bytes + (ix->tab_text ? g.text_bytes : 0));
    if (want < mfree / 10 * (syms ? 5 : 7)) {
      ix->tab_plan = want;
      ix->tab_syms = syms;
      return;
    }
    if (syms == 0) break;
  }
  (void)plan_direct();
}

static void free_row_tables(sigax_index* ix) {
  for (int s = 0; s < 2; ++s) {
    if (ix->d_sa[s]) hipFree(ix->d_sa[s]);
    if (ix->d_text[s]) hipFree(ix->d_text[s]);
    if (ix->d_xmap[s]) hipFree(ix->d_xmap[s]);
    ix->d_sa[s] = ix->d_text[s] = ix->d_xmap[s] = nullptr;
    ix->sa_alloc[s] = ix->text_alloc[s] = 0;
  }
}

// Allocation, on the caller's thread (by the bound of the longest stretch) ...
static bool alloc_row_tables(sigax_index* ix) {
  const RowTabGeom g = row_tab_geom(ix, maxlen_bound(ix), ix->tab_syms);
  ix->tab_plan = 0;
  hipError_t e = hipSuccess;
  for (int s = 0; s < 2 && e == hipSuccess; ++s) {
    if (!ix->tab_direct) {
      e = hipMalloc(&ix->d_sa[s], g.sa_bytes);
      if (e == hipSuccess) ix->sa_alloc[s] = g.sa_bytes;
    } else {
      e = hipMalloc(&ix->d_xmap[s], std::max<u64>(ix->n_strings, 1) * 8);
    }
    if (e == hipSuccess && ix->tab_text) {
      e = hipMalloc(&ix->d_text[s], g.text_bytes);
      if (e == hipSuccess) ix->text_alloc[s] = g.text_bytes;
    }
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    free_row_tables(ix);
    if (getenv("SIGAX_VERBOSE")) fprintf(stderr, "[sigax] row tables not allocated (%s): the extractor walks\n", hipGetErrorString(e));
    return false;
  }
  return true;
}

// ... and the fill, on a stream of its own (possibly on a side thread): the first walk measures the longest stretch, which
// fixes the entry width; buffers that turn out too small for it (the bound was an estimate) are allocated again here
static void fill_row_tables(sigax_index* ix, FmStrand out[2], u64* out_bytes) {
  out[0] = ix->st[0];
  out[1] = ix->st[1];
  *out_bytes = 0;
  const u64 n_stretch = ix->st[0].C[1];
  hipStream_t sb = nullptr;
  void* info = nullptr;
  u32* d_max = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&info, std::max<u64>(n_stretch, 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&d_max, 8);
  u32 maxlen[2] = {0, 0};
  RowTabGeom g[2];
  const bool direct = ix->tab_direct;
  // the stretches' lengths by '$' rank: what the direct maps are composed from, and what the deep start table's build reads
  // a row's remaining symbols off (kept with the index: 4 bytes per read and strand)
  u32* slen[2] = {nullptr, nullptr};
  u32* isai = nullptr;
  if (ix->tab_text) {
    for (int s = 0; s < 2 && e == hipSuccess; ++s) {
      if (!ix->d_slen[s]) e = hipMalloc((void**)&ix->d_slen[s], std::max<u64>(n_stretch, 1) * 4);
      slen[s] = ix->d_slen[s];
    }
  }
  if (direct && e == hipSuccess) e = hipMalloc((void**)&isai, std::max<u64>(n_stretch, 1) * 4);
  for (int s = 0; s < 2 && e == hipSuccess; ++s) {
    e = hipMemsetAsync(d_max, 0, 8, sb);
    if (e != hipSuccess) break;
    launch_stretch_scan(ix->st[s], ix->wide, n_stretch, (u64*)info, d_max, sb);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&maxlen[s], d_max, 4, hipMemcpyDeviceToHost, sb);
    if (e == hipSuccess) e = hipStreamSynchronize(sb);
    if (e != hipSuccess) break;
    g[s] = row_tab_geom(ix, maxlen[s], ix->tab_syms);
    if (g[s].sa_bits > 57) { e = hipErrorInvalidValue; break; }
    if (!direct && g[s].sa_bytes > ix->sa_alloc[s]) {
      hipFree(ix->d_sa[s]);
      ix->d_sa[s] = nullptr;
      ix->sa_alloc[s] = 0;
      e = hipMalloc(&ix->d_sa[s], g[s].sa_bytes);
      if (e != hipSuccess) break;
      ix->sa_alloc[s] = g[s].sa_bytes;
    }
    if (ix->tab_text && g[s].text_bytes > ix->text_alloc[s]) {
      hipFree(ix->d_text[s]);
      ix->d_text[s] = nullptr;
      ix->text_alloc[s] = 0;
      e = hipMalloc(&ix->d_text[s], g[s].text_bytes);
      if (e != hipSuccess) break;
      ix->text_alloc[s] = g[s].text_bytes;
    }
    if (!direct) e = hipMemsetAsync(ix->d_sa[s], 0, ix->sa_alloc[s], sb);
    if (e == hipSuccess && ix->tab_text) e = hipMemsetAsync(ix->d_text[s], 0, ix->text_alloc[s], sb);
    if (e != hipSuccess) break;
    launch_rows_fill(ix->st[s], ix->wide, n_stretch, (const u64*)info, direct ? nullptr : (unsigned char*)ix->d_sa[s], g[s].sa_bits, g[s].ld_bits,
                     g[s].t_bits, ix->tab_text ? (unsigned char*)ix->d_text[s] : nullptr, g[s].text_stride, slen[s], sb);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(sb);
  }
  // direct maps: strand s as extension index serves the blocks whose capped[0] counts the OTHER strand's '$' rows
  for (int s = 0; direct && s < 2 && e == hipSuccess; ++s) {
    launch_xmap(ix->d_sai[1 - s], ix->d_sai[s], isai, slen[s], ix->n_strings, (u64*)ix->d_xmap[s], sb);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(sb);
  }
  if (isai) hipFree(isai);
  if (sb) (void)hipStreamDestroy(sb);
  if (info) hipFree(info);
  if (d_max) hipFree(d_max);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (getenv("SIGAX_VERBOSE")) fprintf(stderr, "[sigax] row tables not built (%s): the extractor walks\n", hipGetErrorString(e));
    return;  // the buffers are freed with the index
  }
  for (int s = 0; s < 2; ++s) {
    out[s].sa = direct ? nullptr : (const unsigned char*)ix->d_sa[s];
    out[s].xmap = direct ? (const u64*)ix->d_xmap[s] : nullptr;
    if (direct) *out_bytes += 8 * ix->n_strings;
    if (ix->d_slen[s]) *out_bytes += 4 * n_stretch;
    out[s].text = ix->tab_text ? (const unsigned char*)ix->d_text[s] : nullptr;
    out[s].sa_bits = g[s].sa_bits;
    out[s].ld_bits = g[s].ld_bits;
    out[s].t_bits = g[s].t_bits;
    out[s].text_stride = g[s].text_stride;
    *out_bytes += ix->sa_alloc[s] + ix->text_alloc[s];
  }
  if (getenv("SIGAX_VERBOSE") && direct)
    fprintf(stderr, "[sigax] direct maps (8 bytes per read and strand) + text rows of %u bytes, %.2f GB\n", g[0].text_stride, *out_bytes / 1e9);
  else if (getenv("SIGAX_VERBOSE"))
    fprintf(stderr, "[sigax] row tables: %u bits per row (stretch %u + offset %u + %u symbols), text rows of %u bytes, %.2f GB\n", g[0].sa_bits,
            g[0].ld_bits, g[0].t_bits, (g[0].sa_bits - g[0].ld_bits - g[0].t_bits) / 2, ix->tab_text ? g[0].text_stride : 0u, *out_bytes / 1e9);
}

// the tables of a finished build become visible to the runs enqueued from now on
static void publish_tables(sigax_index* ix) {
  if (!ix->tab_state || ix->tab_state->load(std::memory_order_acquire) != 2) return;
  for (int s = 0; s < 2; ++s) {
    ix->st[s].sa = ix->tab_st[s].sa;
    ix->st[s].xmap = ix->tab_st[s].xmap;
    ix->st[s].text = ix->tab_st[s].text;
    ix->st[s].sa_bits = ix->tab_st[s].sa_bits;
    ix->st[s].ld_bits = ix->tab_st[s].ld_bits;
    ix->st[s].t_bits = ix->tab_st[s].t_bits;
    ix->st[s].text_stride = ix->tab_st[s].text_stride;
  }
  ix->device_bytes += ix->tab_bytes;
  ix->tab_state->store(0, std::memory_order_release);
}

// start (or do) the build: allocate here, fill on a side thread unless `sync`
static void start_row_tables(sigax_index* ix, bool sync) {
  if (ix->tab_plan == 0 || ix->tab_tried) return;
  ix->tab_tried = true;
  if (!alloc_row_tables(ix)) return;
  if (sync) {
    fill_row_tables(ix, ix->tab_st, &ix->tab_bytes);
    ix->tab_state->store(2);
    publish_tables(ix);
    return;
  }
  ix->tab_state->store(1);
  ix->tab_thread = new std::thread([ix] {
    (void)hipSetDevice(ix->device);
    fill_row_tables(ix, ix->tab_st, &ix->tab_bytes);
    ix->tab_state->store(2, std::memory_order_release);
  });
}

// When are they built?  The build walks the whole index (C3: 1.8 s, 45 GB) and saves ~20 ns per read afterwards: it pays
// on an index that stays open -- a service, bench.py -- and does not in one pass of `siga overlap` over the reads the
// index was made of (BASELINE configs[2]'s read set through the CLI: 7.2 s with the tables, 3.6 s without).  So: small
// indexes (and SIGAX_TABLES_SYNC=1) at once; the others in the background once the index has been asked for as many
// reads as it holds (enqueue()), or at once when the caller says the index is here to stay (sigax_index_prepare).
static void build_rowend(sigax_index* ix) {
  ix->tab_state = new std::atomic<int>(0);
  ix->deep_state = new std::atomic<int>(0);
  plan_row_tables(ix);
  if (ix->n_symbols < (1ull << 26) || getenv("SIGAX_TABLES_SYNC") != nullptr) start_row_tables(ix, true);
}
// the tables in place before this returns (caller holds enqueue_mu)
static void row_tables_now(sigax_index* ix) {
  if (ix->tab_thread) {  // a build in flight: wait for it
    ix->tab_thread->join();
    delete ix->tab_thread;
    ix->tab_thread = nullptr;
  }
  publish_tables(ix);
  start_row_tables(ix, true);  // (no-op when they were built, or tried, before)
}


// ------------------------------------------------------------------------------------------------------
// Deep start tables of the block finder (fm_layout.h, sigax_index_prepare_overlap).  Needs the row tables and the
// stretch text (the distinct K-mers are read off them); an accelerator like those: when memory is short, the index has no
// row tables, or a K-mer's walk does not come out at its own rows, there is no table and every chain walks.
// SIGAX_FIND_DEEP=0 never builds them; SIGAX_DEEP_K=k overrides K (tests); SIGAX_DEEP_LOAD=percent sets the load factor.
// ------------------------------------------------------------------------------------------------------
static uint32_t deep_k_for(uint32_t min_overlap) {
  static const char* env = getenv("SIGAX_FIND_DEEP");
  if (env && env[0] == '0') return 0;
  static const char* envk = getenv("SIGAX_DEEP_K");
  uint32_t k = envk ? (uint32_t)atoi(envk) : std::min<uint32_t>(min_overlap, SIGAX_DEEP_KMAX);
  if (k > min_overlap || k > SIGAX_DEEP_KMAX) k = std::min<uint32_t>(min_overlap, SIGAX_DEEP_KMAX);
  if (k < (envk ? 2u : (uint32_t)SIGAX_DEEP_KMIN)) return 0;
  return k;
}
// Tables for K from the strands' row tables `st` (a snapshot taken under enqueue_mu).  `share` = the part of the free
// memory they may take, in per cent.  On success tab[] / slots[] / *bytes are set; on any failure nothing is left allocated.
static bool build_deep_tables(sigax_index* ix, const FmStrand st[2], uint32_t K, unsigned share, void* tab[2], u64 slots[2], u64* bytes) {
  tab[0] = tab[1] = nullptr;
  slots[0] = slots[1] = 0;
  *bytes = 0;
  if (K == 0 || !st[0].sa || !st[1].sa || !st[0].text || !st[1].text || !ix->d_slen[0] || !ix->d_slen[1]) return false;
  const bool verbose = getenv("SIGAX_VERBOSE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  const u64 n_stretch = ix->st[0].C[1];
  static const char* envl = getenv("SIGAX_DEEP_LOAD");
  hipStream_t sb = nullptr;
  u64* d_cnt = nullptr;  // [0] distinct K-mers, [1] errors
  u64* list = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&d_cnt, 16);
  bool ok = e == hipSuccess;
  u64 distinct[2] = {0, 0};
  for (int s = 0; s < 2 && ok; ++s) {
    u64 h[2] = {0, 0};
    ok = hipMemsetAsync(d_cnt, 0, 16, sb) == hipSuccess;
    if (!ok) break;
    launch_deep_scan(st[s], ix->d_slen[s], n_stretch, K, d_cnt, nullptr, 0, sb);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, sb) == hipSuccess &&
         hipStreamSynchronize(sb) == hipSuccess;
    distinct[s] = h[0];
    if (distinct[s] >= (1ull << 32) - 256) ok = false;  // (one launch of k_deep_fill, one lane per K-mer)
  }
  if (ok) {
    // both strands' tables + the larger list must fit `share` per cent of what is free now
    size_t mfree = 0, mtotal = 0;
    (void)hipMemGetInfo(&mfree, &mtotal);
    unsigned load = envl ? (unsigned)std::min(95, std::max(5, atoi(envl))) : 50u;
    for (;;) {
      for (int s = 0; s < 2; ++s) slots[s] = std::max<u64>(64, distinct[s] * 100 / load + 16);
      const u64 need = (slots[0] + slots[1]) * deep_entry_bytes() + std::max(distinct[0], distinct[1]) * 8;
      if (need <= (u64)mfree / 100 * share) break;
      if (envl || load >= 80) { ok = false; break; }
      load += 15;  // 50, 65, 80 per cent: longer probe sequences before no table at all
    }
    if (!ok && verbose) fprintf(stderr, "[sigax] deep start tables (K = %u, %llu + %llu K-mers) do not fit %u %% of the free memory\n", K,
                                distinct[0], distinct[1], share);
  }
  for (int s = 0; s < 2 && ok; ++s) {
    u64 h[2] = {0, 0};
    ok = hipMalloc((void**)&list, std::max<u64>(distinct[s], 1) * 8) == hipSuccess && hipMalloc(&tab[s], slots[s] * deep_entry_bytes()) == hipSuccess &&
         hipMemsetAsync(tab[s], 0, slots[s] * deep_entry_bytes(), sb) == hipSuccess && hipMemsetAsync(d_cnt, 0, 16, sb) == hipSuccess;
    if (!ok) break;
    launch_deep_scan(st[s], ix->d_slen[s], n_stretch, K, d_cnt, list, distinct[s], sb);
    launch_deep_fill(st[s], st[1 - s], ix->wide, ix->d_slen[s], n_stretch, K, list, distinct[s], tab[s], slots[s], d_cnt + 1, sb);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, sb) == hipSuccess &&
         hipStreamSynchronize(sb) == hipSuccess;
    if (ok && (h[0] != distinct[s] || h[1] != 0)) {
      if (verbose) fprintf(stderr, "[sigax] deep start table of strand %d: %llu K-mers listed of %llu, %llu walks astray: no table\n", s, h[0], distinct[s], h[1]);
      ok = false;
    }
    hipFree(list);
    list = nullptr;
    *bytes += slots[s] * deep_entry_bytes();
  }
  if (list) hipFree(list);
  if (d_cnt) hipFree(d_cnt);
  if (sb) (void)hipStreamDestroy(sb);
  if (!ok) {
    (void)hipGetLastError();
    for (int s = 0; s < 2; ++s) {
      if (tab[s]) hipFree(tab[s]);
      tab[s] = nullptr;
      slots[s] = 0;
    }
    *bytes = 0;
    return false;
  }
  if (verbose)
    fprintf(stderr, "[sigax] deep start tables: K = %u, %llu + %llu distinct K-mers, %.2f GB, %.3f s\n", K, distinct[0], distinct[1], *bytes / 1e9,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return true;
}
// (caller holds enqueue_mu) a finished background build becomes visible to the runs enqueued from now on
static void publish_deep(sigax_index* ix) {
  if (!ix->deep_state || ix->deep_state->load(std::memory_order_acquire) != 2) return;
  if (ix->deep_thread) {
    ix->deep_thread->join();
    delete ix->deep_thread;
    ix->deep_thread = nullptr;
  }
  if (ix->deep_new[0] && ix->deep_new[1]) {
    for (int s = 0; s < 2; ++s) {
      ix->d_deep[s] = ix->deep_new[s];
      ix->deep_slots[s] = ix->deep_new_slots[s];
      ix->deep_new[s] = nullptr;
      ix->st[s].deep = ix->d_deep[s];
      ix->st[s].deep_slots = ix->deep_slots[s];
      ix->st[s].deep_k = ix->deep_new_k;
    }
    ix->deep_k = ix->deep_new_k;
    ix->deep_bytes = ix->deep_new_bytes;
    ix->device_bytes += ix->deep_bytes;
  }
  ix->deep_state->store(0, std::memory_order_release);
}
// (caller holds enqueue_mu) the index is being reused and this run's min-overlap has no table: build one beside the runs
static void start_deep_tables(sigax_index* ix, uint32_t min_overlap) {
  if (ix->deep_tried || !ix->deep_state || ix->deep_state->load() != 0) return;
  if (ix->deep_k != 0 && ix->deep_k <= min_overlap) return;
  const uint32_t K = deep_k_for(min_overlap);
  if (K == 0 || !ix->st[0].sa || !ix->st[0].text || !ix->st[1].sa || !ix->st[1].text) return;
  ix->deep_tried = true;  // one background attempt per index; sigax_index_prepare_overlap may still replace the table
  if (ix->deep_k != 0) return;  // a table for a larger K is in use by runs in flight: only prepare_overlap swaps tables
  ix->deep_state->store(1);
  FmStrand snap[2] = {ix->st[0], ix->st[1]};
  ix->deep_new_k = K;
  ix->deep_thread = new std::thread([ix, snap, K] {
    (void)hipSetDevice(ix->device);
    (void)build_deep_tables(ix, snap, K, 25, ix->deep_new, ix->deep_new_slots, &ix->deep_new_bytes);
    ix->deep_state->store(2, std::memory_order_release);
  });
}

// The index's own streams.  The finder is the critical path of a step: its stream gets the higher priority.
// SIGAX_CU_SPLIT=K (an experiment, off by default): the finder's stream is confined to all but K of the CUs and the
// filter/extract and tail streams to those K (CU mask bits interleave over XCDs and shader engines, so a run of mask
// bits is an even share of every XCD) -- no priorities then, hipExtStreamCreateWithCUMask takes none.
static hipError_t pipeline_streams(sigax_index* ix) {
  const char* env = getenv("SIGAX_CU_SPLIT");
  const int k = env ? atoi(env) : 0;
  if (k > 0 && k < ix->n_cu) {
    const int words = (ix->n_cu + 31) / 32;
    std::vector<uint32_t> lo((size_t)words, 0u), hi((size_t)words, 0u);
    for (int c = 0; c < ix->n_cu; ++c) (c < ix->n_cu - k ? lo : hi)[(size_t)c / 32] |= 1u << (c % 32);
    hipError_t e = hipExtStreamCreateWithCUMask(&ix->s_find, (uint32_t)words, lo.data());
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ix->s_fx, (uint32_t)words, hi.data());
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ix->s_tail, (uint32_t)words, hi.data());
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ix->s_ord, (uint32_t)words, hi.data());
    return e;
  }
  int prio_least = 0, prio_greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&ix->s_find, hipStreamNonBlocking, prio_greatest);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&ix->s_fx, hipStreamNonBlocking, prio_least);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&ix->s_tail, hipStreamNonBlocking, prio_greatest);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&ix->s_ord, hipStreamNonBlocking, prio_greatest);
  return e;
}

// SIGAX_VERBOSE: where the time of opening an index goes
struct OpenClock {
  bool on;
  std::chrono::steady_clock::time_point t;
  OpenClock() : on(getenv("SIGAX_VERBOSE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    const auto n = std::chrono::steady_clock::now();
    if (on) fprintf(stderr, "[sigax] open: %-34s %7.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

extern "C" int sigax_index_open_mem(const uint8_t* runs, uint64_t n_runs, const uint8_t* rruns, uint64_t n_rruns,
                                    uint64_t n_symbols, uint64_t n_strings, const uint32_t* sai, const uint32_t* rsai,
                                    int device, sigax_index** out) {
  if (!out || (!runs && n_runs) || (!rruns && n_rruns)) return fail(SIGAX_E_ARG, "NULL argument");
  *out = nullptr;
  // Forward strand only (rruns == NULL, n_rruns == 0): the index `siga index --no-reverse` writes and `siga correct` reads
  // (src/correct.cpp:41-47 loads <prefix>.bwt alone; examples/siga-ecoli-miseq.sh:64-70).  Serves Occ, k-mer counts and the
  // corrector; overlap runs need both strands and fail with SIGAX_E_STATE.
  const bool fwd_only = rruns == nullptr && n_rruns == 0 && n_symbols > 0;
  const int nst = fwd_only ? 1 : 2;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SIGAX_E_DEVICE, "no HIP device visible: the overlap path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(SIGAX_E_ARG, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  OpenClock clk;
  sigax_index* ix = new sigax_index();
  memset(ix, 0, sizeof(*ix));
  ix->device = device;
  ix->enqueue_mu = new std::mutex();
  ix->cap_seen = new std::atomic<uint32_t>(0);
  if (hipDeviceGetAttribute(&ix->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ix->n_cu <= 0) ix->n_cu = 256;
  {
    const hipError_t e = pipeline_streams(ix);
    if (e != hipSuccess) {
      sigax_index_close(ix);
      return fail(SIGAX_E_DEVICE, "creating the pipeline streams: %s", hipGetErrorString(e));
    }
  }
  ix->n_symbols = n_symbols;
  ix->n_strings = n_strings;
  ix->fwd_only = fwd_only;
  // 64-bit positions when the BWT does not fit 32 bits (SIGAX_FORCE_WIDE=1 exercises that path on small inputs)
  ix->wide = n_symbols >= 0xFFFFFFF0ull || getenv("SIGAX_FORCE_WIDE") != nullptr;
  const uint8_t* rr[2] = {runs, rruns};
  u64 nr[2] = {n_runs, n_rruns};
  {
    // The second strand is decoded in the first one's scratch memory.  The session spans the two decodes only: its parked
    // scratch blocks are invisible to hipMemGetInfo, and the optional tables below are planned from the free memory.
    struct DecodeSession {
      DecodeSession() { sigax_build_session(1); }
      ~DecodeSession() { sigax_build_session(0); }
    } decode_session;
    for (int s = 0; s < nst; ++s) {
      u64 C[5], total[5], gb = 0, sb = 0;
      int rc = sigax_decode_strand(rr[s], nr[s], n_symbols, ix->wide, &ix->d_gran[s], &gb, &ix->d_super[s], &sb, C, total);
      if (rc != SIGAX_OK) {
        sigax_index_close(ix);
        return rc;
      }
      ix->device_bytes += gb + sb;
      ix->st[s].granules = (const uint32_t*)ix->d_gran[s];
      ix->st[s].super = (const u64*)ix->d_super[s];
      ix->st[s].n = n_symbols;
      for (int k = 0; k < 5; ++k) {
        ix->st[s].C[k] = C[k];
        ix->st[s].total[k] = total[k];
      }
    }
  }
  clk.lap("streams, upload + decode");
  for (int k = 0; k < 5 && !fwd_only; ++k) {
    if (ix->st[0].total[k] != ix->st[1].total[k]) {
      sigax_index_close(ix);
      return fail(SIGAX_E_IO, "forward and reverse BWT hold different symbol counts: not a .bwt/.rbwt pair");
    }
  }
  // Two-step tables for the block finder (2 bytes per symbol and strand), built on the device from the granules just
  // uploaded.  The finder then runs one launch per strand (chains 0,1 / 2,3): gathering from one table at a time keeps
  // the randomly accessed footprint small -- measured on MI355X per 1 M reads, index of 0.15 / 0.6 / 1.2 G symbols:
  // one-step finder 10.6 / 11.5 / 13.9 ms, two-step with both tables in one launch 7.3 / 9.1 / 17.4 ms (one lane per
  // 128-byte granule runs into address translation once more than ~4 GB are gathered from: tools/gather_probe3.hip),
  // two-step with one launch per strand 6.8 / 8.5 / 9.2 ms.  Up to 1.6 G symbols (u32 byte offsets into the table).
  // SIGAX_TWO_STEP=0 turns the tables off, SIGAX_TWO_STEP_MAX_SYMBOLS moves the limit (never beyond 2^31).
  {
    const char* env2 = getenv("SIGAX_TWO_STEP");
    const char* envm = getenv("SIGAX_TWO_STEP_MAX_SYMBOLS");
    // every index with 32-bit positions: below SIGAX_COOP_MIN_SYMBOLS (2^31) the finder gathers per lane with u32 byte
    // offsets (k_find_n2, tables under 4 GiB), above it lines come cooperatively through LDS with 64-bit addresses (k_find_c2)
    // 64-bit-position indexes too: the lines' counters are then relative to 2^32-row superblocks (fm_layout.h)
    const u64 max2 = envm ? strtoull(envm, nullptr, 10) : ~0ull;
    const bool want2 = n_symbols < max2 && !(env2 && env2[0] == '0');
    if (want2) {
      const u64 ng2 = n_symbols / SIGAX_GRAN2_SYMS + 1;
      void *cnt = nullptr, *offs = nullptr, *partial = nullptr, *total = nullptr;
      hipError_t e = hipMalloc(&cnt, 20 * ng2 * 4);
      if (e == hipSuccess) e = hipMalloc(&offs, (ng2 + 2) * 8);
      if (e == hipSuccess) e = hipMalloc(&partial, scan_partials_needed(ng2) * 8);
      if (e == hipSuccess) e = hipMalloc(&total, 8);
      const u64 nsup2 = ((ng2 - 1) >> (SIGAX_SUPER_SHIFT - 6)) + 1;
      for (int s = 0; s < nst && e == hipSuccess; ++s) {
        e = hipMalloc(&ix->d_gran2[s], ng2 * SIGAX_GRAN2_WORDS * 4);
        if (e != hipSuccess) break;
        ix->device_bytes += ng2 * SIGAX_GRAN2_WORDS * 4;
        if (ix->wide) {
          e = hipMalloc(&ix->d_super2[s], nsup2 * 20 * 8);
          if (e != hipSuccess) break;
        }
        launch_build2(ix->st[s], ix->wide, (uint32_t*)ix->d_gran2[s], (u64*)ix->d_super2[s], (uint32_t*)cnt, (u64*)offs, (u64*)partial,
                      (u64*)total, nullptr);
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipGetLastError();
        ix->st[s].gran2 = (const uint32_t*)ix->d_gran2[s];
        ix->st[s].super2 = (const u64*)ix->d_super2[s];
      }
      ix->split_strands = true;
      if (cnt) hipFree(cnt);
      if (offs) hipFree(offs);
      if (partial) hipFree(partial);
      if (total) hipFree(total);
      if (e != hipSuccess) {
        // the tables are an accelerator, not a requirement: without them the one-step finder runs
        (void)hipGetLastError();
        for (int s = 0; s < 2; ++s) {
          if (ix->d_gran2[s]) {
            hipFree(ix->d_gran2[s]);
            ix->device_bytes -= ng2 * SIGAX_GRAN2_WORDS * 4;
          }
          if (ix->d_super2[s]) hipFree(ix->d_super2[s]);
          ix->d_gran2[s] = ix->d_super2[s] = nullptr;
          ix->st[s].gran2 = nullptr;
          ix->st[s].super2 = nullptr;
        }
        ix->split_strands = false;
        if (getenv("SIGAX_VERBOSE")) fprintf(stderr, "[sigax] two-step tables not built (%s): one-step finder\n", hipGetErrorString(e));
      }
    }
  }
  clk.lap("two-step tables");
  // Start tables of the finder (fm_layout.h): from 2^22 symbols on (the 2 x 268 MB and 20 ms are out of proportion for
  // less; SIGAX_FIND_START=1 forces them, =0 turns them off), an accelerator like the others.
  {
    const char* envs = getenv("SIGAX_FIND_START");
    const bool want = envs ? envs[0] != '0' : n_symbols >= (1ull << 22);
    if (want && n_symbols > 0 && !fwd_only) {
      hipError_t e = hipSuccess;
      for (int s = 0; s < 2 && e == hipSuccess; ++s) {
        e = hipMalloc(&ix->d_start[s], start_table_bytes(ix->wide));
        if (e != hipSuccess) break;
        launch_start_build(ix->st[s], ix->st[1 - s], ix->wide, ix->d_start[s], nullptr);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipDeviceSynchronize();
      if (e != hipSuccess) {
        (void)hipGetLastError();
        for (int s = 0; s < 2; ++s) {
          if (ix->d_start[s]) hipFree(ix->d_start[s]);
          ix->d_start[s] = nullptr;
        }
      } else {
        for (int s = 0; s < 2; ++s) ix->st[s].start = ix->d_start[s];
        ix->device_bytes += 2 * start_table_bytes(ix->wide);
      }
    }
  }
  clk.lap("start tables");
  if (sai && rsai && !fwd_only) {
    const uint32_t* ss[2] = {sai, rsai};
    for (int s = 0; s < 2; ++s)  // k_edges indexes the read tables with these ids
      for (u64 i = 0; i < n_strings; ++i)
        if (ss[s][i] >= n_strings) {
          sigax_index_close(ix);
          return fail(SIGAX_E_IO, "%s table: read id %u at row %llu, the index holds %llu strings", s ? ".rsai" : ".sai", ss[s][i], i,
                      (u64)n_strings);
        }
    for (int s = 0; s < 2; ++s) {
      int rc = upload(ss[s], n_strings * 4, (void**)&ix->d_sai[s], &ix->device_bytes);
      if (rc != SIGAX_OK) {
        sigax_index_close(ix);
        return rc;
      }
    }
    ix->n_sai = n_strings;
  }
  clk.lap(".sai check + upload");
  if (fwd_only) {
    ix->tab_state = new std::atomic<int>(0);  // no extractor, no row tables
    ix->deep_state = new std::atomic<int>(0);
  } else {
    build_rowend(ix);  // after the .sai tables: with them the extractor's tables are direct maps (fm_layout.h)
  }
  clk.lap("row tables (plan, start of build)");
  *out = ix;
  return SIGAX_OK;
}

extern "C" int sigax_index_open(const char* bwt_path, const char* rbwt_path, const char* sai_path, const char* rsai_path,
                                int device, sigax_index** out) {
  if (!bwt_path || !out) return fail(SIGAX_E_ARG, "NULL argument");
  if (!rbwt_path || !rbwt_path[0]) {  // forward strand only (sigax_index_open_mem says what that serves)
    std::vector<uint8_t> fb;
    int rc = read_file(bwt_path, &fb);
    if (rc != SIGAX_OK) return rc;
    u64 ns = 0, nsym = 0, nruns = 0;
    const uint8_t* runs = nullptr;
    if ((rc = parse_bwt(fb, bwt_path, &ns, &nsym, &runs, &nruns)) != SIGAX_OK) return rc;
    return sigax_index_open_mem(runs, nruns, nullptr, 0, nsym, ns, nullptr, nullptr, device, out);
  }
  std::vector<uint8_t> fb, rb;
  std::vector<uint32_t> sai, rsai;
  const bool have_sai = sai_path && rsai_path && sai_path[0] && rsai_path[0];
  // the four files side by side (.sai text: tens of millions of lines at BASELINE configs[2] and [4]); the error text is
  // thread-local, so every side thread hands its own over
  int rcs[4] = {SIGAX_OK, SIGAX_OK, SIGAX_OK, SIGAX_OK};
  std::string errs[4];
  auto side = [&](int k, auto fn) {
    return std::thread([&rcs, &errs, k, fn] {
      rcs[k] = fn();
      if (rcs[k] != SIGAX_OK) errs[k] = g_err;
    });
  };
  std::vector<std::thread> sides;
  // the HIP runtime comes up (0.2-0.3 s in a fresh process) while the files are read, not after them
  sides.push_back(std::thread([device] {
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr);
    (void)hipGetLastError();
  }));
  sides.push_back(side(1, [&] { return read_file(rbwt_path, &rb); }));
  if (have_sai) {
    sides.push_back(side(2, [&] { return load_sai(sai_path, &sai); }));
    sides.push_back(side(3, [&] { return load_sai(rsai_path, &rsai); }));
  }
  OpenClock clk;
  rcs[0] = read_file(bwt_path, &fb);
  if (rcs[0] != SIGAX_OK) errs[0] = g_err;
  clk.lap(".bwt read");
  for (auto& t : sides) t.join();
  clk.lap(".rbwt read, .sai tables parsed");
  for (int k = 0; k < 2; ++k)
    if (rcs[k] != SIGAX_OK) return fail(rcs[k], "%s", errs[k].c_str());
  int rc;
  u64 ns[2], nsym[2], nruns[2];
  const uint8_t* runs[2];
  if ((rc = parse_bwt(fb, bwt_path, &ns[0], &nsym[0], &runs[0], &nruns[0])) != SIGAX_OK) return rc;
  if ((rc = parse_bwt(rb, rbwt_path, &ns[1], &nsym[1], &runs[1], &nruns[1])) != SIGAX_OK) return rc;
  if (ns[0] != ns[1] || nsym[0] != nsym[1]) return fail(SIGAX_E_IO, "%s and %s describe different read sets", bwt_path, rbwt_path);
  for (int k = 2; k < 4; ++k)  // what is wrong with the .bwt files is said first, as when the files were read one by one
    if (rcs[k] != SIGAX_OK) return fail(rcs[k], "%s", errs[k].c_str());
  if (have_sai && (sai.size() != ns[0] || rsai.size() != ns[0]))
    return fail(SIGAX_E_IO, ".sai tables (%zu, %zu entries) do not match the %llu strings of the .bwt", sai.size(), rsai.size(), ns[0]);
  return sigax_index_open_mem(runs[0], nruns[0], runs[1], nruns[1], nsym[0], ns[0], have_sai ? sai.data() : nullptr,
                              have_sai ? rsai.data() : nullptr, device, out);
}

// Replica of an open index on another GPU of the node, copied device to device (xGMI between MI355X peers) instead of
// being decoded and uploaded again: SURVEY.md 8(e) "index broadcast at start-up".
extern "C" int sigax_index_prepare(sigax_index* ix) {
  if (!ix) return fail(SIGAX_E_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ix->device));
  std::lock_guard<std::mutex> lock(*ix->enqueue_mu);
  row_tables_now(ix);
  return SIGAX_OK;
}

extern "C" int sigax_index_prepare_overlap(sigax_index* ix, uint32_t min_overlap) {
  if (!ix) return fail(SIGAX_E_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ix->device));
  std::lock_guard<std::mutex> lock(*ix->enqueue_mu);
  row_tables_now(ix);
  // a background build in flight: let it finish, then see whether its table serves
  if (ix->deep_thread) {
    ix->deep_thread->join();
    delete ix->deep_thread;
    ix->deep_thread = nullptr;
  }
  publish_deep(ix);
  const uint32_t K = deep_k_for(min_overlap);
  if (K == 0 || (ix->deep_k != 0 && ix->deep_k <= min_overlap)) return SIGAX_OK;
  if (ix->d_deep[0]) {
    // a table for a larger K: runs in flight may still read it
    HIP_TRY(hipDeviceSynchronize());
    for (int s = 0; s < 2; ++s) {
      hipFree(ix->d_deep[s]);
      ix->d_deep[s] = nullptr;
      ix->deep_slots[s] = 0;
      ix->st[s].deep = nullptr;
      ix->st[s].deep_slots = 0;
      ix->st[s].deep_k = 0;
    }
    ix->device_bytes -= ix->deep_bytes;
    ix->deep_bytes = 0;
    ix->deep_k = 0;
  }
  FmStrand snap[2] = {ix->st[0], ix->st[1]};
  ix->deep_new_k = K;
  (void)build_deep_tables(ix, snap, K, 45, ix->deep_new, ix->deep_new_slots, &ix->deep_new_bytes);
  ix->deep_state->store(2);
  publish_deep(ix);
  return SIGAX_OK;
}

extern "C" int sigax_index_clone(const sigax_index* src, int device, sigax_index** out) {
  if (!src || !out) return fail(SIGAX_E_ARG, "NULL argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SIGAX_E_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(SIGAX_E_ARG, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  if (device != src->device) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(src->device, 0);
    (void)hipGetLastError();  // already enabled is fine; hipMemcpyPeer works either way (staged when there is no direct path)
  }
  sigax_index* ix = new sigax_index();
  memset(ix, 0, sizeof(*ix));
  ix->device = device;
  ix->enqueue_mu = new std::mutex();
  ix->cap_seen = new std::atomic<uint32_t>(src->cap_seen->load());
  ix->n_cu = src->n_cu;
  (void)hipDeviceGetAttribute(&ix->n_cu, hipDeviceAttributeMultiprocessorCount, device);
  {
    const hipError_t e = pipeline_streams(ix);
    if (e != hipSuccess) {
      sigax_index_close(ix);
      return fail(SIGAX_E_DEVICE, "creating the pipeline streams: %s", hipGetErrorString(e));
    }
  }
  ix->wide = src->wide;
  ix->n_symbols = src->n_symbols;
  ix->n_strings = src->n_strings;
  ix->n_sai = src->n_sai;
  ix->n_meta = src->n_meta;
  ix->max_read_len = src->max_read_len;
  ix->split_strands = src->split_strands;
  ix->fwd_only = src->fwd_only;
  const u64 ngran = src->n_symbols / SIGAX_GRANULE_SYMS + 1;
  const u64 nsuper = ((ngran - 1) >> (SIGAX_SUPER_SHIFT - 7)) + 1;
  const u64 ng2 = src->n_symbols / SIGAX_GRAN2_SYMS + 1;
  auto copy = [&](void** dst, const void* from, size_t bytes) -> int {
    *dst = nullptr;
    if (!from) return SIGAX_OK;
    HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
    if (bytes) HIP_TRY(hipMemcpyPeer(*dst, device, from, src->device, bytes));
    ix->device_bytes += bytes;
    return SIGAX_OK;
  };
  int rc = SIGAX_OK;
  for (int s = 0; s < 2 && rc == SIGAX_OK; ++s) {
    rc = copy(&ix->d_gran[s], src->d_gran[s], ngran * 64);
    if (rc == SIGAX_OK) rc = copy(&ix->d_super[s], src->d_super[s], nsuper * 32);
    if (rc == SIGAX_OK) rc = copy(&ix->d_gran2[s], src->d_gran2[s], ng2 * SIGAX_GRAN2_WORDS * 4);
    if (rc == SIGAX_OK) rc = copy(&ix->d_super2[s], src->d_super2[s], (((ng2 - 1) >> (SIGAX_SUPER_SHIFT - 6)) + 1) * 20 * 8);
    if (rc == SIGAX_OK) rc = copy((void**)&ix->d_sai[s], src->d_sai[s], src->n_sai * 4);
    if (rc == SIGAX_OK) rc = copy(&ix->d_start[s], src->d_start[s], start_table_bytes(src->wide));
    ix->st[s] = src->st[s];
    ix->st[s].sa = nullptr;
    ix->st[s].xmap = nullptr;
    ix->st[s].text = nullptr;
    ix->st[s].deep = nullptr;  // the replica builds its own (sigax_index_prepare_overlap, or once it is reused)
    ix->st[s].deep_slots = 0;
    ix->st[s].deep_k = 0;
    ix->st[s].granules = (const uint32_t*)ix->d_gran[s];
    ix->st[s].super = (const u64*)ix->d_super[s];
    ix->st[s].gran2 = (const uint32_t*)ix->d_gran2[s];
    ix->st[s].super2 = (const u64*)ix->d_super2[s];
    ix->st[s].start = ix->d_start[s];
  }
  if (rc == SIGAX_OK) rc = copy((void**)&ix->d_read_len, src->d_read_len, src->n_meta * 4);
  if (rc == SIGAX_OK) rc = copy((void**)&ix->d_name_rank, src->d_name_rank, src->n_meta * 4);
  if (rc != SIGAX_OK) {
    sigax_index_close(ix);
    return rc;
  }
  if (ix->fwd_only) {
    ix->tab_state = new std::atomic<int>(0);
    ix->deep_state = new std::atomic<int>(0);
  } else {
    build_rowend(ix);  // plans its own row tables; built on this device once it is reused (or prepared)
  }
  *out = ix;
  return SIGAX_OK;
}

extern "C" int sigax_index_info_get(const sigax_index* ix, sigax_index_info* out) {
  if (!ix || !out) return fail(SIGAX_E_ARG, "NULL argument");
  out->n_symbols = ix->n_symbols;
  out->n_strings = ix->n_strings;
  out->device_bytes = ix->device_bytes;
  for (int k = 0; k < 5; ++k) out->pred[k] = ix->st[0].C[k];
  out->device = ix->device;
  out->wide = ix->wide ? 1 : 0;
  return SIGAX_OK;
}

extern "C" int sigax_index_set_reads(sigax_index* ix, const uint32_t* lengths, const uint32_t* name_rank, uint64_t n) {
  if (!ix || !lengths || !name_rank) return fail(SIGAX_E_ARG, "NULL argument");
  if (n != ix->n_strings) return fail(SIGAX_E_ARG, "%llu reads given, index holds %llu", (u64)n, ix->n_strings);
  HIP_TRY(hipSetDevice(ix->device));
  if (ix->d_read_len) hipFree(ix->d_read_len);
  if (ix->d_name_rank) hipFree(ix->d_name_rank);
  ix->d_read_len = ix->d_name_rank = nullptr;
  int rc = upload(lengths, n * 4, (void**)&ix->d_read_len, &ix->device_bytes);
  if (rc == SIGAX_OK) rc = upload(name_rank, n * 4, (void**)&ix->d_name_rank, &ix->device_bytes);
  if (rc == SIGAX_OK) {
    ix->n_meta = n;
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n; ++i) mx = std::max(mx, lengths[i]);
    ix->max_read_len = mx;
    if (ix->tab_plan) plan_row_tables(ix);  // planned with an estimate of the longest stretch, not started yet: now with the bound
  }
  return rc;
}

// Are the BWT rows of strand `which` in the suffix order of record?  Checked on the device from the row table and the
// stretch text (built now if they were only planned): every pair of adjacent rows.  For tests of the index builder at
// sizes no second suffix sorter reaches in reasonable time.
extern "C" int sigax_index_check_order(sigax_index* ix, int which, uint64_t* n_bad, uint64_t* first_bad, uint64_t* n_undecided) {
  if (!ix || which < 0 || which > 1 || !n_bad) return fail(SIGAX_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(ix->device));
  if (!ix->d_sai[which] || !ix->d_read_len) return fail(SIGAX_E_STATE, "the order check needs the .sai tables and sigax_index_set_reads()");
  if (ix->st[which].C[1] != ix->n_strings) return fail(SIGAX_E_STATE, "reads with non-ACGT bases: stretches are not reads, order not checkable");
  {
    std::lock_guard<std::mutex> lock(*ix->enqueue_mu);
    row_tables_now(ix);
  }
  if (!ix->st[which].text) return fail(SIGAX_E_STATE, "no extractor tables on this index (memory short or turned off)");
  // The check reads the suffix array.  An index that runs on direct maps has none: a bare row table of this strand is
  // built for the duration of the call (two LF walks over the strand).
  FmStrand cs = ix->st[which];
  DevGuard tg;
  if (!cs.sa) {
    const u64 n_stretch = cs.C[1];
    void* info = nullptr;
    u32* d_max = nullptr;
    HIP_TRY(tg.alloc(&info, std::max<u64>(n_stretch, 1) * 8));
    HIP_TRY(tg.alloc((void**)&d_max, 8));
    HIP_TRY(hipMemset(d_max, 0, 8));
    launch_stretch_scan(cs, ix->wide, n_stretch, (u64*)info, d_max, nullptr);
    HIP_TRY(hipGetLastError());
    u32 maxlen = 0;
    HIP_TRY(hipMemcpy(&maxlen, d_max, 4, hipMemcpyDeviceToHost));
    const RowTabGeom g = row_tab_geom(ix, maxlen, 0);
    if (g.sa_bits > 57) return fail(SIGAX_E_STATE, "stretches too long for a row table");
    void* sa = nullptr;
    HIP_TRY(tg.alloc(&sa, g.sa_bytes));
    HIP_TRY(hipMemset(sa, 0, g.sa_bytes));
    launch_rows_fill(cs, ix->wide, n_stretch, (const u64*)info, (unsigned char*)sa, g.sa_bits, g.ld_bits, g.t_bits, nullptr, g.text_stride, nullptr,
                     nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    cs.sa = (const unsigned char*)sa;
    cs.sa_bits = g.sa_bits;
    cs.ld_bits = g.ld_bits;
    cs.t_bits = g.t_bits;
  }
  DevGuard g;
  uint32_t* isai = nullptr;
  u64* bad = nullptr;
  HIP_TRY(g.alloc((void**)&isai, ix->n_strings * 4));
  HIP_TRY(g.alloc((void**)&bad, 32));
  const u64 init[4] = {0, ~0ull, 0, 0};
  HIP_TRY(hipMemcpy(bad, init, 32, hipMemcpyHostToDevice));
  launch_suffix_order_check(cs, ix->d_sai[which], isai, ix->d_read_len, ix->n_strings, bad, nullptr);
  HIP_TRY(hipGetLastError());
  u64 out[4];
  HIP_TRY(hipMemcpy(out, bad, 32, hipMemcpyDeviceToHost));
  *n_bad = out[0];
  if (first_bad) *first_bad = out[1];
  if (n_undecided) *n_undecided = out[2];
  return SIGAX_OK;
}

extern "C" int sigax_occ_batch(sigax_index* ix, int which, const uint64_t* positions, uint64_t n, uint64_t* counts5) {
  if (!ix || (n && (!positions || !counts5)) || which < 0 || which > 1) return fail(SIGAX_E_ARG, "bad argument");
  if (which == 1 && ix->fwd_only) return fail(SIGAX_E_STATE, "the index was opened without its reverse strand");
  HIP_TRY(hipSetDevice(ix->device));
  if (n == 0) return SIGAX_OK;
  u64 *d_pos = nullptr, *d_out = nullptr;
  DevGuard g;
  HIP_TRY(g.alloc((void**)&d_pos, n * 8));
  HIP_TRY(g.alloc((void**)&d_out, n * 40));
  HIP_TRY(hipMemcpy(d_pos, positions, n * 8, hipMemcpyHostToDevice));
  launch_occ_batch(ix->st[which], ix->wide, d_pos, n, d_out, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(counts5, d_out, n * 40, hipMemcpyDeviceToHost));
  return SIGAX_OK;
}

extern "C" int sigax_kmer_count_batch(sigax_index* ix, const char* kmers, uint32_t k, uint64_t n, uint64_t* counts) {
  if (!ix || k == 0 || (n && (!kmers || !counts))) return fail(SIGAX_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(ix->device));
  if (n == 0) return SIGAX_OK;
  unsigned char* d_k = nullptr;
  u64* d_out = nullptr;
  DevGuard g;
  HIP_TRY(g.alloc((void**)&d_k, n * k));
  HIP_TRY(g.alloc((void**)&d_out, n * 8));
  HIP_TRY(hipMemcpy(d_k, kmers, n * k, hipMemcpyHostToDevice));
  launch_kmer_count(ix->st[0], ix->wide, d_k, k, n, d_out, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(counts, d_out, n * 8, hipMemcpyDeviceToHost));
  return SIGAX_OK;
}

// the 12-mer table of the k-mer lookups, built on first use (an accelerator: without it every lookup walks all its steps)
// Built by the first correction call ON THAT CALL'S STREAM (no device-wide wait: sigax_correct_device stays asynchronous;
// the allocation itself is the one synchronous step); later calls on other streams wait for the build's event.
static void ensure_prefix_table(sigax_index* ix, hipStream_t st) {
  std::lock_guard<std::mutex> lock(*ix->enqueue_mu);
  if (ix->ptab_tried) {
    if (ix->d_ptab && ix->ptab_ev) (void)hipStreamWaitEvent(st, ix->ptab_ev, 0);
    return;
  }
  ix->ptab_tried = true;
  // 0 = none, 8 .. 14 = that many symbols.  Default 13 (537 MB): measured at BASELINE configs[3], k = 31, 28.1 / 31.3 / 29.5 M
  // reads/s with 12 / 13 / 14 symbols (21.1 M without) -- the 2 GB table of all 14-mers no longer sits in the caches
  const char* env = getenv("SIGAX_KMER_PREFIX");
  uint32_t pk = env ? (uint32_t)atoi(env) : 13u;
  if (pk == 0) return;
  pk = std::min(std::max(pk, 8u), 14u);
  void* tab = nullptr;
  if (hipMalloc(&tab, prefix_table_bytes(ix->wide, pk)) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  launch_prefix_build(ix->st[0], ix->wide, tab, pk, st);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ptab_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(ix->ptab_ev, st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(st);
    hipFree(tab);
    return;
  }
  ix->d_ptab = tab;
  ix->ptab_k = pk;
  ix->device_bytes += prefix_table_bytes(ix->wide, pk);
}

// The corrector's k-mer table (see sigax_index): (re)built when a correction call comes with another k.  Synchronous (the
// build takes 0.1 s at BASELINE configs[3]; it happens once per index and k); an accelerator: whatever fails leaves the
// prefix-table + walk path in charge.  SIGAX_KMER_TABLE=0 turns it off.
static void ensure_kmer_table(sigax_index* ix, uint32_t k) {
  std::lock_guard<std::mutex> lock(*ix->enqueue_mu);
  if (ix->ktab_k == k || ix->ktab_tried_k == k) return;
  ix->ktab_tried_k = k;
  static const char* env = getenv("SIGAX_KMER_TABLE");
  if ((env && env[0] == '0') || k < 8 || k > SIGAX_DEEP_KMAX || ix->n_symbols == 0 || ix->st[0].C[1] >= 0xFFFFFFFFull) return;
  const bool verbose = getenv("SIGAX_VERBOSE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (ix->d_ktab) {  // a table for another k: no correction call is running on it (the caller serialises calls that change k)
    (void)hipDeviceSynchronize();
    hipFree(ix->d_ktab);
    ix->d_ktab = nullptr;
    ix->device_bytes -= ix->ktab_bytes;
    ix->ktab_k = 0;
    ix->ktab_bytes = 0;
  }
  const u64 n_stretch = ix->st[0].C[1];
  FmStrand f = ix->st[0];
  hipStream_t sb = nullptr;
  if (hipStreamCreateWithFlags(&sb, hipStreamNonBlocking) != hipSuccess) return;
  bool ok = true;
  const uint32_t* slen = ix->d_slen[0];
  if (!(f.sa && f.text && slen)) {
    // forward row table (bare entries) + text + stretch lengths of our own
    if (!ix->d_csa) {
      void* info = nullptr;
      u32* d_max = nullptr;
      uint32_t maxlen = 0;
      ok = hipMalloc(&info, std::max<u64>(n_stretch, 1) * 8) == hipSuccess && hipMalloc((void**)&d_max, 8) == hipSuccess &&
           hipMemsetAsync(d_max, 0, 8, sb) == hipSuccess;
      if (ok) {
        launch_stretch_scan(ix->st[0], ix->wide, n_stretch, (u64*)info, d_max, sb);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&maxlen, d_max, 4, hipMemcpyDeviceToHost, sb) == hipSuccess &&
             hipStreamSynchronize(sb) == hipSuccess;
      }
      RowTabGeom g = row_tab_geom(ix, maxlen, 0);
      ok = ok && g.sa_bits <= 57;
      size_t mfree = 0, mtotal = 0;
      (void)hipMemGetInfo(&mfree, &mtotal);
      ok = ok && g.sa_bytes + g.text_bytes + n_stretch * 4 < (u64)mfree / 2;
      ok = ok && hipMalloc(&ix->d_csa, g.sa_bytes) == hipSuccess && hipMalloc(&ix->d_ctext, g.text_bytes) == hipSuccess &&
           hipMalloc((void**)&ix->d_cslen, std::max<u64>(n_stretch, 1) * 4) == hipSuccess &&
           hipMemsetAsync(ix->d_csa, 0, g.sa_bytes, sb) == hipSuccess && hipMemsetAsync(ix->d_ctext, 0, g.text_bytes, sb) == hipSuccess;
      if (ok) {
        launch_rows_fill(ix->st[0], ix->wide, n_stretch, (const u64*)info, (unsigned char*)ix->d_csa, g.sa_bits, g.ld_bits, g.t_bits,
                         (unsigned char*)ix->d_ctext, g.text_stride, ix->d_cslen, sb);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(sb) == hipSuccess;
      }
      if (info) hipFree(info);
      if (d_max) hipFree(d_max);
      if (ok) {
        ix->csa_bits = g.sa_bits;
        ix->cld_bits = g.ld_bits;
        ix->ct_bits = g.t_bits;
        ix->ctext_stride = g.text_stride;
        ix->device_bytes += g.sa_bytes + g.text_bytes + n_stretch * 4;
      } else {
        (void)hipGetLastError();
        if (ix->d_csa) hipFree(ix->d_csa);
        if (ix->d_ctext) hipFree(ix->d_ctext);
        if (ix->d_cslen) hipFree(ix->d_cslen);
        ix->d_csa = ix->d_ctext = nullptr;
        ix->d_cslen = nullptr;
      }
    }
    if (ok && ix->d_csa) {
      f.sa = (const unsigned char*)ix->d_csa;
      f.text = (const unsigned char*)ix->d_ctext;
      f.xmap = nullptr;
      f.sa_bits = ix->csa_bits;
      f.ld_bits = ix->cld_bits;
      f.t_bits = ix->ct_bits;
      f.text_stride = ix->ctext_stride;
      slen = ix->d_cslen;
    } else {
      ok = false;
    }
  }
  u64* d_cnt = nullptr;
  u64* list = nullptr;
  void* tab = nullptr;
  u64 h[2] = {0, 0}, distinct = 0, slots = 0;
  ok = ok && hipMalloc((void**)&d_cnt, 16) == hipSuccess && hipMemsetAsync(d_cnt, 0, 16, sb) == hipSuccess;
  if (ok) {
    launch_deep_scan(f, slen, n_stretch, k, d_cnt, nullptr, 0, sb);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, sb) == hipSuccess && hipStreamSynchronize(sb) == hipSuccess;
    distinct = h[0];
  }
  if (ok) {
    size_t mfree = 0, mtotal = 0;
    (void)hipMemGetInfo(&mfree, &mtotal);
    slots = std::max<u64>(64, distinct * 2 + 16);
    ok = distinct < (1ull << 32) - 256 && slots * deep_entry_bytes() + distinct * 8 < (u64)mfree / 100 * 45;
  }
  ok = ok && hipMalloc((void**)&list, std::max<u64>(distinct, 1) * 8) == hipSuccess && hipMalloc(&tab, slots * deep_entry_bytes()) == hipSuccess &&
       hipMemsetAsync(tab, 0, slots * deep_entry_bytes(), sb) == hipSuccess && hipMemsetAsync(d_cnt, 0, 16, sb) == hipSuccess;
  if (ok) {
    launch_deep_scan(f, slen, n_stretch, k, d_cnt, list, distinct, sb);
    launch_deep_fill(f, f, ix->wide, slen, n_stretch, k, list, distinct, tab, slots, d_cnt + 1, sb);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, sb) == hipSuccess && hipStreamSynchronize(sb) == hipSuccess &&
         h[0] == distinct && h[1] == 0;
  }
  if (list) hipFree(list);
  if (d_cnt) hipFree(d_cnt);
  (void)hipStreamDestroy(sb);
  if (!ok) {
    (void)hipGetLastError();
    if (tab) hipFree(tab);
    if (verbose) fprintf(stderr, "[sigax] k-mer table for k = %u not built: the corrector walks\n", k);
    return;
  }
  ix->d_ktab = tab;
  ix->ktab_slots = slots;
  ix->ktab_k = k;
  ix->ktab_bytes = slots * deep_entry_bytes();
  ix->device_bytes += ix->ktab_bytes;
  if (verbose)
    fprintf(stderr, "[sigax] k-mer table: k = %u, %llu distinct k-mers, %.2f GB, %.3f s\n", k, distinct, ix->ktab_bytes / 1e9,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

static CorrectArgs correct_args(sigax_index* ix, const unsigned char* d_seqs, const unsigned char* d_quals, const u64* d_offs, u64 n_reads,
                                uint32_t kmer_size, int32_t kmer_threshold, uint32_t kmer_rounds, uint32_t count_offset,
                                unsigned char* d_out, unsigned char* d_valid, u64* d_stat, hipStream_t st) {
  CorrectArgs ca;
  ca.fwd = ix->st[0];
  ca.seqs = d_seqs;
  ca.quals = d_quals;
  ca.offs = d_offs;
  ca.n_reads = n_reads;
  ca.k = kmer_size;
  ca.low = (uint32_t)std::max(kmer_threshold, 0);       // CorrectThreshold::minSupport (src/correct_processor.cpp:28-31)
  ca.high = (uint32_t)std::max(kmer_threshold + 1, 0);
  ca.cutoff = 20;
  ca.rounds = kmer_rounds;
  ca.offset = count_offset;
  ca.out = d_out;
  ca.valid = d_valid;
  ca.dstat = d_stat;
  ensure_kmer_table(ix, kmer_size);
  ca.ktab = ix->ktab_k == kmer_size ? ix->d_ktab : nullptr;
  ca.ktab_slots = ix->ktab_slots;
  ensure_prefix_table(ix, st);
  ca.ptab = ix->d_ptab;
  ca.pk = ix->ptab_k;
  ca.max_len = 0;  // unknown here: sigax_correct_batch sees the offsets and says
  ca.only_deferred = 0;
  return ca;
}

extern "C" int sigax_correct_device(sigax_index* ix, const void* d_seqs, const void* d_quals, const void* d_offs, uint64_t n_reads,
                                    uint32_t kmer_size, int32_t kmer_threshold, uint32_t kmer_rounds, uint32_t count_offset,
                                    void* d_out_seqs, void* d_valid, void* d_stat4, void* stream) {
  if (!ix || kmer_size == 0 || (n_reads && (!d_seqs || !d_offs || !d_out_seqs || !d_valid || !d_stat4))) return fail(SIGAX_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(ix->device));
  if (n_reads == 0) return SIGAX_OK;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(d_stat4, 0, 32, st));
  CorrectArgs ca = correct_args(ix, (const unsigned char*)d_seqs, (const unsigned char*)d_quals, (const u64*)d_offs, n_reads, kmer_size,
                                kmer_threshold, kmer_rounds, count_offset, (unsigned char*)d_out_seqs, (unsigned char*)d_valid, (u64*)d_stat4, st);
  launch_correct(ca, ix->wide, st);
  HIP_TRY(hipGetLastError());
  return SIGAX_OK;
}

extern "C" int sigax_correct_batch(sigax_index* ix, const char* seqs, const char* quals, const uint64_t* offs, uint32_t n_reads,
                                   uint32_t kmer_size, int32_t kmer_threshold, uint32_t kmer_rounds, uint32_t count_offset,
                                   char* out_seqs, uint8_t* valid) {
  if (!ix || kmer_size == 0 || (n_reads && (!seqs || !offs || !out_seqs || !valid))) return fail(SIGAX_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(ix->device));
  if (n_reads == 0) return SIGAX_OK;
  const u64 nb = offs[n_reads];
  unsigned char *d_seqs = nullptr, *d_quals = nullptr, *d_out = nullptr, *d_valid = nullptr;
  u64 *d_offs = nullptr, *d_stat = nullptr;
  DevGuard g;
  HIP_TRY(g.alloc((void**)&d_seqs, nb + 16));
  HIP_TRY(g.alloc((void**)&d_out, nb + 16));
  HIP_TRY(g.alloc((void**)&d_offs, ((size_t)n_reads + 1) * 8));
  HIP_TRY(g.alloc((void**)&d_valid, (size_t)n_reads + 16));
  HIP_TRY(g.alloc((void**)&d_stat, 64));
  HIP_TRY(hipMemset(d_stat, 0, 64));
  HIP_TRY(hipMemcpy(d_seqs, seqs, nb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_offs, offs, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
  if (quals) {
    HIP_TRY(g.alloc((void**)&d_quals, nb + 16));
    HIP_TRY(hipMemcpy(d_quals, quals, nb, hipMemcpyHostToDevice));
  }
  CorrectArgs ca = correct_args(ix, d_seqs, d_quals, d_offs, n_reads, kmer_size, kmer_threshold, kmer_rounds, count_offset, d_out,
                                d_valid, d_stat, (hipStream_t)0);
  for (uint32_t i = 0; i < n_reads; ++i) ca.max_len = std::max<uint32_t>(ca.max_len, (uint32_t)std::min<u64>(offs[i + 1] - offs[i], 0xFFFFFFFFull));
  ca.max_len = std::max(ca.max_len, 1u);
  launch_correct(ca, ix->wide, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  u64 toolong = 0;
  HIP_TRY(hipMemcpy(&toolong, d_stat, 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_seqs, d_out, nb, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(valid, d_valid, n_reads, hipMemcpyDeviceToHost));
  if (toolong) return fail(SIGAX_E_ARG, "%llu reads are longer than the 1024 bases the correction kernel supports", toolong);
  return SIGAX_OK;
}

// Slots per chain of the candidate arena for reads of up to max_len bases.  Worst case: overlaps of length max(m,1)..L-1,
// plus one for the containment block.  Given: what the longest chain so far needed plus headroom -- or, before any run has
// finished, a third of the worst case -- never more than the worst case, never less than `floor_slots` records.  Even:
// each chain's slots start on a 64-byte line (k_find's quad stores write whole lines).
static uint32_t worst_cap(uint32_t max_len, uint32_t minov) {
  const uint32_t mm = std::max<uint32_t>(minov, 1u);
  const uint32_t cap = (max_len > mm ? max_len - mm : 0u) + 1u;
  return (cap + 1u) & ~1u;
}
static uint32_t chain_cap(const sigax_index* ix, uint32_t max_len, uint32_t minov, uint32_t floor_slots) {
  const uint32_t worst = worst_cap(max_len, minov);
  static const char* env = getenv("SIGAX_CAND_CAP");  // "worst" = the round-2 sizing; a number = slots of the first try (tests)
  if (env && env[0] == 'w') return worst;
  const uint32_t seen = ix->cap_seen->load();
  uint32_t want = seen ? seen + seen / 8 + 3 : (env ? (uint32_t)atoi(env) : std::max<uint32_t>(worst / 3, 16u));
  want = std::max(want, floor_slots + 1u);  // + the containment slot
  want = (want + 1u) & ~1u;
  return std::min(worst, std::max(want, 2u));
}

// ------------------------------------------------------------------------------------------------------
// batch workspace
// ------------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p;
  size_t bytes;
  DevBuf() : p(nullptr), bytes(0) {}
};

static int ensure(DevBuf* b, size_t bytes) {
  if (bytes <= b->bytes && b->p) return SIGAX_OK;
  if (b->p) hipFree(b->p);
  b->p = nullptr;
  b->bytes = 0;
  size_t want = bytes ? bytes : 16;
  hipError_t e = hipMalloc(&b->p, want);
  if (e != hipSuccess) return fail(SIGAX_E_DEVICE, "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e));
  b->bytes = want;
  return SIGAX_OK;
}

enum { EV_START = 0, EV_FX_DONE, EV_ORDER, EV_EDGES, EV_ORD0, EV_ORD1, EV_COUNT };
// per sub-batch: find begin/end on the find stream, fast begin/end and general end on the filter/extract stream
enum { SV_F0 = 0, SV_F1, SV_X0, SV_X1, SV_G1, SV_COUNT };

struct sigax_batch {
  sigax_index* ix;
  uint32_t max_reads;
  u64 max_bases;
  uint32_t max_len;
  DevBuf seqs_own, offs_own;
  const unsigned char* d_seqs;
  const u64* d_offs;
  uint32_t n_reads;
  u64 n_bases;
  uint32_t cur_max_len;
  // parameters of the last run (for the regrow-and-rerun loop)
  uint32_t read_base, minov, flags;
  const uint32_t* d_ids;  // the reads' ids in the index's read table, or NULL: read_base + r (sigax_batch_set_device_read_ids)
  uint32_t ids_n;         // ... for this many reads
  bool ran;
  // arenas
  DevBuf ids_own;  // sigax_batch_upload_read_ids
  DevBuf arena, chain_cnt, pool, wpool, work, work64, work64b, work64c, work64d, perm, ord_keys, ord_tmp, occ_side, slow_flag, offs2, item_base, fin, fin_cnt, substring, block_offs, outb, edge_cnt,
      edge_offs, edges, partial, dstat;
  uint32_t cap;
  uint32_t cap_floor;  // records the longest chain of this batch's last (overflowed) run needed
  uint32_t pool_cap;
  unsigned fx_grid;
  u64 fin_cap, edge_cap;
  bool fin_grown;
  u64 n_reruns;  // runs repeated because an arena was too small
  hipEvent_t ev[EV_COUNT];
  hipEvent_t sev[SIGAX_MAX_SUB][SV_COUNT];
  unsigned nsub;
  bool perm_valid;    // `perm_cur` (one half of `perm`) holds the locality order of the reads now set, for perm_nsub sub-batches
  const uint32_t* perm_cur;
  unsigned perm_nsub;
  u64 qhint[4];       // items per sub-batch in the four filter/extract queues in the previous run (~0: none yet)
  bool qhint_lean_off;  // ... measured with lean_off in this state
  bool lean_off;      // see sigax_batch_finish
  bool fx_heavy;      // the last run's filter/extract launches took longer than its finder launches (sigax_batch_finish)
  // which finder between 2^30 and 2^31 symbols (want_coop): 0, 1 = per-lane runs (the second one timed), 2, 3 = cooperative runs
  // (the second one timed), 4 = decided
  int coop_tune;
  float coop_t_lane;
  bool coop_pick;
  unsigned lean_off_runs;
  unsigned nsub_req;  // 0 = automatic
  unsigned find_per_sub;  // finder launches per sub-batch (2 = one per strand's two-step table)
  bool last_two_step, last_coop, last_perm, last_ordered;  // what the last enqueued run's finder did (sigax_batch_run_info)
  uint32_t last_deep_k;
  sigax_stats last;
  u64 last_total_blocks, last_total_edges;
  bool finished;
};

extern "C" void sigax_batch_destroy(sigax_batch* b) {
  if (!b) return;
  hipSetDevice(b->ix->device);
  DevBuf* all[] = {&b->ids_own, &b->seqs_own, &b->offs_own, &b->arena, &b->chain_cnt, &b->pool, &b->wpool, &b->work, &b->work64, &b->work64b, &b->work64c, &b->work64d, &b->perm, &b->ord_keys, &b->ord_tmp, &b->occ_side, &b->slow_flag, &b->offs2, &b->item_base, &b->fin,
                   &b->fin_cnt, &b->substring, &b->block_offs, &b->outb, &b->edge_cnt, &b->edge_offs, &b->edges,
                   &b->partial, &b->dstat};
  for (DevBuf* d : all)
    if (d->p) hipFree(d->p);
  for (int i = 0; i < EV_COUNT; ++i)
    if (b->ev[i]) hipEventDestroy(b->ev[i]);
  for (int i = 0; i < SIGAX_MAX_SUB; ++i)
    for (int j = 0; j < SV_COUNT; ++j)
      if (b->sev[i][j]) hipEventDestroy(b->sev[i][j]);
  delete b;
}

extern "C" int sigax_batch_create(sigax_index* ix, uint32_t max_reads, uint64_t max_bases, uint32_t max_read_len,
                                  sigax_batch** out) {
  if (!ix || !out) return fail(SIGAX_E_ARG, "NULL argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(ix->device));
  sigax_batch* b = new sigax_batch();
  b->ix = ix;
  b->max_reads = max_reads;
  b->max_bases = max_bases;
  b->max_len = max_read_len;
  b->d_seqs = nullptr;
  b->d_offs = nullptr;
  b->n_reads = 0;
  b->n_bases = 0;
  b->cur_max_len = 0;
  b->read_base = b->minov = b->flags = 0;
  b->d_ids = nullptr;
  b->ids_n = 0;
  b->ran = b->finished = false;
  b->cap = 0;
  b->cap_floor = 0;
  b->pool_cap = 0;
  b->fx_grid = 0;
  b->fin_cap = b->edge_cap = 0;
  b->fin_grown = false;
  b->n_reruns = 0;
  memset(&b->last, 0, sizeof(b->last));
  b->last_total_blocks = b->last_total_edges = 0;
  for (int i = 0; i < EV_COUNT; ++i) b->ev[i] = nullptr;
  for (int i = 0; i < SIGAX_MAX_SUB; ++i)
    for (int j = 0; j < SV_COUNT; ++j) b->sev[i][j] = nullptr;
  b->nsub = 1;
  b->lean_off = false;
  b->lean_off_runs = 0;
  b->fx_heavy = false;
  b->coop_tune = 0;
  b->coop_t_lane = 0.f;
  b->coop_pick = false;
  b->qhint[0] = b->qhint[1] = b->qhint[2] = b->qhint[3] = ~0ull;
  b->qhint_lean_off = false;
  b->perm_valid = false;
  b->perm_cur = nullptr;
  b->perm_nsub = 0;
  b->nsub_req = 0;
  b->find_per_sub = 1;
  b->last_two_step = b->last_coop = b->last_perm = b->last_ordered = false;
  b->last_deep_k = 0;
  {
    hipError_t e = hipSuccess;
    for (int i = 0; i < EV_COUNT && e == hipSuccess; ++i) e = hipEventCreate(&b->ev[i]);
    for (int i = 0; i < SIGAX_MAX_SUB && e == hipSuccess; ++i)
      for (int j = 0; j < SV_COUNT && e == hipSuccess; ++j) e = hipEventCreate(&b->sev[i][j]);
    if (e != hipSuccess) {
      sigax_batch_destroy(b);
      return fail(SIGAX_E_DEVICE, "creating events/streams: %s", hipGetErrorString(e));
    }
  }
  int rc = ensure(&b->dstat, DS_COUNT * 8);
  if (rc != SIGAX_OK) {
    sigax_batch_destroy(b);
    return rc;
  }
  *out = b;
  return SIGAX_OK;
}

extern "C" int sigax_batch_upload(sigax_batch* b, const char* seqs, const uint64_t* offs, uint32_t n_reads, void* stream) {
  if (!b || (n_reads && (!seqs || !offs))) return fail(SIGAX_E_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(b->ix->device));
  hipStream_t st = (hipStream_t)stream;
  u64 nb = n_reads ? offs[n_reads] - offs[0] : 0;
  if (n_reads && offs[0] != 0) return fail(SIGAX_E_ARG, "offs[0] must be 0");
  uint32_t mx = 0;
  for (uint32_t i = 0; i < n_reads; ++i) {
    u64 l = offs[i + 1] - offs[i];
    if (offs[i + 1] < offs[i] || l > 0x0FFFFFFFull) return fail(SIGAX_E_ARG, "bad offsets at read %u", i);
    mx = std::max<uint32_t>(mx, (uint32_t)l);
  }
  int rc = ensure(&b->seqs_own, nb + 16);
  if (rc == SIGAX_OK) rc = ensure(&b->offs_own, ((size_t)n_reads + 1) * 8);
  if (rc != SIGAX_OK) return rc;
  if (nb) HIP_TRY(hipMemcpyAsync(b->seqs_own.p, seqs, nb, hipMemcpyHostToDevice, st));
  static const u64 zero = 0;
  HIP_TRY(hipMemcpyAsync(b->offs_own.p, n_reads ? (const void*)offs : (const void*)&zero, ((size_t)n_reads + 1) * 8,
                         hipMemcpyHostToDevice, st));
  b->d_seqs = (const unsigned char*)b->seqs_own.p;
  b->d_offs = (const u64*)b->offs_own.p;
  b->n_reads = n_reads;
  b->n_bases = nb;
  b->cur_max_len = mx;
  b->perm_valid = false;
  b->ran = b->finished = false;
  return SIGAX_OK;
}

extern "C" int sigax_batch_set_device_reads(sigax_batch* b, const void* d_seqs, const void* d_offs, uint32_t n_reads,
                                            uint64_t n_bases, uint32_t max_len) {
  if (!b || (n_reads && (!d_seqs || !d_offs))) return fail(SIGAX_E_ARG, "NULL argument");
  b->d_seqs = (const unsigned char*)d_seqs;
  b->d_offs = (const u64*)d_offs;
  b->n_reads = n_reads;
  b->n_bases = n_bases;
  b->cur_max_len = max_len;
  b->perm_valid = false;
  b->ran = b->finished = false;
  return SIGAX_OK;
}

// Cooperative or per-lane finder for this run?  The cooperative one (lines through LDS, eight lanes per line: one address
// translation) wherever the per-lane finder's 32-bit byte offsets do not reach (2^31 symbols; 64-bit positions); the per-lane
// one below 2^30 symbols.  In between it depends on what the library cannot see -- how local the batch's reads are: one
// rank's view of the 4-GPU job of bench.py, 1.51e9 symbols: file-range shard 117.6 M reads/s per lane against 111.0 M
// cooperative, key-range shard 132-134 M against 138-139 M (profiles/r04_key_sharding.txt) -- so a batch object measures:
// two runs per lane, two cooperative, the second of each timed (its finder launches' own durations, as they ran), then it
// keeps the faster.  SIGAX_FIND_COOP / SIGAX_COOP_MIN_SYMBOLS decide statically as before; SIGAX_COOP_TUNE_MIN moves the
// lower end of the measured range (tests: 0 = every index).
static bool coop_tunable(const sigax_index* ix) {
  static const char* env_coop = getenv("SIGAX_FIND_COOP");
  static const char* env_cmin = getenv("SIGAX_COOP_MIN_SYMBOLS");
  static const char* env_tmin = getenv("SIGAX_COOP_TUNE_MIN");
  if (env_coop || env_cmin || ix->wide) return false;
  const u64 lo = env_tmin ? strtoull(env_tmin, nullptr, 10) : (1ull << 30);
  return ix->n_symbols >= lo && ix->n_symbols < (1ull << 31);
}
static bool want_coop(const sigax_index* ix, const sigax_batch* b) {
  static const char* env_coop = getenv("SIGAX_FIND_COOP");
  static const char* env_cmin = getenv("SIGAX_COOP_MIN_SYMBOLS");
  if (env_coop) return env_coop[0] != '0';
  if (ix->wide || ix->n_symbols >= (env_cmin ? strtoull(env_cmin, nullptr, 10) : (1ull << 31))) return true;
  if (!coop_tunable(ix)) return false;
  return b->coop_tune >= 4 ? b->coop_pick : b->coop_tune >= 2;
}

extern "C" int sigax_batch_set_device_read_ids(sigax_batch* b, const void* d_ids, uint32_t n_reads) {
  if (!b) return fail(SIGAX_E_ARG, "NULL batch");
  if ((d_ids != nullptr) != (b->d_ids != nullptr)) b->coop_tune = 0;  // another kind of shard: measure the finders again (want_coop)
  b->d_ids = (const uint32_t*)d_ids;
  b->ids_n = d_ids ? n_reads : 0;
  b->ran = b->finished = false;
  return SIGAX_OK;
}

extern "C" int sigax_batch_upload_read_ids(sigax_batch* b, const uint32_t* ids, uint32_t n_reads, void* stream) {
  if (!b || (n_reads && !ids)) return fail(SIGAX_E_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(b->ix->device));
  for (uint32_t i = 0; i < n_reads; ++i)
    if (ids[i] >= b->ix->n_strings) return fail(SIGAX_E_ARG, "read id %u (entry %u) is beyond the %llu indexed reads", ids[i], i, (unsigned long long)b->ix->n_strings);
  int rc = ensure(&b->ids_own, ((size_t)n_reads + 1) * 4);
  if (rc != SIGAX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(b->ids_own.p, ids, (size_t)n_reads * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
  return sigax_batch_set_device_read_ids(b, n_reads ? b->ids_own.p : nullptr, n_reads);
}

static int enqueue(sigax_batch* b, hipStream_t st) {
  sigax_index* ix = b->ix;
  if (ix->fwd_only) return fail(SIGAX_E_STATE, "the index was opened without its reverse strand: overlap runs need <prefix>.rbwt too");
  std::lock_guard<std::mutex> lock(*ix->enqueue_mu);  // one batch's launch sequence at a time on the shared streams
  publish_tables(ix);
  publish_deep(ix);
  // a whole pass over the indexed reads has been asked for before this run: the index is being reused
  if (ix->reads_asked >= std::max<u64>(ix->n_strings, 1)) {
    start_row_tables(ix, false);
    if (!(b->flags & SIGAX_DUPLICATE)) start_deep_tables(ix, b->minov);
  }
  ix->reads_asked += b->n_reads;
  const uint32_t n = b->n_reads;
  const bool edges = (b->flags & SIGAX_EDGES) != 0;
  if (edges && (!ix->d_sai[0] || !ix->d_read_len))
    return fail(SIGAX_E_STATE, "SIGAX_EDGES needs the .sai tables and sigax_index_set_reads()");
  if (b->d_ids != nullptr && b->ids_n != n)
    return fail(SIGAX_E_STATE, "the batch holds read ids for %u reads and %u reads (sigax_batch_set_device_read_ids(NULL) forgets them)", b->ids_n, n);
  if (edges && b->d_ids == nullptr && (u64)b->read_base + n > ix->n_strings)
    return fail(SIGAX_E_ARG, "read_base + n_reads exceeds the indexed read set");
  b->cap = chain_cap(ix, b->cur_max_len, b->minov, b->cap_floor);
  int rc;
  if ((rc = ensure(&b->arena, (size_t)n * 4 * b->cap * cand_bytes(ix->wide))) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->chain_cnt, (size_t)n * 4 * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->fin_cnt, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->occ_side, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->slow_flag, ((size_t)n + 1) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->offs2, (2 * (size_t)n + 4) * 8)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->substring, (size_t)n + 16)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->block_offs, ((size_t)n + 2) * 8)) != SIGAX_OK) return rc;
  // fast filter/extract kernel: persistent waves (one read at a time per wave) with a private pool each
  // Three workgroups per CU: all of them are resident beside the finder's two (register file: 2 x 64 + 3 x 128 per
  // SIMD), so no filter/extract workgroup is left waiting to take the slot a finished finder workgroup frees.
  static const char* env_fxg = getenv("SIGAX_FX_GRID");
  // (round 4: 2.5 per CU beside the per-lane finder's three workgroups -- two fit a CU's LDS beside them, the rest queue --;
  // 3 per CU beside the cooperative finder as before)
  const bool coop_idx = want_coop(ix, b);
  // ... and 3 per CU again for a batch object whose last run waited for filter/extract rather than for the finder (reads
  // with sequencing errors: fx_heavy, see sigax_batch_finish).  tools/sweep_env.sh, BASELINE configs[1] shape, 2.5 against 3
  // per CU: no errors 155.9 / 153.6 M reads/s, 0.03 % substitutions 138.7 / 144.7, 0.1 % 121.7 / 121.9, 0.3 % 93.7 / 98.0,
  // 1 % 66.7 / 76.7 (profiles/r04_error_rates.txt).
  const unsigned grid_max = 3u * (unsigned)ix->n_cu;
  unsigned fast_grid = (unsigned)std::min<u64>(env_fxg ? (u64)atoi(env_fxg) : ((coop_idx || b->fx_heavy) ? grid_max : 5u * (unsigned)ix->n_cu / 2u), ((u64)n + 3) / 4);  // two items per wave
  if (fast_grid == 0) fast_grid = 1;
  // (the pools are sized for the larger grid: a change of mind costs no allocation)
  if ((rc = ensure(&b->wpool, (size_t)std::max(fast_grid, std::min<unsigned>(grid_max, (unsigned)(((u64)n + 3) / 4))) * 4 * fast_pool_entries_per_wave() * SIGAX_ENT_BYTES)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->work, ((size_t)n + 1) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->work64, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->work64b, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->work64c, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->work64d, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
  // general filter/extract kernel (reads the fast kernel queued): persistent lanes with a private pool each
  unsigned want_grid = (unsigned)std::min<u64>(128, ((u64)n + 255) / 256);
  if (want_grid == 0) want_grid = 1;
  // SIGAX_TEST_{POOL,FIN,EDGE}_CAP: tiny first sizes, so that tests reach the grow-and-rerun loop of sigax_batch_finish
  const char* t_pool = getenv("SIGAX_TEST_POOL_CAP");
  const char* t_fin = getenv("SIGAX_TEST_FIN_CAP");
  const char* t_edge = getenv("SIGAX_TEST_EDGE_CAP");
  uint32_t want_pool = std::max<uint32_t>(b->pool_cap, t_pool ? (uint32_t)atoi(t_pool) : 4u * (b->cap + 2u) + 128u);
  b->fx_grid = want_grid;
  b->pool_cap = want_pool;
  if ((rc = ensure(&b->pool, (size_t)want_grid * 256 * want_pool * SIGAX_ENT_BYTES)) != SIGAX_OK) return rc;
  // the unordered arena is handed out in chunks (one atomic per chunk): leave room for every wave's / lane's tail
  u64 chunk_slack = (u64)fast_grid * 4 * fast_fin_chunk() + (u64)want_grid * 256 * 64 + 1024;
  u64 want_fin = ((b->flags & SIGAX_IRREDUCIBLE) ? (u64)n * 8 : (u64)n * 64) + chunk_slack;
  if (t_fin) want_fin = std::max<u64>(strtoull(t_fin, nullptr, 10), 1);
  b->fin_cap = std::max<u64>(b->fin_cap, want_fin);
  if ((rc = ensure(&b->fin, b->fin_cap * sizeof(sigax_block))) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->outb, b->fin_cap * sizeof(sigax_block))) != SIGAX_OK) return rc;
  if ((rc = ensure(&b->item_base, (2 * (size_t)n + 2) * 8)) != SIGAX_OK) return rc;
  u64 scan_n = std::max<u64>(2 * (u64)n, b->fin_cap);
  if ((rc = ensure(&b->partial, scan_partials_needed(scan_n) * 8)) != SIGAX_OK) return rc;
  if (edges) {
    if (b->edge_cap == 0) b->edge_cap = t_edge ? std::max<u64>(strtoull(t_edge, nullptr, 10), 1) : b->fin_cap * 2 + 1024;
    if ((rc = ensure(&b->edge_cnt, (2 * (size_t)n + 2) * 4)) != SIGAX_OK) return rc;
    if ((rc = ensure(&b->edge_offs, (2 * (size_t)n + 4) * 8)) != SIGAX_OK) return rc;
    if ((rc = ensure(&b->edges, b->edge_cap * sizeof(sigax_edge))) != SIGAX_OK) return rc;
  }
  u64* dstat = (u64*)b->dstat.p;
  HIP_TRY(hipMemsetAsync(dstat, 0, DS_COUNT * 8, st));
  HIP_TRY(hipMemsetAsync(b->occ_side.p, 0, (2 * (size_t)n + 2) * 4, st));
  HIP_TRY(hipMemsetAsync(b->slow_flag.p, 0, ((size_t)n + 1) * 4, st));
  // Sub-batches: the finder is bound by the memory system's request rate, filter/extract by VALU issue; running
  // sub-batch i's filter/extract while sub-batch i+1's finder runs overlaps the two.
  static const char* env_sub = getenv("SIGAX_SUBBATCHES");
  unsigned nsub = b->nsub_req ? b->nsub_req : env_sub ? (unsigned)atoi(env_sub) : (n >= 4 * 131072u ? 4u : n >= 2 * 131072u ? 2u : 1u);
  if (nsub < 1) nsub = 1;
  if (nsub > SIGAX_MAX_SUB) nsub = SIGAX_MAX_SUB;
  b->nsub = nsub;
  // Locality order of the reads for the per-lane finder (sigax_order_reads): once per set of reads and sub-batch count,
  // on the caller's stream behind the upload (no host wait: the pipeline streams start behind EV_START); only where the
  // workgroup's reads are staged in LDS by slot (the longest read decides).
  // SIGAX_READ_ORDER=0 turns it off.
  const uint32_t* d_perm = nullptr;
  b->last_ordered = false;
  const uint32_t perm_stride = (b->cur_max_len + 3u) & ~3u;
  {
    // Off unless SIGAX_READ_ORDER=1.  Measured with the ordering inside the timed step, as every product batch pays it:
    // BASELINE configs[1] 107-111 M reads/s with it against 119-121 M without (the table is cache-resident there and the
    // order buys the finder 1 to 5 %); the configs[2] shape 87.2 / 89.3 M against 90.9 / 89.6 M (3.75-fold coverage inside a
    // batch: the finder gains 0.4 ms of 26.7, the batch's chain gets 7 ms longer); the configs[4] shape 35.7 against 38.8 M
    // (a batch covers the genome 1.1 times: hardly any read has a neighbour in its batch).  What reads in GENOME order are
    // worth (+14 % / +23 %) needs them sorted by position inside a class too -- round 2's full radix sort got
    // the configs[2] finder from 26.9 to 22.4 ms -- and that costs twenty dependent launches on a GPU the other batches
    // keep full: 17 to 40 ms on the batch's chain.  Kept as an option for callers whose batches are deep.
    static const char* env_ord = getenv("SIGAX_READ_ORDER");
    const bool order_on = env_ord != nullptr && env_ord[0] != '0';
    const bool coop_would = (ix->st[0].gran2 && ix->st[1].gran2) && want_coop(ix, b) && 32ull * perm_stride + 32 <= 32768;
    if (order_on && n >= 2 && (coop_would || 128ull * perm_stride + 8 <= find_stage_capacity(b->cur_max_len))) {
      if (!b->perm_valid || b->perm_nsub != nsub) {
        HIP_TRY(hipEventRecord(b->ev[EV_ORD0], st));  // behind the upload of the reads
        HIP_TRY(hipStreamWaitEvent(ix->s_ord, b->ev[EV_ORD0], 0));
        b->last_ordered = true;
        const size_t tb = sigax_order_reads_tmp_bytes(nsub);
        if ((rc = ensure(&b->ord_keys, (size_t)n * 4)) != SIGAX_OK) return rc;
        if ((rc = ensure(&b->perm, (size_t)n * 4)) != SIGAX_OK) return rc;
        if ((rc = ensure(&b->ord_tmp, tb)) != SIGAX_OK) return rc;
        uint32_t bounds[SIGAX_MAX_SUB + 1];
        for (unsigned i = 0; i <= nsub; ++i) bounds[i] = (uint32_t)((u64)n * i / nsub);
        rc = sigax_order_reads(b->d_seqs, b->d_offs, n, b->cur_max_len, bounds, nsub, (uint32_t*)b->ord_keys.p, (uint32_t*)b->perm.p, b->ord_tmp.p, tb,
                               &b->perm_cur, ix->s_ord);
        if (rc != SIGAX_OK) return rc;
        HIP_TRY(hipEventRecord(b->ev[EV_ORD1], ix->s_ord));
        HIP_TRY(hipStreamWaitEvent(st, b->ev[EV_ORD1], 0));
        b->perm_valid = true;
        b->perm_nsub = nsub;
      }
      d_perm = b->perm_cur;
    }
  }
  HIP_TRY(hipEventRecord(b->ev[EV_START], st));

  static const bool only_general = getenv("SIGAX_GENERAL_ONLY") != nullptr;  // debugging aid: skip the fast kernel
  HIP_TRY(hipStreamWaitEvent(ix->s_find, b->ev[EV_START], 0));
  HIP_TRY(hipStreamWaitEvent(ix->s_fx, b->ev[EV_START], 0));
  for (unsigned i = 0; i < nsub; ++i) {
    const uint32_t rb = (uint32_t)((u64)n * i / nsub), re = (uint32_t)((u64)n * (i + 1) / nsub);
    FindArgs fa;
    fa.fwd = ix->st[0];
    fa.rev = ix->st[1];
    fa.seqs = b->d_seqs;
    fa.offs = b->d_offs;
    fa.n_reads = n;
    fa.minov = b->minov;
    fa.max_len = b->cur_max_len;
    fa.chain_mask = (b->flags & SIGAX_DUPLICATE) ? 0x9u : (b->flags & SIGAX_RC) ? 0xFu : 0x5u;
    fa.cap = b->cap;
    fa.max_seen = ix->cap_seen->load();
    fa.start_ok = (ix->st[0].start && ix->st[1].start && b->minov >= (uint32_t)SIGAX_START_K) ? 1u : 0u;
    static const bool deep_off = getenv("SIGAX_FIND_DEEP_USE") != nullptr && getenv("SIGAX_FIND_DEEP_USE")[0] == '0';  // A/B aid: built, not used
    fa.deep_k = (!deep_off && ix->st[0].deep && ix->st[1].deep && ix->st[0].deep_k == ix->st[1].deep_k && ix->st[0].deep_k <= b->minov) ? ix->st[0].deep_k : 0u;
    b->last_deep_k = fa.deep_k;
    fa.read_begin = rb;
    fa.read_end = re;
    fa.stage_bytes = 0;  // set by launch_find
    fa.two_step = (ix->st[0].gran2 && ix->st[1].gran2) ? 1u : 0u;
    static const char* env_mu = getenv("SIGAX_FIND_MASK_UPPER");  // A/B aid: 0 = ten loads for every lane
    fa.mask_upper = (env_mu && env_mu[0] == '0') ? 0u : 1u;
    {
      // From 2^30 symbols the cooperative finder is the faster one (round 3, one rank's view of the 2- / 4- / 8-GPU jobs of
      // bench.py: 7.6e8 symbols 93.6 M reads/s per lane vs 92.3 M cooperative; 1.5e9 symbols 76.7 vs 90.4 M; round 2, before
      // the cooperative finder's LDS diet, had 80 vs 66 M at 1.2e9); from 2^31 the per-lane finder's 32-bit byte offsets
      // no longer reach the table at all.
      // Round 4: with the deep start table and three workgroups per CU the per-lane finder leads again wherever it can reach
      // (same box, one rank's view of the 4-GPU job, 1.51e9 symbols: 117.6 M reads/s per lane vs 111.0 M cooperative; 2-GPU
      // job, 7.6e8: 128.6 vs 120.0 M), so the switch sits at its reach: 2^31 symbols.
      // the workgroup's 64 reads, staged as 4-bit ranks: one byte range, or by slot under the locality order
      const u64 need = d_perm ? 32ull * perm_stride + 32 : (64ull * b->cur_max_len + 16) / 2 + 16;
      const bool can = fa.two_step && need <= 32768;
      const bool want = want_coop(ix, b);
      fa.coop = (can && want) ? 1u : 0u;
      fa.coop_stage_bytes = (uint32_t)((need + 15) & ~15ull);
      // measurement aid: cap the grid at this many workgroups per CU (they then walk the tiles).  Not a way to set the
      // residency: the dispatcher packs a CU before it moves on, so a grid of 4 per CU fills two CUs in three with 6 each
      // (finder 10.3 ms per 1 M reads at C2 against 7.7 uncapped, tools/coop_c2.sh)
      static const char* env_cwg = getenv("SIGAX_FIND_COOP_WGS");
      fa.coop_grid = env_cwg ? (uint32_t)ix->n_cu * (uint32_t)atoi(env_cwg) : 0u;
      // per-lane gathers use 32-bit byte offsets into the two-step table: beyond 2^31 symbols only the cooperative form works
      if (fa.two_step && !fa.coop && ix->n_symbols >= (1ull << 31)) fa.two_step = 0;
    }
    fa.arena = b->arena.p;
    fa.chain_cnt = (uint32_t*)b->chain_cnt.p;
    fa.dstat = dstat;
    fa.perm = (fa.coop || 128ull * perm_stride + 8 <= find_stage_capacity(b->cur_max_len)) ? d_perm : nullptr;
    fa.stage_stride = perm_stride;
    b->last_two_step = fa.two_step != 0;
    b->last_coop = fa.coop != 0;
    b->last_perm = fa.perm != nullptr;
    HIP_TRY(hipEventRecord(b->sev[i][SV_F0], ix->s_find));
    fa.chain_base = 0;
    fa.chains_per_wg = 4;
    static const char* env_split = getenv("SIGAX_SPLIT_STRANDS");
    // (only while the 128 reads of such a workgroup still fit the LDS staging buffer: the double step needs them there)
    // Without the two-step tables (indexes of 1.6 G symbols and more) the same split keeps one launch's gathers inside one
    // strand's granule table once the two tables together pass the translation reach (profiles/r01_gather_probe.txt).
    const bool big_one_step = !fa.two_step && ix->n_symbols >= (1ull << 30);
    const bool split = fa.coop || ((env_split ? env_split[0] != '0' : ((fa.two_step && ix->split_strands) || big_one_step)) &&
                                   128ull * b->cur_max_len + 8 <= find_stage_capacity(b->cur_max_len));
    b->find_per_sub = split ? 2u : 1u;
    if (split) {
      // one launch per strand's two-step table (chains 0,1 gather from the forward index, 2,3 from the reverse one)
      fa.chains_per_wg = 2;
      launch_find(fa, ix->wide, ix->s_find);
      fa.chain_base = 2;
    }
    launch_find(fa, ix->wide, ix->s_find);
    HIP_TRY(hipEventRecord(b->sev[i][SV_F1], ix->s_find));
    HIP_TRY(hipStreamWaitEvent(ix->s_fx, b->sev[i][SV_F1], 0));

    FxArgs xa;
    xa.fwd = ix->st[0];
    xa.rev = ix->st[1];
    xa.offs = b->d_offs;
    static const bool fx_one_step = getenv("SIGAX_FX_ONE_STEP") != nullptr;  // A/B aid: extractor without the two-step table
    if (fx_one_step) xa.fwd.gran2 = xa.rev.gran2 = nullptr;
    xa.n_reads = n;
    xa.cap = b->cap;
    xa.irreducible = (b->flags & SIGAX_IRREDUCIBLE) ? 1u : 0u;
    static const bool skip_strict = getenv("SIGAX_FX_SKIP_STRICT") != nullptr;  // A/B aid
    xa.no_lean = (b->lean_off || skip_strict) ? 1u : 0u;
    xa.arena = b->arena.p;
    xa.chain_cnt = (const uint32_t*)b->chain_cnt.p;
    xa.n_map = ix->n_strings;
    xa.pool = (Ent*)b->pool.p;
    xa.pool_cap = b->pool_cap;
    xa.wpool = (Ent*)b->wpool.p;
    xa.work_out = (uint32_t*)b->work.p + rb;
    xa.slow_counter = dstat + DS_SLOW_BASE + i;
    xa.work64 = (uint32_t*)b->work64.p + 2 * (size_t)rb;
    xa.w64_counter = dstat + DS_W64_BASE + i;
    xa.work64b = (uint32_t*)b->work64b.p + 2 * (size_t)rb;
    xa.w64b_counter = dstat + DS_W64B_BASE + i;
    xa.work64c = (uint32_t*)b->work64c.p + 2 * (size_t)rb;
    xa.w64c_counter = dstat + DS_W64C_BASE + i;
    xa.work64d = (uint32_t*)b->work64d.p + 2 * (size_t)rb;
    xa.w64d_counter = dstat + DS_W64D_BASE + i;
    xa.q_in = nullptr;
    xa.q_in_n = nullptr;
    xa.q_out = nullptr;
    xa.q_out_n = nullptr;
    xa.q_wide = nullptr;
    xa.q_wide_n = nullptr;
    xa.read_begin = rb;
    xa.read_end = re;
    xa.item_base = (u64*)b->item_base.p;
    xa.fin = (sigax_block*)b->fin.p;
    xa.fin_cap = b->fin_cap;
    xa.fin_cnt = (uint32_t*)b->fin_cnt.p;
    xa.occ_side = (uint32_t*)b->occ_side.p;
    xa.slow_flag = (uint32_t*)b->slow_flag.p;
    xa.substring = (uint8_t*)b->substring.p;
    xa.dstat = dstat;
    HIP_TRY(hipEventRecord(b->sev[i][SV_X0], ix->s_fx));
    if (!only_general) {
      xa.work = nullptr;
      xa.n_work = 0;
      xa.n_work_ptr = nullptr;
      static const char* env_g64 = getenv("SIGAX_FX_GRID64");  // grid of the 64-lane launches (they size themselves down by their queues)
      const unsigned grid64 = env_g64 ? (unsigned)std::max(1, atoi(env_g64)) : 512u;
      launch_filter_extract_fast(xa, ix->wide, fast_grid, std::min(fast_grid, grid64), b->qhint_lean_off == b->lean_off ? b->qhint : nullptr, ix->s_fx);
      xa.work = (const uint32_t*)b->work.p + rb;  // the general kernel redoes what the fast one queued
      xa.n_work = 0;
      xa.n_work_ptr = dstat + DS_SLOW_BASE + i;
    } else {
      // every read of the sub-batch through the general kernel
      std::vector<uint32_t> ids(re - rb);
      for (uint32_t k = rb; k < re; ++k) ids[k - rb] = k;
      HIP_TRY(hipMemcpyAsync((uint32_t*)b->work.p + rb, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, ix->s_fx));
      HIP_TRY(hipStreamSynchronize(ix->s_fx));
      xa.work = (const uint32_t*)b->work.p + rb;
      xa.n_work = re - rb;
      xa.n_work_ptr = nullptr;
    }
    HIP_TRY(hipEventRecord(b->sev[i][SV_X1], ix->s_fx));
    // The general kernel (what the lane-group launches queued: usually nothing) goes to the high-priority tail stream: on
    // the low-priority one its 170-register workgroups found no room beside the finder's, whose own new workgroups took
    // every slot that came free first -- at the BASELINE configs[4] shape an EMPTY launch sat there for 3 to 6 ms per run
    // and held up the next batch's filter/extract chain behind it.
    HIP_TRY(hipStreamWaitEvent(ix->s_tail, b->sev[i][SV_X1], 0));
    launch_filter_extract(xa, ix->wide, b->fx_grid, ix->s_tail);
    HIP_TRY(hipEventRecord(b->sev[i][SV_G1], ix->s_tail));
  }
  HIP_TRY(hipEventRecord(b->ev[EV_FX_DONE], ix->s_fx));
  // The short tail (scan, ordered scatter, edge records) runs on its own high-priority stream: queued behind the
  // long kernels of the next batch on an ordinary stream it took ten times its own duration and held up the
  // batch's completion, i.e. the moment the caller can submit this batch object again.
  hipStream_t ts = ix->s_tail;
  HIP_TRY(hipStreamWaitEvent(ts, b->ev[EV_FX_DONE], 0));

  launch_scan((const uint32_t*)b->fin_cnt.p, 2 * (u64)n, (u64*)b->partial.p, (u64*)b->offs2.p, dstat + DS_TOTAL_BLOCKS, ts);
  launch_pick_read_offsets((const u64*)b->offs2.p, n, (u64*)b->block_offs.p, ts);
  OrderArgs oa;
  oa.fin = (const sigax_block*)b->fin.p;
  oa.item_base = (const u64*)b->item_base.p;
  oa.fin_cnt = (const uint32_t*)b->fin_cnt.p;
  oa.n_items = 2 * (u64)n;
  oa.fin_cap = b->fin_cap;
  oa.offs2 = (const u64*)b->offs2.p;
  oa.out = (sigax_block*)b->outb.p;
  oa.out_cap = b->fin_cap;
  oa.arena = b->arena.p;
  oa.cap = b->cap;
  oa.wide = ix->wide ? 1u : 0u;
  oa.item_edges = edges ? (uint32_t*)b->edge_cnt.p : nullptr;
  oa.read_base = b->read_base;
  oa.read_ids = b->d_ids;
  oa.n_index_reads = (uint32_t)std::min<u64>(ix->n_strings, 0xFFFFFFFFull);
  oa.bad_ids = dstat + DS_BAD_IDS;
  oa.sai = ix->d_sai[0];
  oa.rsai = ix->d_sai[1];
  oa.n_sai = ix->n_sai;
  oa.read_len = ix->d_read_len;
  oa.name_rank = ix->d_name_rank;
  launch_order_scatter(oa, ts);
  HIP_TRY(hipEventRecord(b->ev[EV_ORDER], ts));

  if (edges) {
    EdgeArgs ea;
    ea.blocks = (const sigax_block*)b->outb.p;
    ea.blocks_cap = b->fin_cap;
    ea.offs2 = (const u64*)b->offs2.p;
    ea.fin_cnt = (const uint32_t*)b->fin_cnt.p;
    ea.n_items = 2 * (u64)n;
    ea.read_base = b->read_base;
    ea.read_ids = b->d_ids;
    ea.sai = ix->d_sai[0];
    ea.rsai = ix->d_sai[1];
    ea.n_sai = ix->n_sai;
    ea.read_len = ix->d_read_len;
    ea.name_rank = ix->d_name_rank;
    ea.edge_offs = (const u64*)b->edge_offs.p;
    ea.edges = (sigax_edge*)b->edges.p;
    ea.edge_cap = b->edge_cap;
    launch_scan((const uint32_t*)b->edge_cnt.p, 2 * (u64)n, (u64*)b->partial.p, (u64*)b->edge_offs.p, dstat + DS_TOTAL_EDGES, ts);
    launch_edges_fill(ea, ts);
  }
  HIP_TRY(hipEventRecord(b->ev[EV_EDGES], ts));
  HIP_TRY(hipStreamWaitEvent(st, b->ev[EV_EDGES], 0));
  HIP_TRY(hipGetLastError());
  return SIGAX_OK;
}

extern "C" int sigax_batch_run(sigax_batch* b, uint32_t read_base, uint32_t min_overlap, uint32_t flags, void* stream) {
  if (!b) return fail(SIGAX_E_ARG, "NULL batch");
  if (b->n_reads && !b->d_seqs) return fail(SIGAX_E_STATE, "no reads set on this batch");
  HIP_TRY(hipSetDevice(b->ix->device));
  b->read_base = read_base;
  b->minov = min_overlap;
  b->flags = flags;
  if (flags & SIGAX_DUPLICATE) {  // duplicate(): minOverlap = seq.length() so no proper-overlap block is ever pushed;
    b->minov = 0xFFFFFFFFu;       // the exhaustive plumbing then returns exactly {containment of find 0, of find 3}
    b->flags = (flags & SIGAX_EDGES) | SIGAX_DUPLICATE;
  }
  b->finished = false;
  int rc = enqueue(b, (hipStream_t)stream);
  b->ran = rc == SIGAX_OK;
  return rc;
}

extern "C" int sigax_batch_finish(sigax_batch* b, void* stream, sigax_stats* stats) {
  if (!b) return fail(SIGAX_E_ARG, "NULL batch");
  if (!b->ran) return fail(SIGAX_E_STATE, "sigax_batch_run was not called");
  HIP_TRY(hipSetDevice(b->ix->device));
  hipStream_t st = (hipStream_t)stream;
  for (int attempt = 0; attempt < 8; ++attempt) {
    u64 ds[DS_COUNT];
    HIP_TRY(hipMemcpyAsync(ds, b->dstat.p, sizeof(ds), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    bool again = false;
    {
      // what the longest chain needed becomes the index's knowledge (later runs and other batch objects size by it)
      uint32_t seen = b->ix->cap_seen->load();
      const uint32_t got = (uint32_t)std::min<u64>(ds[DS_MAX_CHAIN], 0x0FFFFFFFull);
      while (got > seen && !b->ix->cap_seen->compare_exchange_weak(seen, got)) {}
    }
    if (ds[DS_BAD_IDS])
      return fail(SIGAX_E_ARG, "%llu (read, side) items carry a read id beyond the %llu indexed reads (sigax_batch_set_device_read_ids)",
                  (unsigned long long)ds[DS_BAD_IDS], (unsigned long long)b->ix->n_strings);
    if (ds[DS_FIND_OVERFLOW]) {
      if (b->cap >= worst_cap(b->cur_max_len, b->minov))
        return fail(SIGAX_E_CAPACITY, "candidate arena overflow (max read length given too small?)");
      b->cap_floor = (uint32_t)std::min<u64>(ds[DS_MAX_CHAIN], 0x0FFFFFFFull);  // the chains ran out of slots: once more with what they need
      again = true;
    }
    if (ds[DS_POOL_OVERFLOW]) {
      if (b->pool_cap > (1u << 22)) return fail(SIGAX_E_CAPACITY, "%llu reads overflow the filter/extract pool", ds[DS_POOL_OVERFLOW]);
      b->pool_cap *= 4;
      again = true;
    }
    if (ds[DS_FIN_TOP] > b->fin_cap) {
      b->fin_cap = ds[DS_FIN_TOP] + ds[DS_FIN_TOP] / 4 + 1024;
      b->fin_grown = true;
      if (b->edge_cap) b->edge_cap = std::max<u64>(b->edge_cap, b->fin_cap * 2);
      again = true;
    }
    if (!again && (b->flags & SIGAX_EDGES) && ds[DS_TOTAL_EDGES] > b->edge_cap) {
      b->edge_cap = ds[DS_TOTAL_EDGES] + ds[DS_TOTAL_EDGES] / 4 + 1024;
      again = true;
    }
    if (again) {
      ++b->n_reruns;
      int rc = enqueue(b, st);
      if (rc != SIGAX_OK) return rc;
      continue;
    }
    memset(&b->last, 0, sizeof(b->last));
    b->last.n_reads = b->n_reads;
    b->last.n_candidate_blocks = ds[DS_CAND_BLOCKS];
    b->last.n_blocks = ds[DS_TOTAL_BLOCKS];
    b->last.n_edges = (b->flags & SIGAX_EDGES) ? ds[DS_TOTAL_EDGES] : 0;
    b->last.n_occ_find = ds[DS_OCC_FIND];
    b->last.n_occ_extract = ds[DS_OCC_EXTRACT];
    b->last.n_substring = ds[DS_SUBSTRING];
    b->last.n_slow_reads = 0;
    for (int i = 0; i < SIGAX_MAX_SUB; ++i) b->last.n_slow_reads += ds[DS_SLOW_BASE + i];
    {
      // Reads with sequencing errors make the extraction branch; the strict lean launch hands such items on after wasted
      // work.  If more than a tenth of the (read, side) items went on, this batch object starts its next 32 runs with the
      // branching lean launch (9 % slower on items that do not branch, no wasted work on those that do: measured equal
      // at 8 % handed on, 12 % ahead at 23 %, tools/ab_skip_strict.sh), then tries the strict one again.
      u64 q64 = 0;
      for (int i = 0; i < SIGAX_MAX_SUB; ++i) q64 += ds[DS_W64_BASE + i];
      for (int k = 0; k < 4; ++k) {
        const int base = k == 0 ? DS_W64_BASE : k == 1 ? DS_W64B_BASE : k == 2 ? DS_W64C_BASE : DS_W64D_BASE;
        u64 m = 0;
        for (int i = 0; i < SIGAX_MAX_SUB; ++i) m = std::max<u64>(m, ds[base + i]);
        b->qhint[k] = m;
      }
      b->qhint_lean_off = b->lean_off;
      const u64 items = 2ull * b->n_reads;
      if (!b->lean_off) {
        if (q64 * 10 > items) {
          b->lean_off = true;
          b->lean_off_runs = 32;
        }
      } else if (--b->lean_off_runs == 0) {
        b->lean_off = false;
      }
    }
    {
      // Which kernel did the step wait for?  The launches' own durations (beside each other, as they ran) decide the
      // residency split of this batch object's next run: the finder's three workgroups per CU leave two and a half
      // filter/extract workgroups room, which is the better split while the two kernels take about the same time; reads with
      // sequencing errors make filter/extract the longer one by up to 2x, and it then gets three per CU (enqueue).  With
      // hysteresis: the ratio itself moves by 0.1 with the split.
      float tf = 0.f, tx = 0.f, t = 0.f;
      bool ok = true;
      for (unsigned i = 0; i < b->nsub && ok; ++i) {
        ok = hipEventElapsedTime(&t, b->sev[i][SV_F0], b->sev[i][SV_F1]) == hipSuccess;
        tf += t;
        ok = ok && hipEventElapsedTime(&t, b->sev[i][SV_X0], b->sev[i][SV_X1]) == hipSuccess;
        tx += t;
      }
      if (!ok) (void)hipGetLastError();
      else if (tf > 0.f) {
        if (!b->fx_heavy && tx > 1.12f * tf) b->fx_heavy = true;
        else if (b->fx_heavy && tx < 0.95f * tf) b->fx_heavy = false;
      }
      if (b->coop_tune < 4 && coop_tunable(b->ix)) {  // want_coop: this run was per lane (0, 1) or cooperative (2, 3)
        if (!ok || tf <= 0.f) {
          b->coop_tune = 4;  // no times, no tuning: per lane
          b->coop_pick = false;
        } else {
          if (b->coop_tune == 1) b->coop_t_lane = tf;
          if (b->coop_tune == 3) b->coop_pick = b->last_coop && tf < b->coop_t_lane;
          ++b->coop_tune;
        }
      }
    }
    b->last.n_extract_errors = ds[DS_EXTRACT_ERRORS];
    b->last.n_sectors_find = ds[DS_SEC_FIND];
    b->last.n_sectors_extract = ds[DS_SEC_EXTRACT];
    b->last_total_blocks = ds[DS_TOTAL_BLOCKS];
    b->last_total_edges = b->last.n_edges;
    b->finished = true;
    if (stats) *stats = b->last;
    return SIGAX_OK;
  }
  return fail(SIGAX_E_CAPACITY, "arenas still overflowing after 8 attempts");
}

extern "C" int sigax_batch_device_outputs(sigax_batch* b, const sigax_block** d_blocks, const uint64_t** d_block_offs,
                                          const uint8_t** d_substring, const sigax_edge** d_edges) {
  if (!b || !b->finished) return fail(SIGAX_E_STATE, "batch not finished");
  if (d_blocks) *d_blocks = (const sigax_block*)b->outb.p;
  if (d_block_offs) *d_block_offs = (const uint64_t*)b->block_offs.p;
  if (d_substring) *d_substring = (const uint8_t*)b->substring.p;
  if (d_edges) *d_edges = (const sigax_edge*)b->edges.p;
  return SIGAX_OK;
}

extern "C" void sigax_result_free(sigax_result* r) {
  if (!r) return;
  free(r->block_offs);
  free(r->blocks);
  free(r->substring);
  free(r->edges);
  memset(r, 0, sizeof(*r));
}

extern "C" int sigax_batch_download(sigax_batch* b, sigax_result* out) {
  if (!b || !out) return fail(SIGAX_E_ARG, "NULL argument");
  if (!b->finished) return fail(SIGAX_E_STATE, "batch not finished");
  HIP_TRY(hipSetDevice(b->ix->device));
  memset(out, 0, sizeof(*out));
  uint32_t n = b->n_reads;
  out->n_reads = n;
  out->stats = b->last;
  out->block_offs = (uint64_t*)malloc(((size_t)n + 1) * 8);
  out->blocks = (sigax_block*)malloc(std::max<size_t>(1, b->last_total_blocks) * sizeof(sigax_block));
  out->substring = (uint8_t*)malloc(std::max<size_t>(1, n));
  out->n_edges = b->last_total_edges;
  out->edges = (sigax_edge*)malloc(std::max<size_t>(1, b->last_total_edges) * sizeof(sigax_edge));
  if (!out->block_offs || !out->blocks || !out->substring || !out->edges) {
    sigax_result_free(out);
    return fail(SIGAX_E_ARG, "host allocation failed");
  }
  HIP_TRY(hipMemcpy(out->block_offs, b->block_offs.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
  if (b->last_total_blocks)
    HIP_TRY(hipMemcpy(out->blocks, b->outb.p, b->last_total_blocks * sizeof(sigax_block), hipMemcpyDeviceToHost));
  if (n) HIP_TRY(hipMemcpy(out->substring, b->substring.p, n, hipMemcpyDeviceToHost));
  if (b->last_total_edges)
    HIP_TRY(hipMemcpy(out->edges, b->edges.p, b->last_total_edges * sizeof(sigax_edge), hipMemcpyDeviceToHost));
  return SIGAX_OK;
}

// Diagnostic builds (-DSIGAX_FX_PROFILE): which path the extension rounds of the last finished run took.  Not in the header.
extern "C" int sigax_debug_counters(sigax_batch* b, uint64_t out[32]) {
  if (!b || !out) return fail(SIGAX_E_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(b->ix->device));
  HIP_TRY(hipMemcpy(out, (u64*)b->dstat.p + DS_PROF_BASE, 32 * 8, hipMemcpyDeviceToHost));
  return SIGAX_OK;
}

// What the ASQG writer needs from a finished batch: substring flags and edge records (not the 80-byte blocks).
extern "C" int sigax_batch_download_edges(sigax_batch* b, uint8_t* substring, sigax_edge** edges, uint64_t* n_edges) {
  if (!b || !edges || !n_edges) return fail(SIGAX_E_ARG, "NULL argument");
  if (!b->finished) return fail(SIGAX_E_STATE, "batch not finished");
  HIP_TRY(hipSetDevice(b->ix->device));
  *edges = nullptr;
  *n_edges = b->last_total_edges;
  if (substring && b->n_reads) HIP_TRY(hipMemcpy(substring, b->substring.p, b->n_reads, hipMemcpyDeviceToHost));
  sigax_edge* e = (sigax_edge*)malloc(std::max<size_t>(1, b->last_total_edges) * sizeof(sigax_edge));
  if (!e) return fail(SIGAX_E_ARG, "host allocation failed");
  if (b->last_total_edges) {
    hipError_t err = hipMemcpy(e, b->edges.p, b->last_total_edges * sizeof(sigax_edge), hipMemcpyDeviceToHost);
    if (err != hipSuccess) {
      free(e);
      return fail(SIGAX_E_DEVICE, "copying edge records: %s", hipGetErrorString(err));
    }
  }
  *edges = e;
  return SIGAX_OK;
}

// How many reads of up to max_read_len bases one batch object may hold when `in_flight` of them share the device's
// free memory (the candidate arena is sized for the worst case: 4 chains x (L - m + 1) records per read).
extern "C" int sigax_batch_size_hint(sigax_index* ix, uint32_t max_read_len, uint32_t min_overlap, uint32_t flags, uint32_t in_flight,
                                     uint32_t* max_reads) {
  if (!ix || !max_reads) return fail(SIGAX_E_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ix->device));
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  free_b = free_b > ix->tab_plan ? free_b - ix->tab_plan : 0;  // the row tables are allocated later (build_rowend)
  // Slots per chain: the WORST case (one per overlap length).  A run sizes its arena by the longest chain seen so far and
  // repeats itself with more when a chain outgrows that (sigax_batch_finish) -- up to the worst case, so that is what a
  // batch of this many reads must be able to get: a hint from the typical size let a rerun's arena grow fail with
  // SIGAX_E_DEVICE in the middle of a job on a memory-tight index (BASELINE configs[4]: 190 GB of tables).  The price is
  // nil where memory is plentiful (the callers cap their batches at 2^20 reads) and 1 M -> 0.7 M reads per batch at configs[4].
  const uint32_t mo = (flags & SIGAX_DUPLICATE) ? max_read_len : min_overlap;
  const u64 cap = worst_cap(max_read_len, mo);
  if (!ix->ptab_tried) free_b = free_b > prefix_table_bytes(ix->wide, 13) ? free_b - prefix_table_bytes(ix->wide, 13) : 0;  // `siga correct`'s table, built at its first call
  // + the locality order's keys, values and sort space (40 bytes per read), the per-read queues and counters (~70)
  const u64 per_read = 4 * cap * cand_bytes(ix->wide) + max_read_len + ((flags & SIGAX_IRREDUCIBLE) ? 8 : 64) * (2 * 80 + 2 * 16 + 12) + 256;
  const u64 fixed = (2ull << 30) + (u64)32768 * (4 * (cap + 2) + 128) * SIGAX_ENT_BYTES;  // pools of the lane-group and general kernels
  const u64 share = (u64)(free_b * 0.85) / std::max<uint32_t>(in_flight, 1u);
  u64 n = share > fixed ? (share - fixed) / per_read : 0;
  n = std::min<u64>(n, 1u << 22);
  if (n < 1024) return fail(SIGAX_E_CAPACITY, "not enough free device memory for a batch of reads of %u bases (%zu bytes free)", max_read_len, free_b);
  *max_reads = (uint32_t)n;
  return SIGAX_OK;
}

extern "C" int sigax_batch_set_subbatches(sigax_batch* b, uint32_t n) {
  if (!b) return fail(SIGAX_E_ARG, "NULL batch");
  if (n > SIGAX_MAX_SUB) return fail(SIGAX_E_ARG, "at most %d sub-batches", SIGAX_MAX_SUB);
  b->nsub_req = n;
  return SIGAX_OK;
}

extern "C" int sigax_batch_kernel_ms(sigax_batch* b, float ms[5], uint32_t* n_sub) {
  if (!b || !ms) return fail(SIGAX_E_ARG, "NULL argument");
  if (!b->finished) return fail(SIGAX_E_STATE, "batch not finished");
  if (n_sub) *n_sub = b->nsub * b->find_per_sub;
  for (int i = 0; i < 5; ++i) ms[i] = 0.f;
  for (unsigned i = 0; i < b->nsub; ++i) {  // sums over the sub-batch launches (which overlap across the two streams)
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, b->sev[i][SV_F0], b->sev[i][SV_F1]));
    ms[0] += t;
    HIP_TRY(hipEventElapsedTime(&t, b->sev[i][SV_X0], b->sev[i][SV_X1]));
    ms[1] += t;
    HIP_TRY(hipEventElapsedTime(&t, b->sev[i][SV_X1], b->sev[i][SV_G1]));
    ms[2] += t;
  }
  HIP_TRY(hipEventElapsedTime(&ms[3], b->ev[EV_FX_DONE], b->ev[EV_ORDER]));
  HIP_TRY(hipEventElapsedTime(&ms[4], b->ev[EV_ORDER], b->ev[EV_EDGES]));
  return SIGAX_OK;
}

extern "C" int sigax_batch_run_info(sigax_batch* b, sigax_run_info* out) {
  if (!b || !out) return fail(SIGAX_E_ARG, "NULL argument");
  if (!b->finished) return fail(SIGAX_E_STATE, "batch not finished");
  HIP_TRY(hipSetDevice(b->ix->device));
  memset(out, 0, sizeof(*out));
  out->n_sub = b->nsub;
  out->find_per_sub = b->find_per_sub;
  out->two_step = b->last_two_step ? 1u : 0u;
  out->coop = b->last_coop ? 1u : 0u;
  out->read_order = b->last_perm ? 1u : 0u;
  out->cap = b->cap;
  out->worst_cap = worst_cap(b->cur_max_len, b->minov);
  const FmStrand& f = b->ix->st[0];
  out->row_bits = f.sa ? f.sa_bits : 0u;
  out->row_syms = f.sa ? (f.sa_bits - f.ld_bits - f.t_bits) / 2u : 0u;
  out->row_text = f.text ? 1u : 0u;
  out->row_direct = f.xmap ? 1u : 0u;
  out->deep_k = b->last_deep_k;
  out->arena_bytes = b->arena.bytes;
  DevBuf* all[] = {&b->ids_own, &b->seqs_own, &b->offs_own, &b->arena, &b->chain_cnt, &b->pool, &b->wpool, &b->work, &b->work64, &b->work64b, &b->work64c, &b->work64d, &b->perm,
                   &b->ord_keys, &b->ord_tmp, &b->occ_side, &b->slow_flag, &b->offs2, &b->item_base, &b->fin, &b->fin_cnt, &b->substring, &b->block_offs,
                   &b->outb, &b->edge_cnt, &b->edge_offs, &b->edges, &b->partial, &b->dstat};
  for (DevBuf* d : all) out->workspace_bytes += d->bytes;
  out->reruns = b->n_reruns;
  if (b->last_ordered) HIP_TRY(hipEventElapsedTime(&out->order_ms, b->ev[EV_ORD0], b->ev[EV_ORD1]));
  return SIGAX_OK;
}

// One call = OverlapBuilder::overlap for any number of reads: what does not fit one device workspace beside the index goes
// through it in pieces (sized by sigax_batch_size_hint) and the pieces' results are joined in read order.
extern "C" int sigax_overlap_batch(sigax_index* ix, const char* seqs, const uint64_t* offs, uint32_t n_reads,
                                   uint32_t read_base, uint32_t min_overlap, uint32_t flags, sigax_result* out) {
  if (!ix || !out || (n_reads && (!seqs || !offs))) return fail(SIGAX_E_ARG, "NULL argument");
  memset(out, 0, sizeof(*out));
  uint32_t max_len = 0;
  for (uint32_t i = 0; i < n_reads; ++i) {
    if (offs[i + 1] < offs[i] || offs[i + 1] - offs[i] > 0x0FFFFFFFull) return fail(SIGAX_E_ARG, "bad offsets at read %u", i);
    max_len = std::max<uint32_t>(max_len, (uint32_t)(offs[i + 1] - offs[i]));
  }
  uint32_t piece = n_reads;
  if (n_reads > 65536) {
    uint32_t hint = 0;
    int rc = sigax_batch_size_hint(ix, max_len, min_overlap, flags, 1, &hint);
    if (rc != SIGAX_OK) return rc;
    static const char* env = getenv("SIGAX_TEST_PIECE");  // tests: force several pieces on small inputs
    if (env) hint = std::max<uint32_t>(1u, (uint32_t)atoi(env));
    piece = std::min(n_reads, hint);
  } else if (const char* env = getenv("SIGAX_TEST_PIECE")) {
    piece = std::min<uint32_t>(n_reads, std::max<uint32_t>(1u, (uint32_t)atoi(env)));
  }
  sigax_batch* b = nullptr;
  int rc = sigax_batch_create(ix, piece, 0, max_len, &b);
  if (rc != SIGAX_OK) return rc;
  if (piece == n_reads) {
    rc = sigax_batch_upload(b, seqs, offs, n_reads, nullptr);
    if (rc == SIGAX_OK) rc = sigax_batch_run(b, read_base, min_overlap, flags, nullptr);
    if (rc == SIGAX_OK) rc = sigax_batch_finish(b, nullptr, nullptr);
    if (rc == SIGAX_OK) rc = sigax_batch_download(b, out);
    sigax_batch_destroy(b);
    return rc;
  }
  out->n_reads = n_reads;
  out->block_offs = (uint64_t*)malloc(((size_t)n_reads + 1) * 8);
  out->substring = (uint8_t*)malloc(std::max<size_t>(1, n_reads));
  size_t blk_cap = 0, edge_cap = 0;
  u64 nblk = 0, nedge = 0;
  if (!out->block_offs || !out->substring) rc = fail(SIGAX_E_ARG, "host allocation failed");
  std::vector<uint64_t> po;
  for (uint32_t lo = 0; rc == SIGAX_OK && lo < n_reads; lo += piece) {
    const uint32_t n = std::min(piece, n_reads - lo);
    po.resize((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) po[i] = offs[lo + i] - offs[lo];
    sigax_result part;
    memset(&part, 0, sizeof(part));
    rc = sigax_batch_upload(b, seqs + offs[lo], po.data(), n, nullptr);
    if (rc == SIGAX_OK) rc = sigax_batch_run(b, read_base + lo, min_overlap, flags, nullptr);
    if (rc == SIGAX_OK) rc = sigax_batch_finish(b, nullptr, nullptr);
    if (rc == SIGAX_OK) rc = sigax_batch_download(b, &part);
    if (rc != SIGAX_OK) break;
    const u64 pb = part.block_offs[n], pe = part.n_edges;
    if (nblk + pb > blk_cap) {
      blk_cap = std::max<size_t>((size_t)(nblk + pb), blk_cap + blk_cap / 2);
      void* q = realloc(out->blocks, std::max<size_t>(1, blk_cap) * sizeof(sigax_block));
      if (!q) rc = fail(SIGAX_E_ARG, "host allocation failed");
      else out->blocks = (sigax_block*)q;
    }
    if (rc == SIGAX_OK && nedge + pe > edge_cap) {
      edge_cap = std::max<size_t>((size_t)(nedge + pe), edge_cap + edge_cap / 2);
      void* q = realloc(out->edges, std::max<size_t>(1, edge_cap) * sizeof(sigax_edge));
      if (!q) rc = fail(SIGAX_E_ARG, "host allocation failed");
      else out->edges = (sigax_edge*)q;
    }
    if (rc == SIGAX_OK) {
      for (uint32_t i = 0; i < n; ++i) out->block_offs[lo + i] = nblk + part.block_offs[i];
      if (pb) memcpy(out->blocks + nblk, part.blocks, pb * sizeof(sigax_block));
      if (pe) memcpy(out->edges + nedge, part.edges, pe * sizeof(sigax_edge));
      memcpy(out->substring + lo, part.substring, n);
      nblk += pb;
      nedge += pe;
      uint64_t* acc = (uint64_t*)&out->stats;
      const uint64_t* add = (const uint64_t*)&part.stats;
      for (size_t k = 0; k < sizeof(sigax_stats) / 8; ++k) acc[k] += add[k];
    }
    sigax_result_free(&part);
  }
  sigax_batch_destroy(b);
  if (rc != SIGAX_OK) {
    sigax_result_free(out);
    return rc;
  }
  out->block_offs[n_reads] = nblk;
  out->n_edges = nedge;
  if (!out->blocks) out->blocks = (sigax_block*)malloc(sizeof(sigax_block));
  if (!out->edges) out->edges = (sigax_edge*)malloc(sizeof(sigax_edge));
  return SIGAX_OK;
}
