// siga_amd/host/siga_host.cpp -- see siga_host.hpp.  Host C++ above the C-ABI; all overlap compute happens in
// libsigax.so on the GPU.
#include "siga_host.hpp"

#include <zlib.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <type_traits>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string_view>
#include <thread>

#include "line_deflate.hpp"
#include "sais.hpp"

namespace sigah {

// ------------------------------------------------------------------------------------------------------
// bzip2 input (Utils::ifstream, src/utils.cpp:50-126: ".bz2" goes through a bzip2 filter).  The image has libbz2's shared
// library but not its header: the three entry points of its streaming interface are bound at run time (the bz_stream layout
// is libbz2's documented public one); without the library a .bz2 input fails to open.
// ------------------------------------------------------------------------------------------------------
namespace {
struct BzStream {
  char* next_in;
  unsigned int avail_in, total_in_lo32, total_in_hi32;
  char* next_out;
  unsigned int avail_out, total_out_lo32, total_out_hi32;
  void* state;
  void* (*bzalloc)(void*, int, int);
  void (*bzfree)(void*, void*);
  void* opaque;
};
struct Bz2Lib {
  int (*init)(BzStream*, int, int) = nullptr;
  int (*run)(BzStream*) = nullptr;
  int (*end)(BzStream*) = nullptr;
  Bz2Lib() {
    void* h = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!h) h = dlopen("libbz2.so.1", RTLD_NOW);
    if (!h) return;
    init = (int (*)(BzStream*, int, int))dlsym(h, "BZ2_bzDecompressInit");
    run = (int (*)(BzStream*))dlsym(h, "BZ2_bzDecompress");
    end = (int (*)(BzStream*))dlsym(h, "BZ2_bzDecompressEnd");
    if (!init || !run || !end) init = nullptr;
  }
};
}  // namespace
static bool is_bz2(const unsigned char* magic, ssize_t got) { return got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h'; }
// the whole of a (possibly multi-stream) bzip2 file, decompressed; false on a corrupt or truncated stream
static bool bz2_expand(const std::vector<char>& in, std::vector<char>* out) {
  static const Bz2Lib lib;
  if (!lib.init) return false;
  out->resize(std::max<size_t>(in.size() * 5, 1 << 20));
  size_t len = 0, pos = 0;
  while (pos < in.size()) {
    BzStream z;
    memset(&z, 0, sizeof(z));
    if (lib.init(&z, 0, 0) != 0) return false;
    int rc = 0;
    while (rc == 0) {
      if (out->size() - len < (1u << 20)) out->resize(out->size() * 2);
      z.next_in = const_cast<char*>(in.data()) + pos;
      z.avail_in = (unsigned)std::min<size_t>(in.size() - pos, 1u << 30);
      z.next_out = out->data() + len;
      z.avail_out = (unsigned)std::min<size_t>(out->size() - len, 1u << 30);
      const unsigned in0 = z.avail_in, out0 = z.avail_out;
      rc = lib.run(&z);
      pos += in0 - z.avail_in;
      len += out0 - z.avail_out;
      if (rc == 0 && in0 == z.avail_in && out0 == z.avail_out) rc = -1;  // no progress: truncated
    }
    lib.end(&z);
    if (rc != 4) return false;  // BZ_STREAM_END
  }
  out->resize(len);
  return true;
}

// ------------------------------------------------------------------------------------------------------
// line source / sequence readers
// ------------------------------------------------------------------------------------------------------
class LineSource {
 public:
  explicit LineSource(const std::string& path) : _f(nullptr), _pos(0), _len(0), _eof(false), _err(false), _mpos(0), _ismem(false) {
    unsigned char magic[3] = {0, 0, 0};
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return;
    const ssize_t got = pread(fd, magic, 3, 0);
    if (is_bz2(magic, got)) {  // expanded in memory, then served like a file
      std::vector<char> raw;
      struct stat st;
      bool ok = fstat(fd, &st) == 0;
      if (ok) {
        raw.resize((size_t)st.st_size);
        size_t n = 0;
        while (n < raw.size()) {
          ssize_t k = read(fd, raw.data() + n, raw.size() - n);
          if (k <= 0) break;
          n += (size_t)k;
        }
        ok = n == raw.size() && bz2_expand(raw, &_mem);
      }
      close(fd);
      _ismem = ok;
      return;
    }
    close(fd);
    _f = gzopen(path.c_str(), "rb");
  }
  ~LineSource() {
    if (_f) gzclose(_f);
  }
  bool ok() const { return _f != nullptr || _ismem; }
  bool failed() const { return _err; }  // a read error (corrupt .gz), as opposed to the end of the file
  int peek() {
    if (_pos >= _len && !fill()) return -1;
    return (unsigned char)_buf[_pos];
  }
  // std::getline semantics: false only when nothing at all could be read
  bool getline(std::string& line) {
    line.clear();
    bool any = false;
    while (true) {
      if (_pos >= _len && !fill()) return any;
      any = true;
      const char* b = _buf + _pos;
      const char* nl = (const char*)memchr(b, '\n', _len - _pos);
      if (nl) {
        line.append(b, nl - b);
        _pos += (nl - b) + 1;
        return true;
      }
      line.append(b, _len - _pos);
      _pos = _len;
    }
  }
  void rewind() {
    if (_f) gzrewind(_f);
    _mpos = 0;
    _pos = _len = 0;
    _eof = false;
  }

 private:
  bool fill() {
    if (_eof) return false;
    if (_ismem) {
      const size_t n = std::min(sizeof(_buf), _mem.size() - _mpos);
      if (n == 0) {
        _eof = true;
        return false;
      }
      memcpy(_buf, _mem.data() + _mpos, n);
      _mpos += n;
      _pos = 0;
      _len = n;
      return true;
    }
    int n = gzread(_f, _buf, sizeof(_buf));
    if (n <= 0) {
      if (n < 0) _err = true;  // a corrupt or truncated .gz is not the end of the reads
      _eof = true;
      return false;
    }
    _pos = 0;
    _len = (size_t)n;
    return true;
  }
  gzFile _f;
  char _buf[1 << 16];
  size_t _pos, _len;
  bool _eof, _err;
  std::vector<char> _mem;  // a .bz2 input, expanded
  size_t _mpos;
  bool _ismem;
};

// ------------------------------------------------------------------------------------------------------
// host parallelism: the GPU replaces the reference's OpenMP loop over reads; what is left on the host (parsing, name
// ranks, text formatting, deflate) is spread over plain threads
// ------------------------------------------------------------------------------------------------------
static unsigned host_threads(size_t requested) {
  const char* env = getenv("SIGA_HOST_THREADS");
  if (env && atoi(env) > 0) return (unsigned)atoi(env);
  unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  unsigned autoN = std::min(hw, 32u);
  return (unsigned)std::max<size_t>(requested > 1 ? requested : 0, autoN);
}

template <class F>
static void parallel_for(size_t n, unsigned nt, F f) {
  if (n == 0) return;
  nt = (unsigned)std::min<size_t>(std::max(1u, nt), n);
  if (nt == 1) {
    for (size_t i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<size_t> next(0);
  auto work = [&] {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= n) break;
      f(i);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
}

static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

// ------------------------------------------------------------------------------------------------------
// ReadStore: a whole reads file parsed at once with the reference's reader semantics (src/kseq.cpp:140-228), in
// parallel for FASTA; names and comments are spans of the file image, sequences are packed the way the device batches
// take them
// ------------------------------------------------------------------------------------------------------
// The bytes of a reads file: a plain file is mapped (its pages come in under the parsing threads: reading 3.4 GB --
// BASELINE configs[2]'s reads as FASTA -- into a vector took one thread 1.3 s before the first chunk could be parsed),
// a compressed one is expanded into memory.
struct FileImage {
  std::vector<char> owned;
  const char* map = nullptr;
  size_t map_size = 0;
  FileImage() = default;
  FileImage(const FileImage&) = delete;
  FileImage& operator=(const FileImage&) = delete;
  ~FileImage() {
    if (map) munmap((void*)map, map_size);
  }
  const char* data() const { return map ? map : owned.data(); }
  size_t size() const { return map ? map_size : owned.size(); }
};

// Memory that is written before it is read: no zero fill by us (std::vector::resize ran 3 GB of it on one thread), and for
// large blocks 2 MB pages (anonymous mapping + MADV_HUGEPAGE; the GPU boxes run transparent huge pages in `madvise` mode):
// the first touch of BASELINE configs[2]'s 3 GB of bases was 790 k page faults on the join's threads, each with its 4 KB
// cleared by the kernel -- most of the 0.31 s the join took.
struct RawBlock {
  void* p = nullptr;
  size_t bytes = 0;
  bool mapped = false;
  RawBlock() = default;
  RawBlock(const RawBlock&) = delete;
  RawBlock& operator=(const RawBlock&) = delete;
  RawBlock(RawBlock&& o) noexcept : p(o.p), bytes(o.bytes), mapped(o.mapped) { o.p = nullptr; o.bytes = 0; o.mapped = false; }
  RawBlock& operator=(RawBlock&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p; bytes = o.bytes; mapped = o.mapped;
      o.p = nullptr; o.bytes = 0; o.mapped = false;
    }
    return *this;
  }
  ~RawBlock() { release(); }
  void release() {
    if (p) {
      if (mapped) munmap(p, bytes);
      else ::operator delete(p);
    }
    p = nullptr;
    bytes = 0;
    mapped = false;
  }
  void alloc(size_t want) {  // contents undefined
    release();
    if (want == 0) want = 1;
    static const bool no_huge = getenv("SIGA_NO_HUGEPAGES") != nullptr;  // A/B aid
    if (want >= ((size_t)8 << 20) && !no_huge) {
      const size_t two = (size_t)2 << 20;
      const size_t len = (want + two - 1) & ~(two - 1);
      void* q = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (q != MAP_FAILED) {
#ifdef MADV_HUGEPAGE
        (void)madvise(q, len, MADV_HUGEPAGE);
#endif
        p = q;
        bytes = len;
        mapped = true;
        return;
      }
    }
    p = ::operator new(want);
    bytes = want;
    mapped = false;
  }
};

struct RawChars {
  RawBlock b;
  size_t n = 0;
  void resize(size_t k) {  // contents undefined
    b.alloc(k);
    n = k;
  }
  char* data() { return (char*)b.p; }
  const char* data() const { return (const char*)b.p; }
  size_t size() const { return n; }
};

// ... and the per-read tables likewise (every entry is written by the loader's join, on its threads)
template <class T>
struct RawVec {
  static_assert(std::is_trivial<T>::value, "RawVec holds plain values");
  RawBlock b;
  size_t n = 0;
  void resize(size_t k) {  // contents undefined
    b.alloc(k * sizeof(T));
    n = k;
  }
  T* data() { return (T*)b.p; }
  const T* data() const { return (const T*)b.p; }
  size_t size() const { return n; }
  T& operator[](size_t i) { return ((T*)b.p)[i]; }
  const T& operator[](size_t i) const { return ((const T*)b.p)[i]; }
  const T* begin() const { return (const T*)b.p; }
  const T* end() const { return (const T*)b.p + n; }
};

struct ReadStore {
  FileImage file;
  RawChars seqs;
  RawVec<uint64_t> offs;                 // n + 1
  RawVec<uint64_t> head_off;             // raw header (after '>' / '@'), a span of `file`
  RawVec<uint32_t> head_len, name_len;   // name = head[0, name_len); comment = head[name_len + 1, head_len)
  RawVec<uint64_t> qual_off;             // FASTQ: span of `file`, seq length long
  bool fastq = false;
  size_t size() const { return head_off.size(); }
  std::string_view name(size_t i) const { return std::string_view(file.data() + head_off[i], name_len[i]); }
  std::string_view comment(size_t i) const {
    return name_len[i] < head_len[i] ? std::string_view(file.data() + head_off[i] + name_len[i] + 1, head_len[i] - name_len[i] - 1)
                                     : std::string_view();
  }
  std::string_view seq(size_t i) const { return std::string_view(seqs.data() + offs[i], offs[i + 1] - offs[i]); }
  std::string_view quality(size_t i) const {
    return fastq ? std::string_view(file.data() + qual_off[i], offs[i + 1] - offs[i]) : std::string_view();
  }
};

static bool slurp(const std::string& path, FileImage* img) {
  std::vector<char>* out = &img->owned;
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  unsigned char magic[3] = {0, 0, 0};
  ssize_t got = pread(fd, magic, 3, 0);
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return false;
  }
  if (is_bz2(magic, got)) {  // bzip2 (Utils::ifstream, src/utils.cpp:91-126)
    std::vector<char> raw((size_t)st.st_size);
    size_t n = 0;
    while (n < raw.size()) {
      ssize_t k = read(fd, raw.data() + n, raw.size() - n);
      if (k <= 0) break;
      n += (size_t)k;
    }
    close(fd);
    return n == raw.size() && bz2_expand(raw, out);
  }
  if (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {  // gzip (Utils::ifstream, src/utils.cpp:50-90)
    close(fd);
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    out->resize(std::max<size_t>((size_t)st.st_size * 4, 1 << 20));
    size_t len = 0;
    for (;;) {
      if (out->size() - len < (1u << 20)) out->resize(out->size() * 2);
      int n = gzread(f, out->data() + len, (unsigned)std::min<size_t>(out->size() - len, 1u << 30));
      if (n < 0) {  // a corrupt or truncated .gz must not be taken for a shorter read set
        gzclose(f);
        return false;
      }
      if (n == 0) break;
      len += (size_t)n;
    }
    const bool whole = gzclose(f) == Z_OK;  // Z_BUF_ERROR: the stream ended inside a member
    out->resize(len);
    return whole;
  }
  if (S_ISREG(st.st_mode) && st.st_size >= (1 << 20)) {
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
      (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
      img->map = (const char*)m;
      img->map_size = (size_t)st.st_size;
      close(fd);
      return true;
    }
  }
  out->resize((size_t)st.st_size);
  size_t len = 0;
  while (len < out->size()) {
    ssize_t n = read(fd, out->data() + len, out->size() - len);
    if (n <= 0) break;
    len += (size_t)n;
  }
  close(fd);
  out->resize(len);
  return true;
}

namespace {
struct ChunkOut {
  std::vector<uint64_t> head_off, seq_len, qual_off;
  std::vector<uint64_t> seq_off;  // by_ref: where the record's one sequence line starts in the file image
  std::vector<uint32_t> head_len;
  std::vector<char> seqs;
  bool by_ref = false;    // every record of the chunk has its bases on ONE line: they stay in the file image until the join
  size_t n_bases = 0;     // bases of the chunk's records (by_ref: nothing was copied; else seqs.size())
  bool stopped = false;   // the reader returned false inside this chunk: nothing after it is read
  bool nameless = false;  // a header with no text: its sequence lines leak into the next record (serial semantics only)
  bool open_empty = false;  // the chunk ends in a named record without sequence
};
}  // namespace

// FASTAReader::read over [b, e) of the file image (src/kseq.cpp:187-228); the chunk starts at a header line.
// by_ref: the bases are not copied here -- a record's sequence is remembered as a span of the file image, which works as
// long as every record has its bases on one line (reads: always); returns false at the first record that has not, and the
// caller parses the chunk again the copying way.  (BASELINE configs[2]'s 20 M reads: the chunk buffers were 3 GB written,
// read once by the join and unmapped again.)
static bool parse_fasta_chunk(const char* base, size_t b, size_t e, bool last_chunk, ChunkOut* o, bool by_ref) {
  bool have_name = false;
  uint64_t hoff = 0;
  uint32_t hlen = 0;
  o->by_ref = by_ref;
  if (!by_ref) o->seqs.reserve(o->seqs.size() + (e - b));  // a chunk's bases are fewer than its bytes: no regrowth
  size_t seq_start = o->seqs.size();
  uint64_t cur_off = 0, cur_len = 0;  // by_ref: the open record's sequence line
  unsigned cur_lines = 0;
  size_t p = b;
  auto cur_seq = [&]() -> size_t { return by_ref ? (size_t)cur_len : o->seqs.size() - seq_start; };
  auto emit = [&] {
    o->head_off.push_back(hoff);
    o->head_len.push_back(hlen);
    o->seq_len.push_back(cur_seq());
    if (by_ref) {
      o->seq_off.push_back(cur_off);
      o->n_bases += cur_len;
      cur_len = 0;
      cur_lines = 0;
    }
    seq_start = o->seqs.size();
  };
  while (p < e) {
    const char* nl = (const char*)memchr(base + p, '\n', e - p);
    size_t le = nl ? (size_t)(nl - base) : e;
    size_t ls = p;
    p = nl ? le + 1 : e;
    while (ls < le && is_space(base[ls])) ++ls;
    while (le > ls && is_space(base[le - 1])) --le;
    if (ls == le) continue;
    if (base[ls] == '>') {
      const size_t have = cur_seq();
      if (have > 0 && have_name && hlen > 0) {
        emit();
      } else if (have_name && hlen > 0) {  // a named record without sequence: the reader gives up here
        o->stopped = true;
        if (!by_ref) o->n_bases = o->seqs.size();
        return true;
      }
      if (have_name && hlen == 0) o->nameless = true;
      have_name = true;
      hoff = ls + 1;
      hlen = (uint32_t)(le - ls - 1);
    } else if (by_ref) {
      if (cur_lines != 0) return false;  // a second sequence line: not a span of the file
      cur_off = ls;
      cur_len = le - ls;
      cur_lines = 1;
    } else {
      o->seqs.insert(o->seqs.end(), base + ls, base + le);
    }
  }
  const size_t have = cur_seq();
  if (have_name && hlen == 0) o->nameless = true;
  if (have > 0 && have_name && hlen > 0) emit();
  else if (have_name && hlen > 0 && !last_chunk) o->open_empty = true;  // the next header makes the reader give up
  else if (have > 0 && !by_ref) o->seqs.resize(seq_start);
  if (!by_ref) o->n_bases = o->seqs.size();
  return true;
}

// FASTQReader::read (src/kseq.cpp:140-185), serial
static void parse_fastq(const char* base, size_t e, ChunkOut* o) {
  int state = 0;
  uint64_t hoff = 0, soff = 0;
  uint32_t hlen = 0, slen = 0;
  size_t p = 0;
  while (p < e) {
    const char* nl = (const char*)memchr(base + p, '\n', e - p);
    size_t le = nl ? (size_t)(nl - base) : e;
    size_t ls = p;
    p = nl ? le + 1 : e;
    while (ls < le && is_space(base[ls])) ++ls;
    while (le > ls && is_space(base[le - 1])) --le;
    if (ls == le) continue;
    if (state == 0) {
      if (base[ls] != '@') return;
      hoff = ls + 1;
      hlen = (uint32_t)(le - ls - 1);
      state = 1;
    } else if (state == 1) {
      soff = ls;
      slen = (uint32_t)(le - ls);
      state = 2;
    } else if (state == 2) {
      const size_t len = le - ls;
      const bool ends = len >= hlen && memcmp(base + le - hlen, base + hoff, hlen) == 0;
      if (base[ls] == '+' && (len == 1 || ends)) state = 3;
      else return;
    } else {
      if (le - ls != slen) return;
      o->head_off.push_back(hoff);
      o->head_len.push_back(hlen);
      o->seq_len.push_back(slen);
      o->qual_off.push_back(ls);
      o->seqs.insert(o->seqs.end(), base + soff, base + soff + slen);
      state = 0;
    }
  }
}

static bool LoadReads(const std::string& path, ReadStore* rs, unsigned nt) {
  const bool timing = getenv("SIGA_TIMING_LOADER") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    const auto t = std::chrono::steady_clock::now();
    if (timing) fprintf(stderr, "[siga]     loader: %-20s %7.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  if (!slurp(path, &rs->file)) return false;
  lap("file image");
  const char* base = rs->file.data();
  const size_t size = rs->file.size();
  if (size == 0 || (base[0] != '@' && base[0] != '>')) return false;  // DNASeqReaderFactory::create (src/kseq.cpp:127-138)
  rs->fastq = base[0] == '@';
  std::vector<ChunkOut> outs;
  if (rs->fastq) {
    outs.resize(1);
    parse_fastq(base, size, &outs[0]);
    outs[0].n_bases = outs[0].seqs.size();
  } else {
    // chunk starts: the first header line at or after i * size / K
    const size_t K = std::max<size_t>(1, std::min<size_t>((size_t)nt * 4, size >> 16));
    std::vector<size_t> starts(1, 0);
    for (size_t i = 1; i < K; ++i) {
      size_t p = i * (size / K);
      const char* nl = (const char*)memchr(base + p, '\n', size - p);
      if (!nl) break;
      p = (size_t)(nl - base) + 1;
      while (p < size) {  // find a line whose first non-blank character is '>'
        size_t q = p;
        while (q < size && base[q] != '\n' && is_space(base[q])) ++q;
        if (q < size && base[q] == '>') break;
        const char* n2 = (const char*)memchr(base + p, '\n', size - p);
        if (!n2) {
          p = size;
          break;
        }
        p = (size_t)(n2 - base) + 1;
      }
      if (p < size && p > starts.back()) starts.push_back(p);
    }
    outs.resize(starts.size());
    static const bool no_ref = getenv("SIGA_LOADER_COPY") != nullptr;  // A/B aid: every chunk the copying way
    parallel_for(starts.size(), nt, [&](size_t i) {
      const size_t e = i + 1 < starts.size() ? starts[i + 1] : size;
      if (no_ref || !parse_fasta_chunk(base, starts[i], e, i + 1 == starts.size(), &outs[i], true)) {
        outs[i] = ChunkOut();
        parse_fasta_chunk(base, starts[i], e, i + 1 == starts.size(), &outs[i], false);
      }
    });
    bool nameless = false;
    for (auto& o : outs) nameless = nameless || o.nameless;
    if (nameless) {  // state leaks across records: only the serial walk reproduces it
      outs.assign(1, ChunkOut());
      parse_fasta_chunk(base, 0, size, true, &outs[0], false);
    }
  }
  lap("chunks parsed");
  // concatenate up to the point where the serial reader would have given up
  size_t nchunks = 0, n = 0, nb = 0;
  for (; nchunks < outs.size(); ++nchunks) {
    n += outs[nchunks].head_off.size();
    nb += outs[nchunks].n_bases;
    if (outs[nchunks].stopped || outs[nchunks].open_empty) {
      ++nchunks;
      break;
    }
  }
  rs->head_off.resize(n);
  rs->head_len.resize(n);
  rs->name_len.resize(n);
  rs->offs.resize(n + 1);
  rs->seqs.resize(nb);
  if (rs->fastq) rs->qual_off.resize(n);
  std::vector<size_t> rbase(nchunks + 1, 0), bbase(nchunks + 1, 0);
  for (size_t c = 0; c < nchunks; ++c) {
    rbase[c + 1] = rbase[c] + outs[c].head_off.size();
    bbase[c + 1] = bbase[c] + outs[c].n_bases;
  }
  parallel_for(nchunks, nt, [&](size_t c) {
    ChunkOut& o = outs[c];
    uint64_t off = bbase[c];
    for (size_t k = 0; k < o.head_off.size(); ++k) {
      const size_t r = rbase[c] + k;
      rs->head_off[r] = o.head_off[k];
      rs->head_len[r] = o.head_len[k];
      const char* h = base + o.head_off[k];
      uint32_t nl = 0;
      while (nl < o.head_len[k] && h[nl] != ' ' && h[nl] != '\t') ++nl;  // make_seq_name (src/kseq.cpp:71-79)
      rs->name_len[r] = nl;
      rs->offs[r] = off;
      if (o.by_ref) memcpy(rs->seqs.data() + off, base + o.seq_off[k], o.seq_len[k]);  // straight from the file image
      off += o.seq_len[k];
      if (rs->fastq) rs->qual_off[r] = o.qual_off[k];
    }
    if (!o.by_ref && !o.seqs.empty()) memcpy(rs->seqs.data() + bbase[c], o.seqs.data(), o.seqs.size());
    // the chunk's own buffers go back here, on this thread: left to the vector of chunks' destructor they were unmapped
    // one after the other (0.4 s of the 0.97 s BASELINE configs[2]'s reads took to load)
    std::vector<char>().swap(o.seqs);
    std::vector<uint64_t>().swap(o.head_off);
    std::vector<uint64_t>().swap(o.seq_len);
    std::vector<uint64_t>().swap(o.seq_off);
    std::vector<uint64_t>().swap(o.qual_off);
    std::vector<uint32_t>().swap(o.head_len);
  });
  rs->offs[n] = nb;
  lap("joined");
  return true;
}

static void trim(std::string& s) {  // boost::algorithm::trim
  auto sp = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  size_t b = 0, e = s.size();
  while (b < e && sp(s[b])) ++b;
  while (e > b && sp(s[e - 1])) --e;
  if (b > 0 || e < s.size()) s = s.substr(b, e - b);
}

static void make_seq_name(std::string& name, std::string& comment) {  // src/kseq.cpp:71-79
  size_t i = name.find_first_of(" \t");
  if (i != std::string::npos) {
    comment = name.substr(i + 1);
    name.resize(i);
  } else {
    comment.clear();
  }
}

DNASeqReader::DNASeqReader() : _fastq(false) {}
DNASeqReader::~DNASeqReader() {}

DNASeqReader* DNASeqReader::create(const std::string& path) {
  std::unique_ptr<LineSource> src(new LineSource(path));
  if (!src->ok()) return nullptr;
  int c = src->peek();  // src/kseq.cpp:127-138
  if (c != '@' && c != '>') return nullptr;
  DNASeqReader* r = new DNASeqReader();
  r->_fastq = c == '@';
  r->_src = std::move(src);
  return r;
}

bool DNASeqReader::failed() const { return _src->failed(); }

void DNASeqReader::reset() {
  _name.clear();
  _src->rewind();
}

bool DNASeqReader::read(DNASeq& sequence) {
  std::string line;
  if (_fastq) {  // src/kseq.cpp:140-185
    int state = 0;
    while (_src->getline(line)) {
      trim(line);
      if (line.empty()) continue;
      if (state == 0) {
        if (line[0] != '@') return false;
        sequence.name = line.substr(1);
        state = 1;
      } else if (state == 1) {
        sequence.seq = line;
        state = 2;
      } else if (state == 2) {
        const std::string& nm = sequence.name;
        bool ends = line.size() >= nm.size() && line.compare(line.size() - nm.size(), nm.size(), nm) == 0;
        if (line[0] == '+' && (line.length() == 1 || ends)) state = 3;
        else return false;
      } else {
        if (line.length() != sequence.seq.length()) return false;
        sequence.quality = line;
        make_seq_name(sequence.name, sequence.comment);
        return true;
      }
    }
    return false;
  }
  // src/kseq.cpp:187-228
  std::string seq;
  while (_src->getline(line)) {
    trim(line);
    if (line.empty()) continue;
    if (line[0] == '>') {
      if (!seq.empty() && !_name.empty()) {
        sequence.name = _name;
        make_seq_name(sequence.name, sequence.comment);
        sequence.seq.swap(seq);
        sequence.quality.clear();
        _name = line.substr(1);
        return true;
      } else if (!_name.empty()) {
        return false;
      }
      _name = line.substr(1);
    } else {
      seq += line;
    }
  }
  if (!seq.empty() && !_name.empty()) {
    sequence.name = _name;
    make_seq_name(sequence.name, sequence.comment);
    sequence.seq.swap(seq);
    sequence.quality.clear();
    _name.clear();  // the reference's stream is at EOF here and never reads again
    return true;
  }
  return false;
}

bool ReadDNASequences(const std::string& file, DNASeqList& sequences, uint32_t flags) {  // src/kseq.cpp:230-256
  std::unique_ptr<DNASeqReader> reader(DNASeqReader::create(file));
  if (!reader) return false;
  DNASeq seq;
  while (reader->read(seq)) {
    if (!(flags & kSeqWithQuality)) seq.quality.clear();
    if (!(flags & kSeqWithComment)) seq.comment.clear();
    sequences.push_back(seq);
  }
  return !reader->failed();  // a corrupt .gz is an error, not a shorter read set
}

std::string Utils::stem(const std::string& filename) {  // src/utils.cpp:128-135
  auto ends = [](const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
  };
  if (ends(filename, ".gz")) return stem(filename.substr(0, filename.size() - 3));
  if (ends(filename, ".bz2")) return stem(filename.substr(0, filename.size() - 4));
  size_t slash = filename.find_last_of('/');
  std::string base = slash == std::string::npos ? filename : filename.substr(slash + 1);
  if (base == "." || base == "..") return base;
  size_t dot = base.find_last_of('.');
  return dot == std::string::npos ? base : base.substr(0, dot);
}

// ------------------------------------------------------------------------------------------------------
// index construction
// ------------------------------------------------------------------------------------------------------
static inline int torank(char c) {  // src/alphabet.h:19-39
  switch (c) {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 3;
    case 'T': return 4;
    default: return 0;
  }
}

// Multi-threaded suffix sort for many-core hosts: bucket every suffix by its first KP symbols (counting sort over
// base-6 keys), then comparison-sort the buckets in parallel.  The terminator is unique, so memcmp from offset KP
// decides every pair.  Same total order as SA-IS by construction (plain suffix array, end of text smallest).  Returns
// false (caller falls back to SA-IS) when a bucket is so large that long repeats would make it crawl.
//
// OWN_SENTINELS: the order of `siga index -a sais` (SAISBuilder, src/suffix_array_builder.cpp:31-172: suffixes compared as
// strings up to the end of their read, ties by read index) -- every read's own '$', ordered by read index, instead of one
// shared '$' with comparisons running on into the next read.  In the concatenated text that is: compare up to and including
// the first '$', then by position.  Keys stop at the first '$' (what follows it counts as nothing) and are computed per
// position instead of rolled.  Reads must be ACGT-only (the caller checks: the reference compares raw characters in one
// phase and ranks in the other, which agree only on A, C, G, T).
template <typename I, bool OWN_SENTINELS = false>
static bool parallel_suffix_sort(const uint8_t* T, uint64_t n /* incl. terminator */, I* SA, unsigned threads) {
  const int KP = 9;
  uint64_t nb = 1;
  for (int i = 0; i < KP; ++i) nb *= 6;
  auto key_at = [&](uint64_t p) {
    uint64_t k = 0;
    bool ended = false;
    for (int i = 0; i < KP; ++i) {
      const uint64_t c = (p + i < n && !ended) ? T[p + i] : 0;
      k = k * 6 + c;
      if (OWN_SENTINELS && c <= 1) ended = true;
    }
    return k;
  };
  // counting passes: a modest number of threads (each holds a histogram of nb counters), rolling base-6 keys
  const unsigned ct = std::min<unsigned>(threads, 16);
  uint64_t top = 1;
  for (int i = 0; i < KP - 1; ++i) top *= 6;
  std::vector<uint64_t> start(nb + 1, 0);
  std::vector<std::vector<uint32_t>> hist(ct);
  const uint64_t chunk = (n + ct - 1) / ct;
  auto sweep = [&](unsigned t, bool scatter) {
    uint64_t b = t * chunk, e = std::min(n, b + chunk);
    if (b >= e) return;
    uint64_t k = key_at(b);
    for (uint64_t p = b; p < e; ++p) {
      if (scatter) SA[start[k] + hist[t][k]++] = (I)p;
      else ++hist[t][k];
      if (OWN_SENTINELS) k = p + 1 < n ? key_at(p + 1) : 0;
      else k = (k % top) * 6 + (p + KP < n ? T[p + KP] : 0);
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < ct; ++t)
      th.emplace_back([&, t] {
        hist[t].assign(nb, 0);
        sweep(t, false);
      });
    for (auto& x : th) x.join();
  }
  uint64_t acc = 0, biggest = 0;
  for (uint64_t k = 0; k < nb; ++k) {
    start[k] = acc;
    uint64_t c = 0;
    for (unsigned t = 0; t < ct; ++t) {
      uint32_t h = hist[t][k];
      hist[t][k] = (uint32_t)c;  // this thread's offset inside the bucket (< 2^32 checked below)
      c += h;
    }
    if (c > 0xFFFFFFF0ull) return false;
    biggest = std::max(biggest, c);
    acc += c;
  }
  start[nb] = acc;
  if (biggest > (64ull << 20)) return false;
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < ct; ++t) th.emplace_back([&, t] { sweep(t, true); });
    for (auto& x : th) x.join();
  }
  // sort the buckets, largest first, pulled from a shared counter
  std::vector<uint64_t> order;
  order.reserve(1 << 20);
  for (uint64_t k = 0; k < nb; ++k)
    if (start[k + 1] - start[k] > 1) order.push_back(k);
  std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return start[a + 1] - start[a] > start[b + 1] - start[b]; });
  std::atomic<uint64_t> next(0);
  auto less = [&](I a, I b) {
    uint64_t pa = (uint64_t)a + KP, pb = (uint64_t)b + KP;
    if (OWN_SENTINELS) {
      // same key: either both reads ended inside the first KP symbols (equal strings: the earlier read first), or neither
      // did and the strings go on: compare up to and including the next '$' of the one that ends first
      bool ended = false;
      for (int i = 0; i < KP && !ended; ++i) ended = (uint64_t)a + i >= n || T[(uint64_t)a + i] <= 1;
      if (ended) return a < b;
      const uint8_t* ea = (const uint8_t*)memchr(T + pa, 1, n - pa);
      const uint8_t* eb = (const uint8_t*)memchr(T + pb, 1, n - pb);
      const uint64_t la = (ea ? (uint64_t)(ea - (T + pa)) : n - pa - 1) + 1, lb = (eb ? (uint64_t)(eb - (T + pb)) : n - pb - 1) + 1;
      const int c = memcmp(T + pa, T + pb, std::min(la, lb));
      if (c != 0) return c < 0;
      return a < b;  // both end here ('$' against a base differs above): the earlier read first
    }
    if (pa >= n || pb >= n) return a > b;  // the shorter suffix (later start) is smaller
    uint64_t la = n - pa, lb = n - pb;
    int c = memcmp(T + pa, T + pb, std::min(la, lb));
    if (c != 0) return c < 0;
    return la < lb;
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; ++t)
      th.emplace_back([&] {
        while (true) {
          uint64_t i = next.fetch_add(1);
          if (i >= order.size()) break;
          uint64_t k = order[i];
          std::sort(SA + start[k], SA + start[k + 1], less);
        }
      });
    for (auto& x : th) x.join();
  }
  return true;
}

template <typename I>
static bool build_strand(const char* seqs, const uint64_t* offs, uint64_t nReads, bool reverse, StrandIndex* out,
                         unsigned threads, bool own_sentinels = false) {
  uint64_t total = 0;
  for (uint64_t i = 0; i < nReads; ++i) total += (offs[i + 1] - offs[i]) + 1;
  // text over {terminator 0, $ 1, A 2, C 3, G 4, T 5}; one '$' after every read, unique terminator at the end
  std::vector<uint8_t> T(total + 1);
  std::vector<uint64_t> starts(nReads);
  uint64_t p = 0;
  for (uint64_t i = 0; i < nReads; ++i) {
    starts[i] = p;
    uint64_t b = offs[i], e = offs[i + 1];
    if (!reverse) {
      for (uint64_t k = b; k < e; ++k) T[p++] = (uint8_t)(torank(seqs[k]) + 1);
    } else {  // src/indexer.cpp:60-64: reads reversed, not complemented
      for (uint64_t k = e; k > b; --k) T[p++] = (uint8_t)(torank(seqs[k - 1]) + 1);
    }
    T[p++] = 1;
  }
  T[p] = 0;
  std::vector<I> SA(total + 1);
  if (own_sentinels) {
    // `-a sais`: the bucket sort with the reads' own sentinels; the terminator's row comes out first (key 0) like the others'
    if (!parallel_suffix_sort<I, true>(T.data(), total + 1, SA.data(), std::max(threads, 1u))) return false;
  } else if (threads < 2 || total < (1u << 20) || !parallel_suffix_sort<I>(T.data(), total + 1, SA.data(), threads))
    sais<uint8_t, I>(T.data(), SA.data(), (I)(total + 1), (I)6);
  out->runs.clear();
  out->sai.clear();
  out->sai.reserve(nReads);
  out->nStrings = nReads;
  out->nSymbols = total;
  // BWT(sa, reads): src/bwt.cpp:7-32 (run == c && !full -> ++run; else flush)
  uint8_t run = 0;
  auto push = [&](uint32_t rank) {
    if (run) {
      if ((uint32_t)(run >> 5) == rank && (run & 31) != 31) {
        ++run;
        return;
      }
      out->runs.push_back(run);
    }
    run = (uint8_t)((rank << 5) | 1u);
  };
  for (uint64_t k = 1; k <= total; ++k) {  // SA[0] is the terminator
    uint64_t pos = (uint64_t)SA[k];
    uint32_t prevCode = pos == 0 ? 1u : T[pos - 1];
    push(prevCode - 1);
    if (prevCode == 1) {  // maybe the start of a read: SA row with j == 0 (src/suffix_array_builder.cpp:520-531).
      // Decided by position, not by the preceding symbol: a non-ACGT base ranks like the sentinel (alphabet.h:19-39).
      auto it = std::lower_bound(starts.begin(), starts.end(), pos);
      if (it != starts.end() && *it == pos) out->sai.push_back((uint32_t)(it - starts.begin()));
    }
  }
  if (run) out->runs.push_back(run);
  return true;
}

bool BuildStrandIndex(const char* seqs, const uint64_t* offs, uint64_t nReads, bool reverse, StrandIndex* out,
                      std::string* error, unsigned threads, bool own_sentinels) {
  uint64_t total = 0;
  for (uint64_t i = 0; i < nReads; ++i) {
    if (offs[i + 1] < offs[i]) {
      if (error) *error = "bad read offsets";
      return false;
    }
    total += (offs[i + 1] - offs[i]) + 1;
  }
  if (nReads >= 0xFFFFFFFFull) {  // SuffixArray::Elem is uint32 (src/suffix_array.h:33-34)
    if (error) *error = "too many reads for the .sai format";
    return false;
  }
  if (own_sentinels) {
    for (uint64_t k = offs[0]; k < offs[nReads]; ++k)
      if (torank(seqs[k]) == 0) {
        if (error) *error = "algorithm sais: reads with bases other than A, C, G, T are not supported";
        return false;
      }
  }
  try {
    bool ok;
    if (total + 1 < 0x7FFFFFF0ull) ok = build_strand<int32_t>(seqs, offs, nReads, reverse, out, threads, own_sentinels);
    else ok = build_strand<int64_t>(seqs, offs, nReads, reverse, out, threads, own_sentinels);
    if (!ok && error) *error = "algorithm sais: input too repetitive for the bucket sort";
    return ok;
  } catch (const std::bad_alloc&) {
    if (error) *error = "out of memory building the suffix array";
    return false;
  }
}

// The same on the GPU (libsigax: sigax_build_strand).  *rc receives the library's code: SIGAX_E_CAPACITY means the input
// is too repetitive for the device sort and the caller should use BuildStrandIndex.
bool BuildStrandIndexGPU(const char* seqs, const uint64_t* offs, uint64_t nReads, bool reverse, int device, StrandIndex* out,
                         std::string* error, int* rc_out) {
  uint8_t* runs = nullptr;
  uint32_t* sai = nullptr;
  uint64_t nruns = 0, nsym = 0;
  int rc = sigax_build_strand(seqs, offs, nReads, reverse ? 1 : 0, device, &runs, &nruns, &sai, &nsym);
  if (rc_out) *rc_out = rc;
  if (rc != SIGAX_OK) {
    if (error) *error = sigax_last_error();
    return false;
  }
  out->runs.assign(runs, runs + nruns);
  out->sai.assign(sai, sai + nReads);
  out->nStrings = nReads;
  out->nSymbols = nsym;
  sigax_free(runs);
  sigax_free(sai);
  return true;
}

bool StrandIndex::writeBWT(const std::string& path) const {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  uint16_t magic = 0xCACA;
  uint64_t nruns = runs.size();
  int32_t flag = 0;
  bool ok = fwrite(&magic, 2, 1, f) == 1 && fwrite(&nStrings, 8, 1, f) == 1 && fwrite(&nSymbols, 8, 1, f) == 1 &&
            fwrite(&nruns, 8, 1, f) == 1 && fwrite(&flag, 4, 1, f) == 1 &&
            (nruns == 0 || fwrite(runs.data(), 1, nruns, f) == nruns);
  return fclose(f) == 0 && ok;
}

bool StrandIndex::writeSAI(const std::string& path) const {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  char hdr[64];
  const int hn = snprintf(hdr, sizeof(hdr), "%u\n%llu\n%llu\n", 0xCACAu, (unsigned long long)sai.size(), (unsigned long long)sai.size());
  bool ok = fwrite(hdr, 1, (size_t)hn, f) == (size_t)hn;
  // "<readIdx> 0\n" per row (src/suffix_array.cpp:17-44), formatted in slices of 2^20 rows on a few threads (20 M rows of
  // BASELINE configs[2] took one thread a second, digit by digit into a std::string) and written in order
  const size_t slice = (size_t)1 << 20, nslices = (sai.size() + slice - 1) / slice;
  const unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(nslices, 1), std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
  for (size_t base = 0; ok && base < nslices; base += nt) {
    const size_t cnt = std::min<size_t>(nt, nslices - base);
    std::vector<std::string> text(cnt);
    parallel_for(cnt, nt, [&](size_t k) {
      const size_t b0 = (base + k) * slice, e0 = std::min(sai.size(), b0 + slice);
      std::string& o = text[k];
      o.resize((e0 - b0) * 13);  // ten digits at most, " 0\n"
      char* w = &o[0];
      for (size_t i = b0; i < e0; ++i) {
        uint32_t id = sai[i];
        char d[12];
        int n = 0;
        do {
          d[n++] = (char)('0' + id % 10);
          id /= 10;
        } while (id);
        while (n) *w++ = d[--n];
        *w++ = ' ';
        *w++ = '0';
        *w++ = '\n';
      }
      o.resize((size_t)(w - &o[0]));
    });
    for (size_t k = 0; ok && k < cnt; ++k) ok = fwrite(text[k].data(), 1, text[k].size(), f) == text[k].size();
  }
  return fclose(f) == 0 && ok;
}

// ------------------------------------------------------------------------------------------------------
// FMIndex handle
// ------------------------------------------------------------------------------------------------------
FMIndex::~FMIndex() {
  if (_h) sigax_index_close(_h);
}

bool FMIndex::load(const std::string& prefix, FMIndex& fmi, int device) {
  if (fmi._h) {
    sigax_index_close(fmi._h);
    fmi._h = nullptr;
  }
  int rc = sigax_index_open((prefix + ".bwt").c_str(), (prefix + ".rbwt").c_str(), (prefix + ".sai").c_str(),
                            (prefix + ".rsai").c_str(), device, &fmi._h);
  return rc == SIGAX_OK;
}
// the forward index alone: FMIndex::load(prefix + ".bwt") of src/correct.cpp:41-47 (`siga index --no-reverse` writes no more)
bool FMIndex::loadForward(const std::string& prefix, FMIndex& fmi, int device) {
  if (fmi._h) {
    sigax_index_close(fmi._h);
    fmi._h = nullptr;
  }
  return sigax_index_open((prefix + ".bwt").c_str(), nullptr, nullptr, nullptr, device, &fmi._h) == SIGAX_OK;
}

uint64_t FMIndex::length() const {
  sigax_index_info inf;
  if (!_h || sigax_index_info_get(_h, &inf) != SIGAX_OK) return 0;
  return inf.n_symbols;
}

// ------------------------------------------------------------------------------------------------------
// ASQG output
// ------------------------------------------------------------------------------------------------------
// Utils::ofstream (src/utils.cpp:92-126): gzip when the name ends with .gz.  The gzip stream is ONE member (what any
// gzip reader, boost's gzip_decompressor included, accepts) whose deflate data is produced block-wise by a pool of
// threads: every 1 MiB of the text, counted from the start of the stream, is deflated on its own as raw deflate ending in
// a sync flush, the blocks are concatenated in order and the CRC-32s are combined (the pigz scheme, without dictionary
// priming).  Block boundaries depend on the text alone, so the file's bytes do not depend on how many threads, batches
// or GPUs produced it.
// Blocks of the stream deflated ahead of their turn (VtAhead below): block J = bytes [J, J + 1) MiB of the text.  state: 0 not
// there, 1 ready (out/crc hold what deflate_block gives for the block's text), 2 the text has changed since.
// (a block whose text changes may still be on its way in the thread that deflates ahead: that thread only ever turns a 0 into
// a 1, so the 2 stays whichever of the two comes first)
struct SpecBlocks {
  std::vector<std::string> out;
  std::vector<uLong> crc;
  std::unique_ptr<std::atomic<uint8_t>[]> state;
  size_t n = 0;
  void resize(size_t k) {
    out.resize(k);
    crc.resize(k, 0);
    state.reset(new std::atomic<uint8_t>[k]);
    for (size_t i = 0; i < k; ++i) state[i].store(0);
    n = k;
  }
  void made(size_t J) {
    uint8_t none = 0;
    state[J].compare_exchange_strong(none, 1);
  }
};

class OutFile {
 public:
  explicit OutFile(const std::string& path, unsigned threads = 0) : _f(nullptr), _gz(false), _bad(false), _crc(0), _total(0), _nt(threads) {
    _gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    _f = fopen(path.c_str(), "wb");
    // deflate at level 6 makes ~12 MB/s per thread on read text: the writer takes up to 128 threads whatever -t says
    _nt = std::max(_nt, std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 128));
    if (const char* env = getenv("SIGA_HOST_THREADS")) _nt = std::max(1, atoi(env));
    if (_f && _gz) {
      static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
      put(hdr, 10);
      _crc = crc32(0L, Z_NULL, 0);
    }
  }
  ~OutFile() { close(); }
  bool ok() const { return _f != nullptr; }
  void write(const char* p, size_t n) {
    _buf.append(p, n);
    if (_buf.size() >= kFlush) {
      std::vector<std::string> none;
      write_parts(none);
    }
  }
  void write(const std::string& s) { write(s.data(), s.size()); }
  // the concatenation of `parts` goes out next; whole blocks are deflated now (in parallel), the rest waits in _buf
  // (spec: blocks of this stream that were deflated ahead; the caller vouches that a block in state 1 holds the deflate of
  // exactly the text that arrives here for it)
  void write_parts(const std::vector<std::string>& parts, SpecBlocks* spec = nullptr) {
    if (!_f) return;
    if (!_gz) {
      if (!_buf.empty()) put(_buf.data(), _buf.size());
      _buf.clear();
      for (const std::string& p : parts)
        if (!p.empty()) put(p.data(), p.size());
      return;
    }
    std::vector<const std::string*> segs;
    std::vector<size_t> start;  // offset of each segment in the pending text
    size_t total = 0;
    auto add = [&](const std::string* x) {
      if (x->empty()) return;
      segs.push_back(x);
      start.push_back(total);
      total += x->size();
    };
    add(&_buf);
    for (const std::string& p : parts) add(&p);
    const size_t nfull = total / kBlock;
    auto gather = [&](size_t off, size_t len, char* dst) {
      size_t k = (size_t)(std::upper_bound(start.begin(), start.end(), off) - start.begin()) - 1;
      while (len) {
        const size_t in = off - start[k], take = std::min(len, segs[k]->size() - in);
        memcpy(dst, segs[k]->data() + in, take);
        dst += take;
        off += take;
        len -= take;
        ++k;
      }
    };
    if (nfull) {
      std::vector<std::string> outs(nfull);
      std::vector<uLong> crcs(nfull, 0);
      const size_t first_block = (size_t)(_total / kBlock);
      parallel_for(nfull, _nt, [&](size_t i) {
        const size_t J = first_block + i;
        if (spec && J < spec->n && spec->state[J].load() == 1) {
          outs[i].swap(spec->out[J]);
          crcs[i] = spec->crc[J];
          return;
        }
        const size_t off = i * kBlock, k = (size_t)(std::upper_bound(start.begin(), start.end(), off) - start.begin()) - 1;
        if (off - start[k] + kBlock <= segs[k]->size()) {  // the block lies in one segment: no copy
          deflate_block(segs[k]->data() + (off - start[k]), kBlock, false, &outs[i], &crcs[i]);
          return;
        }
        std::string tmp(kBlock, '\0');
        gather(off, kBlock, &tmp[0]);
        deflate_block(tmp.data(), kBlock, false, &outs[i], &crcs[i]);
      });
      for (size_t i = 0; i < nfull; ++i) {
        put_owned(std::move(outs[i]));
        _crc = crc32_combine(_crc, crcs[i], (z_off_t)kBlock);
      }
      _total += nfull * kBlock;
    }
    std::string rest(total - nfull * kBlock, '\0');
    if (!rest.empty()) gather(nfull * kBlock, rest.size(), &rest[0]);
    _buf.swap(rest);
  }
  static size_t block_bytes() { return kBlock; }
  static void deflate_ahead(const char* in, std::string* out, uLong* crc) { deflate_block(in, kBlock, false, out, crc); }
  bool gz() const { return _gz; }
  static bool gz_name(const std::string& path) { return path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0; }
  bool close() {
    if (!_f) return true;
    std::vector<std::string> none;
    write_parts(none);
    if (_gz) {
      std::string out;
      uLong crc = 0;
      deflate_block(_buf.data(), _buf.size(), true, &out, &crc);  // the last (possibly empty) block ends the deflate stream
      put(out.data(), out.size());
      _crc = crc32_combine(_crc, crc, (z_off_t)_buf.size());
      _total += _buf.size();
      _buf.clear();
      unsigned char tail[8];
      uint32_t c = (uint32_t)_crc, n = (uint32_t)_total;
      for (int i = 0; i < 4; ++i) { tail[i] = (unsigned char)(c >> (8 * i)); tail[4 + i] = (unsigned char)(n >> (8 * i)); }
      put(tail, 8);
    }
    finish_writes();
    // a short write (disk full, I/O error) leaves the stream's error flag set while fclose may still return 0
    bool ok = !_bad && ferror(_f) == 0;
    ok = fclose(_f) == 0 && ok;
    _f = nullptr;
    return ok;
  }

 private:
  static const size_t kBlock = 1 << 20, kFlush = 64u << 20;
  // On read text (four-letter sequences with little to match inside a 32 KiB window) zlib's level 6, what the
  // reference's gzip filter uses, makes 10 MB/s per thread, level 4 62 MB/s for a file 5 % larger; with the kernels done
  // in milliseconds the deflate of the VT lines was the longest phase of `siga overlap`.  The writer's own coder
  // (line_deflate.hpp: matches against the line above, field by field) makes 650 MB/s per thread on VT lines for a
  // stream 9 % SMALLER than level 6, and 300 MB/s on ED lines at level 4's size.
  // SIGA_GZIP_LEVEL=<1..9> sends every block through zlib at that level instead (6 = the reference's setting).
  static int gzip_level(bool* forced = nullptr) {
    static const int env_level = [] {
      const char* env = getenv("SIGA_GZIP_LEVEL");
      const int l = env ? atoi(env) : 0;
      return l < 1 || l > 9 ? 0 : l;
    }();
    if (forced) *forced = env_level != 0;
    return env_level ? env_level : 4;
  }
  static void deflate_block(const char* in, size_t n, bool last, std::string* out, uLong* crc) {
    *crc = ldef::crc32_fast(0, (const unsigned char*)in, n,
                            [](uint32_t c, const unsigned char* p, size_t k) { return (uint32_t)crc32(c, (const Bytef*)p, (uInt)k); });
    bool forced = false;
    const int level = gzip_level(&forced);
    if (!forced) {
      ldef::deflate_lines((const unsigned char*)in, n, last, out);
      return;
    }
    z_stream z;
    memset(&z, 0, sizeof(z));
    deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    out->resize(deflateBound(&z, (uLong)n) + 16);
    z.next_in = (Bytef*)in;
    z.avail_in = (uInt)n;
    z.next_out = (Bytef*)&(*out)[0];
    z.avail_out = (uInt)out->size();
    deflate(&z, last ? Z_FINISH : Z_SYNC_FLUSH);
    out->resize(out->size() - z.avail_out);
    deflateEnd(&z);
  }
  // The bytes go to the file from a thread of their own (1.2 GB of deflate blocks at BASELINE configs[2]: 0.26 s of
  // fwrite that the thread formatting the next batch's lines used to spend between two batches); at most 1 GiB waits.
  // SIGA_SYNC_WRITE=1: written by the caller.
  void put(const void* p, size_t n) {
    if (!n) return;
    if (!_async) {
      if (fwrite(p, 1, n, _f) != n) _bad = true;
      return;
    }
    put_owned(std::string((const char*)p, n));
  }
  void put_owned(std::string&& s) {
    if (s.empty()) return;
    if (!_async) {
      if (fwrite(s.data(), 1, s.size(), _f) != s.size()) _bad = true;
      return;
    }
    std::unique_lock<std::mutex> g(_wmu);
    if (!_wt.joinable()) _wt = std::thread([this] { drain(); });
    _wcv.wait(g, [&] { return _wq_bytes <= ((size_t)1 << 30); });
    _wq_bytes += s.size();
    _wq.push_back(std::move(s));
    _wcv.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> g(_wmu);
    for (;;) {
      _wcv.wait(g, [&] { return !_wq.empty() || _wdone; });
      if (_wq.empty()) return;
      std::deque<std::string> mine;
      mine.swap(_wq);
      g.unlock();
      size_t bytes = 0;
      for (const std::string& x : mine) {
        if (!_bad && fwrite(x.data(), 1, x.size(), _f) != x.size()) _bad = true;
        bytes += x.size();
      }
      mine.clear();
      g.lock();
      _wq_bytes -= bytes;
      _wcv.notify_all();
    }
  }
  void finish_writes() {
    {
      std::lock_guard<std::mutex> g(_wmu);
      _wdone = true;
    }
    _wcv.notify_all();
    if (_wt.joinable()) _wt.join();
  }
  FILE* _f;
  bool _gz;
  std::atomic<bool> _bad;
  bool _async = getenv("SIGA_SYNC_WRITE") == nullptr;
  std::thread _wt;
  std::mutex _wmu;
  std::condition_variable _wcv;
  std::deque<std::string> _wq;
  size_t _wq_bytes = 0;
  bool _wdone = false;
  uLong _crc;
  uint64_t _total;
  unsigned _nt;
  std::string _buf;
};

static void append_u64(std::string& s, uint64_t v) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) s.push_back(tmp[--n]);
}

// TagValue<T>::fromstring (src/asqg.h:43-56): exactly three ':'-separated tokens, one-letter type code
static bool tag_tokens(const std::string& text, char code, std::string* value) {
  size_t a = text.find(':');
  if (a == std::string::npos) return false;
  size_t b = text.find(':', a + 1);
  if (b == std::string::npos) return false;
  if (text.find(':', b + 1) != std::string::npos) return false;
  if (b - a - 1 != 1 || text[a + 1] != code) return false;
  *value = text.substr(b + 1);
  return true;
}
static std::string first_word(const std::string& s) {  // std::istream >> std::string
  size_t b = 0;
  auto sp = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  while (b < s.size() && sp(s[b])) ++b;
  size_t e = b;
  while (e < s.size() && !sp(s[e])) ++e;
  return s.substr(b, e - b);
}
static int parse_int(const std::string& s) {  // std::istream >> int (0 on failure, clamped on overflow)
  size_t i = 0;
  auto sp = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  while (i < s.size() && sp(s[i])) ++i;
  bool neg = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
  if (i >= s.size() || s[i] < '0' || s[i] > '9') return 0;
  long long v = 0;
  while (i < s.size() && s[i] >= '0' && s[i] <= '9') {
    v = v * 10 + (s[i++] - '0');
    if (v > 4000000000LL) v = 4000000000LL;
  }
  if (neg) v = -v;
  if (v > 2147483647LL) v = 2147483647LL;
  if (v < -2147483648LL) v = -2147483648LL;
  return (int)v;
}

// OverlapPostProcess::operator() + VertexRecord << (src/overlap_builder.cpp:301-322, src/asqg.cpp:171-186)
static void write_vertex(std::string& o, std::string_view name, std::string_view comment_sv, std::string_view seq, bool substring) {
  bool hasCov = false, hasBar = false, hasExt = false;
  int cov = 0;
  std::string bar, ext, val;
  if (!comment_sv.empty()) {
    const std::string comment(comment_sv);
    size_t b = 0;
    while (true) {
      size_t e = comment.find(' ', b);
      std::string tok = comment.substr(b, e == std::string::npos ? std::string::npos : e - b);
      if (tok.compare(0, 2, "BX") == 0) {
        if (tag_tokens(tok, 'Z', &val)) { bar = first_word(val); hasBar = true; }
      } else if (tok.compare(0, 2, "CR") == 0) {
        if (tag_tokens(tok, 'i', &val)) { cov = parse_int(val); hasCov = true; }
      } else if (tok.compare(0, 2, "EX") == 0) {
        if (tag_tokens(tok, 'Z', &val)) { ext = first_word(val); hasExt = true; }
      }
      if (e == std::string::npos) break;
      b = e + 1;
    }
  }
  o += "VT\t";
  o.append(name.data(), name.size());
  o += '\t';
  o.append(seq.data(), seq.size());
  o += substring ? "\tSS:i:1" : "\tSS:i:0";
  if (hasCov) { o += "\tCR:i:"; o += std::to_string(cov); }
  if (hasBar) { o += "\tBX:Z:"; o += bar; }
  if (hasExt) { o += "\tEX:Z:"; o += ext; }
  o += '\n';
}

// EdgeRecord << (src/asqg.cpp:228-237, src/coord.cpp:4-80) with OverlapBlock::overlap's coordinates
// (src/overlap_builder.cpp:158-175)
// ... written through a raw pointer: the caller has room for the two names and 3 + 2 + 6 x 21 + 4 bytes more (22 M lines at
// BASELINE configs[2]: the capacity check of every append was a third of the formatting)
static inline char* put_u64(char* p, uint64_t v) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}
static const size_t kEdgeLineExtra = 3 + 2 + 6 * 21 + 4;
static char* write_edge(char* p, const sigax_edge& e, const ReadStore& reads, const uint32_t* lengths) {
  const std::string_view qn = reads.name(e.query), tn = reads.name(e.target);
  uint64_t ql = lengths[e.query], tl = lengths[e.target], len = e.length;
  uint64_t s0 = ql - len, e0 = ql - 1, s1 = 0, e1 = len - 1;
  if (e.af & 1u) { uint64_t t = s0; s0 = ql - e0 - 1; e0 = ql - t - 1; }
  if (e.af & 2u) { uint64_t t = s1; s1 = tl - e1 - 1; e1 = tl - t - 1; }
  *p++ = 'E'; *p++ = 'D'; *p++ = '\t';
  memcpy(p, qn.data(), qn.size());
  p += qn.size();
  *p++ = ' ';
  memcpy(p, tn.data(), tn.size());
  p += tn.size();
  *p++ = ' ';
  p = put_u64(p, s0); *p++ = ' ';
  p = put_u64(p, e0); *p++ = ' ';
  p = put_u64(p, ql); *p++ = ' ';
  p = put_u64(p, s1); *p++ = ' ';
  p = put_u64(p, e1); *p++ = ' ';
  p = put_u64(p, tl); *p++ = ' ';
  *p++ = (e.af & 4u) ? '1' : '0';
  *p++ = ' '; *p++ = '0'; *p++ = '\n';
  return p;
}


// ED text of `cnt` edge records in chunks of ed_chunk lines, one string per chunk, on nt threads: every chunk gets room for
// the longest lines there can be (two names of max_name bytes + kEdgeLineExtra) and is written through a raw pointer.
static void format_edge_text(const ReadStore& reads, const uint32_t* read_len, const sigax_edge* e, uint64_t cnt, unsigned nt, uint32_t max_name,
                             size_t ed_chunk, std::vector<std::string>* parts) {
  parts->assign((cnt + ed_chunk - 1) / ed_chunk, std::string());
  parallel_for(parts->size(), nt, [&](size_t c) {
    const uint64_t cb = c * ed_chunk, ce = std::min<uint64_t>(cnt, cb + ed_chunk);
    std::string& o = (*parts)[c];
    o.resize((ce - cb) * (2 * (size_t)max_name + kEdgeLineExtra));  // room for the longest line there can be, times the lines
    char* w = &o[0];
    // a target's name is three dependent misses away (header offset, name length, the bytes in the file image):
    // asked for sixteen and eight edges ahead
    for (uint64_t i = cb; i < ce; ++i) {
      if (i + 16 < ce) {
        const uint32_t t = e[i + 16].target;
        __builtin_prefetch(&reads.head_off[t]);
        __builtin_prefetch(&reads.name_len[t]);
        __builtin_prefetch(&read_len[t]);
      }
      if (i + 8 < ce) __builtin_prefetch(reads.file.data() + reads.head_off[e[i + 8].target]);
      w = write_edge(w, e[i], reads, read_len);
    }
    o.resize((size_t)(w - &o[0]));
    o.shrink_to_fit();  // the text may be held until the last batch is through: not with three times its size in reserve
  });
}

struct PhaseTimer {  // SIGA_TIMING=1: phase times on stderr
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseTimer() : on(getenv("SIGA_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    auto n = std::chrono::steady_clock::now();
    if (on) fprintf(stderr, "[siga] %-28s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

// ReadInfo{name,length} of the edge converter (src/overlap_builder.cpp:333-343) as lengths + rank of each name under
// std::string operator< (equal names, equal rank).  A sample sort on the host threads: names enter as (first eight bytes,
// big endian; index) pairs -- the file image is only gone back to on a tie --, splitters from a sample cut them into
// buckets of equal names' ranges, the buckets are sorted side by side, and the ranks follow from the distinct names
// counted per bucket.  (Round 2 merged sorted runs pairwise: the last merges ran on one thread, 1.3 s for 20 M names.)
static void name_ranks(const ReadStore& rs, unsigned nt, std::vector<uint32_t>* lengths, std::vector<uint32_t>* ranks) {
  const size_t n = rs.size();
  lengths->resize(n);
  ranks->resize(n);
  if (n == 0) return;
  struct Key {
    uint64_t k;
    uint32_t i;
  };
  auto less = [&](const Key& a, const Key& b) { return a.k != b.k ? a.k < b.k : rs.name(a.i) < rs.name(b.i); };
  auto same = [&](const Key& a, const Key& b) { return a.k == b.k && rs.name(a.i) == rs.name(b.i); };
  std::vector<Key> keys(n), sorted(n);
  const size_t chunks = std::max<size_t>(1, std::min<size_t>((size_t)nt * 4, n / 4096));
  const size_t step = (n + chunks - 1) / chunks;
  parallel_for(chunks, nt, [&](size_t c) {
    const size_t b = std::min(n, c * step), e = std::min(n, b + step);
    for (size_t i = b; i < e; ++i) {
      (*lengths)[i] = (uint32_t)(rs.offs[i + 1] - rs.offs[i]);
      const std::string_view nm = rs.name(i);
      uint64_t k = 0;
      for (size_t j = 0; j < 8; ++j) k = (k << 8) | (j < nm.size() ? (unsigned char)nm[j] : 0u);
      keys[i] = {k, (uint32_t)i};
    }
  });
  // splitters: every bucket takes the names in [splitter b-1, splitter b)
  const size_t nbuckets = chunks > 1 ? std::min<size_t>((size_t)nt * 8, 1024) : 1;
  std::vector<Key> splitters;
  if (nbuckets > 1) {
    const size_t nsample = std::min(n, nbuckets * 64);
    std::vector<Key> sample(nsample);
    for (size_t j = 0; j < nsample; ++j) sample[j] = keys[(size_t)((unsigned __int128)j * n / nsample)];
    std::sort(sample.begin(), sample.end(), less);
    for (size_t b = 1; b < nbuckets; ++b) splitters.push_back(sample[b * nsample / nbuckets]);
  }
  auto bucket_of = [&](const Key& x) {  // splitters <= x
    return (size_t)(std::upper_bound(splitters.begin(), splitters.end(), x, less) - splitters.begin());
  };
  std::vector<uint32_t> which(n);
  std::vector<size_t> count(chunks * nbuckets, 0);
  parallel_for(chunks, nt, [&](size_t c) {
    const size_t b = std::min(n, c * step), e = std::min(n, b + step);
    size_t* cnt = &count[c * nbuckets];
    for (size_t i = b; i < e; ++i) cnt[which[i] = (uint32_t)bucket_of(keys[i])]++;
  });
  std::vector<size_t> bstart(nbuckets + 1, 0);
  {
    size_t at = 0;  // bucket-major, chunk-minor: a chunk's share of a bucket starts at count[c][b] afterwards
    for (size_t b = 0; b < nbuckets; ++b) {
      bstart[b] = at;
      for (size_t c = 0; c < chunks; ++c) {
        const size_t k = count[c * nbuckets + b];
        count[c * nbuckets + b] = at;
        at += k;
      }
    }
    bstart[nbuckets] = at;
  }
  parallel_for(chunks, nt, [&](size_t c) {
    const size_t b = std::min(n, c * step), e = std::min(n, b + step);
    size_t* at = &count[c * nbuckets];
    for (size_t i = b; i < e; ++i) sorted[at[which[i]]++] = keys[i];
  });
  std::vector<uint32_t> distinct(nbuckets, 0);  // names of a bucket that differ from their predecessor IN the bucket
  parallel_for(nbuckets, nt, [&](size_t b) {
    Key* lo = sorted.data() + bstart[b];
    Key* hi = sorted.data() + bstart[b + 1];
    std::sort(lo, hi, less);
    uint32_t d = 0;
    for (Key* q = lo + 1; q < hi; ++q) d += same(q[-1], q[0]) ? 0u : 1u;
    distinct[b] = d;
  });
  // rank of a bucket's first name: the distinct names before it (buckets hold disjoint ranges of names)
  std::vector<uint32_t> first(nbuckets, 0);
  {
    uint32_t rk = 0;
    bool any = false;
    for (size_t b = 0; b < nbuckets; ++b) {
      if (bstart[b] == bstart[b + 1]) continue;
      if (any) ++rk;  // its first name is a new one
      first[b] = rk;
      rk += distinct[b];
      any = true;
    }
  }
  parallel_for(nbuckets, nt, [&](size_t b) {
    uint32_t rk = first[b];
    for (size_t k = bstart[b]; k < bstart[b + 1]; ++k) {
      if (k > bstart[b] && !same(sorted[k - 1], sorted[k])) ++rk;
      (*ranks)[sorted[k].i] = rk;
    }
  });
}

// ------------------------------------------------------------------------------------------------------
// OverlapBuilder::build (src/overlap_builder.cpp:423-509).  The reference reads items in batches of threads x batch-size,
// runs overlap() on them under OpenMP and post-processes the batch serially in input order (parallel::foreach,
// src/parallel_framework.h:16-59).  Here: the reads are parsed once (in parallel), cut into device batches sized from
// free HBM, and every GPU of the run keeps two batch objects in flight (upload / kernels / download overlap); the main
// thread takes finished batches in input order, formats their VT lines on the host threads and feeds the block-parallel
// gzip writer; the ED lines follow in hits order from the collected 16-byte edge records.  With --gpus N the index is
// replicated device to device and batches go to whichever GPU is free: the output does not depend on N.
// ------------------------------------------------------------------------------------------------------
// VT lines ahead of the GPU.  A VT line is known from the reads file but for one character, the digit of SS:i: (is the read
// a substring of another: the device's answer), and in a set that went through `siga rmdup` -- what the extractor asks for,
// src/overlap_builder.cpp:755-756 -- that digit is 0.  So the text of every VT line is written with SS:i:0 as soon as the
// reads are parsed, and the 1 MiB blocks of the output stream (fixed offsets of the TEXT, whose length the digit does not
// change) are deflated, by threads of this object, while the index is still on its way to the GPU and while the batches
// run.  build() takes the chunks in input order once the batches that cover them are back: a chunk with a substring read
// is formatted again and the blocks it touches are deflated again by the writer; every other block goes to the file as
// it is.  The bytes of the file are the ones the in-order path writes.  An option (SIGA_VT_AHEAD=1), see vt_ahead_wanted().
class VtAhead {
 public:
  static constexpr size_t kChunk = 4096;  // reads per chunk of text
  VtAhead(std::shared_ptr<const ReadStore> keep, const ReadStore* reads, unsigned nt, const std::string& header, bool gz)
      : _keep(std::move(keep)), _reads(*reads), _nt(std::max(1u, nt)), _header(header), _gz(gz), _n(reads->size()) {
    _nchunks = (_n + kChunk - 1) / kChunk;
    _text.resize(_nchunks);
    _off.assign(_nchunks + 1, 0);
    _off[0] = header.size();
    // room for the longest text there can be: "VT\t" name "\t" seq "\tSS:i:0" + the three tags (each shorter than the
    // comment it is cut from, plus its six characters) + "\n"
    uint64_t bound = header.size();
    for (size_t i = 0; i < _n; ++i) bound += 3 + 1 + 7 + 1 + 18 + 2 * (uint64_t)_reads.head_len[i] + (_reads.offs[i + 1] - _reads.offs[i]);
    if (gz) _spec.resize((size_t)(bound / OutFile::block_bytes()) + 1);
    if (const char* env = getenv("SIGA_VT_AHEAD_BYTES")) _cap = std::max<uint64_t>(strtoull(env, nullptr, 10), 1);
    _thread = std::thread([this] { run(); });
  }
  ~VtAhead() {
    {
      std::lock_guard<std::mutex> g(_mu);
      _stop = true;
    }
    _cv.notify_all();
    if (_thread.joinable()) _thread.join();
  }
  VtAhead(const VtAhead&) = delete;
  VtAhead& operator=(const VtAhead&) = delete;
  const std::string& header() const { return _header; }
  bool gz() const { return _gz; }
  size_t chunks() const { return _nchunks; }
  // The chunks [from, to) -- formatted, their blocks deflated -- with the substring flags of their reads applied; `parts`
  // takes their text.  Call with from = the previous call's to.
  void take(size_t from, size_t to, const uint8_t* substring, std::vector<std::string>* parts) {
    {
      std::unique_lock<std::mutex> g(_mu);
      _want = to;  // (whatever the cap says: these chunks are waited for)
      _cv.notify_all();
      _cv.wait(g, [&] { return _done >= to; });
    }
    parts->clear();
    parts->resize(to - from);
    std::vector<uint8_t> again(to - from, 0);
    parallel_for(to - from, _nt, [&](size_t k) {
      const size_t c = from + k, cb = c * kChunk, ce = std::min(_n, cb + kChunk);
      bool any = false;
      for (size_t i = cb; substring && i < ce && !any; ++i) any = substring[i] != 0;
      if (any) {
        std::string o;
        o.reserve(_text[c].size());
        for (size_t i = cb; i < ce; ++i) write_vertex(o, _reads.name(i), _reads.comment(i), _reads.seq(i), substring[i] != 0);
        _text[c].swap(o);
        again[k] = 1;
      }
      (*parts)[k].swap(_text[c]);
      std::string().swap(_text[c]);
    });
    // the blocks a re-written chunk touches are the writer's to deflate
    const size_t kb = OutFile::block_bytes();
    for (size_t k = 0; _gz && k < to - from; ++k) {
      const size_t c = from + k;
      if (!again[k] || _off[c + 1] == _off[c]) continue;
      for (size_t J = (size_t)(_off[c] / kb); J <= (size_t)((_off[c + 1] - 1) / kb) && J < _spec.n; ++J) _spec.state[J].store(2);
    }
  }
  SpecBlocks* blocks() { return _gz ? &_spec : nullptr; }
  // the text of chunks below `to` has left: the threads may run further ahead
  void taken(size_t to) {
    {
      std::lock_guard<std::mutex> g(_mu);
      _taken_off = _off[to];
    }
    _cv.notify_all();
  }

 private:
  void run() {
    const size_t kb = OutFile::block_bytes();
    const size_t wave = (size_t)_nt * 8;
    std::string carry = _header;   // text of the stream from `carry_off` on that is in no finished block yet
    uint64_t carry_off = 0;        // a multiple of the block size
    for (size_t c0 = 0; c0 < _nchunks; c0 += wave) {
      {
        std::unique_lock<std::mutex> g(_mu);
        _cv.wait(g, [&] { return _stop || c0 < _want || _off[c0] - _taken_off <= _cap; });
        if (_stop) return;
      }
      const size_t c1 = std::min(_nchunks, c0 + wave);
      parallel_for(c1 - c0, _nt, [&](size_t k) {
        const size_t c = c0 + k, cb = c * kChunk, ce = std::min(_n, cb + kChunk);
        std::string& o = _text[c];
        o.reserve((size_t)(_reads.offs[ce] - _reads.offs[cb]) + (ce - cb) * 32);
        for (size_t i = cb; i < ce; ++i) write_vertex(o, _reads.name(i), _reads.comment(i), _reads.seq(i), false);
      });
      for (size_t c = c0; c < c1; ++c) _off[c + 1] = _off[c] + _text[c].size();
      if (_gz) {
        // pending text = carry + the wave's chunks, from carry_off on
        std::vector<const std::string*> segs;
        std::vector<uint64_t> start;
        uint64_t total = 0;
        auto add = [&](const std::string* x) {
          if (x->empty()) return;
          segs.push_back(x);
          start.push_back(total);
          total += x->size();
        };
        add(&carry);
        for (size_t c = c0; c < c1; ++c) add(&_text[c]);
        auto gather = [&](uint64_t off, size_t len, char* dst) {
          size_t k = (size_t)(std::upper_bound(start.begin(), start.end(), off) - start.begin()) - 1;
          while (len) {
            const size_t in = (size_t)(off - start[k]), take = std::min(len, segs[k]->size() - in);
            memcpy(dst, segs[k]->data() + in, take);
            dst += take;
            off += take;
            len -= take;
            ++k;
          }
        };
        const size_t nfull = (size_t)(total / kb), J0 = (size_t)(carry_off / kb);
        parallel_for(nfull, _nt, [&](size_t i) {
          if (J0 + i >= _spec.n) return;
          const uint64_t off = (uint64_t)i * kb;
          const size_t k = (size_t)(std::upper_bound(start.begin(), start.end(), off) - start.begin()) - 1;
          if (off - start[k] + kb <= segs[k]->size()) {
            OutFile::deflate_ahead(segs[k]->data() + (off - start[k]), &_spec.out[J0 + i], &_spec.crc[J0 + i]);
          } else {
            std::string tmp(kb, '\0');
            gather(off, kb, &tmp[0]);
            OutFile::deflate_ahead(tmp.data(), &_spec.out[J0 + i], &_spec.crc[J0 + i]);
          }
          _spec.made(J0 + i);
        });
        std::string rest((size_t)(total - (uint64_t)nfull * kb), '\0');
        if (!rest.empty()) gather((uint64_t)nfull * kb, rest.size(), &rest[0]);
        carry.swap(rest);
        carry_off += (uint64_t)nfull * kb;
      }
      {
        std::lock_guard<std::mutex> g(_mu);
        _done = c1;
      }
      _cv.notify_all();
    }
  }

  std::shared_ptr<const ReadStore> _keep;
  const ReadStore& _reads;
  unsigned _nt;
  std::string _header;
  bool _gz;
  size_t _n, _nchunks = 0;
  std::vector<std::string> _text;
  std::vector<uint64_t> _off;  // _off[c]: where chunk c starts in the stream (the header first)
  SpecBlocks _spec;
  uint64_t _cap = (uint64_t)3 << 29;  // text held ahead of the writer at most (SIGA_VT_AHEAD_BYTES)
  std::mutex _mu;
  std::condition_variable _cv;
  size_t _done = 0, _want = 0;
  uint64_t _taken_off = 0;
  bool _stop = false;
  std::thread _thread;
};

static std::string asqg_header(size_t minOverlap) {
  // src/overlap_builder.cpp:428-437 (the IN tag is never written: :494-495)
  return "HT\tVN:i:1\tOL:i:" + std::to_string((int)minOverlap) + "\tCN:i:1\n";
}

namespace {
struct BatchOut {
  std::vector<uint8_t> substring;
  sigax_edge* edges = nullptr;
  uint64_t n_edges = 0;
  bool ready = false;
};
struct Pipeline {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<BatchOut> out;
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  std::string error;
  void fail(const std::string& e) {
    std::lock_guard<std::mutex> g(mu);
    if (!failed.exchange(true)) error = e;
    cv.notify_all();
  }
};
}  // namespace

}  // namespace sigah
struct sigah::OverlapBuilder::Preloaded {
  std::string path;
  ReadStore reads;
  std::vector<uint32_t> lengths, ranks;
  bool ok = false;
  std::unique_ptr<VtAhead> ahead;  // (after `reads`: gone before them)
};
namespace sigah {
// Off unless asked for (SIGA_VT_AHEAD=1): on the 16-core boxes the phases of `siga overlap` already keep every core busy, and
// text made early is text the name ranks and the index load wait for (20 M reads 1.53 s without, 1.84-1.95 s with).
static bool vt_ahead_wanted() {
  const char* env = getenv("SIGA_VT_AHEAD");
  return env && atoi(env) > 0 && getenv("SIGA_NO_VT_AHEAD") == nullptr;
}

void OverlapBuilder::preload(const std::string& input, size_t threads, long minOverlap, const std::string& output) const {
  const unsigned nt = host_threads(threads);
  auto p = std::make_shared<Preloaded>();
  p->path = input;
  PhaseTimer pt;
  p->ok = LoadReads(input, &p->reads, nt);
  pt.lap("  reads parsed");
  if (p->ok) name_ranks(p->reads, nt, &p->lengths, &p->ranks);
  pt.lap("  names ranked");
  // (an option) the VT lines start now, while the index is still on its way to the GPU
  if (p->ok && minOverlap >= 0 && !output.empty() && vt_ahead_wanted())
    p->ahead.reset(new VtAhead(nullptr, &p->reads, nt, asqg_header((size_t)minOverlap), OutFile::gz_name(output)));
  _pre = p;
}

static std::vector<int> device_list(int first, int count) {
  // SIGA_DEVICE_MAP=0,0,...: logical GPU k of the run is physical device map[k] (rehearsing --gpus N on fewer GPUs)
  std::vector<int> map;
  if (const char* env = getenv("SIGA_DEVICE_MAP")) {
    for (const char* p = env; *p;) {
      map.push_back(atoi(p));
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  std::vector<int> d;
  for (int k = 0; k < std::max(count, 1); ++k) d.push_back((size_t)k < map.size() ? map[k] : first + k);
  return d;
}

bool OverlapBuilder::build(const std::string& input, size_t minOverlap, const std::string& output, size_t threads,
                           size_t batch, size_t* processed) const {
  (void)processed;  // accepted and never written, like the reference (src/overlap_builder.cpp:423-424)
  PhaseTimer pt;
  _error.clear();
  if (!_fmi || !_fmi->handle()) {
    _error = "FMIndex not loaded";
    return false;
  }
  const unsigned nt = host_threads(threads);
  std::shared_ptr<Preloaded> pre = _pre;
  _pre.reset();
  if (!pre || pre->path != input) {
    pre = std::make_shared<Preloaded>();
    pre->path = input;
    pre->ok = LoadReads(input, &pre->reads, nt);
    if (pre->ok) name_ranks(pre->reads, nt, &pre->lengths, &pre->ranks);
  }
  if (!pre->ok) {
    _error = "Failed to read file " + input;
    return false;
  }
  ReadStore& reads = pre->reads;
  pt.lap("parse reads + name ranks");
  OutFile out(output, nt);
  if (!out.ok()) {
    _error = "Failed to create ASQG " + output;
    return false;
  }
  const std::string header = asqg_header(minOverlap);
  out.write(header);
  const size_t n = reads.size();
  // the VT lines ahead of the batches (started by preload() when it knew the header; from here otherwise)
  std::unique_ptr<VtAhead> ahead = std::move(pre->ahead);
  if (ahead && (ahead->header() != header || ahead->gz() != out.gz())) ahead.reset();
  if (!ahead && vt_ahead_wanted() && n > 0) ahead.reset(new VtAhead(nullptr, &reads, nt, header, out.gz()));
  if (!vt_ahead_wanted()) ahead.reset();
  uint32_t maxLen = 0;
  for (uint32_t l : pre->lengths) maxLen = std::max(maxLen, l);
  if (n > 0 && sigax_index_set_reads(_fmi->handle(), pre->lengths.data(), pre->ranks.data(), n) != SIGAX_OK) {
    _error = std::string("failed to load suffix array index: ") + sigax_last_error();
    return false;
  }
  pt.lap("read info to the device");
  const uint32_t flags = SIGAX_EDGES | (_irreducible ? SIGAX_IRREDUCIBLE : 0u) | (_rc ? SIGAX_RC : 0u);
  // the GPUs of this run: the loaded index on the first, device-to-device replicas on the others
  sigax_index_info inf;
  sigax_index_info_get(_fmi->handle(), &inf);
  const std::vector<int> devs = device_list(inf.device, _gpus);
  std::vector<sigax_index*> idx(devs.size(), nullptr);
  idx[0] = _fmi->handle();
  auto drop_replicas = [&] {
    for (size_t k = 1; k < idx.size(); ++k)
      if (idx[k]) sigax_index_close(idx[k]);
  };
  for (size_t k = 1; k < devs.size(); ++k) {
    if (sigax_index_clone(_fmi->handle(), devs[k], &idx[k]) != SIGAX_OK) {
      _error = std::string("failed to replicate the index on GPU ") + std::to_string(devs[k]) + ": " + sigax_last_error();
      drop_replicas();
      return false;
    }
  }
  if (devs.size() > 1) pt.lap("index replicas");
  // device batches: as many reads as the free memory of a GPU takes with two batches in flight, at most 2^20, and no
  // more than an even share of the input; the reference's threads x batch-size is a lower bound
  // (asked of every replica: two batch objects per replica, and replicas that share a physical device -- SIGA_DEVICE_MAP
  // rehearsals -- share its memory; the smallest answer sizes the batches of all)
  uint32_t hint = 1u << 20;
  for (size_t k = 0; n > 0 && k < devs.size(); ++k) {
    uint32_t sharing = 0, h = 0;
    for (int d : devs) sharing += d == devs[k] ? 1u : 0u;
    if (sigax_batch_size_hint(idx[k], std::max(maxLen, 1u), (uint32_t)minOverlap, flags, 2 * sharing, &h) != SIGAX_OK) {
      _error = std::string("overlap failed: ") + sigax_last_error();
      drop_replicas();
      return false;
    }
    hint = k == 0 ? h : std::min(hint, h);
  }
  size_t per = std::min<size_t>(hint, 1u << 20);
  if (const char* env = getenv("SIGA_BATCH_READS")) per = std::min<size_t>(hint, std::max<size_t>(strtoull(env, nullptr, 10), 1));
  per = std::min(per, std::max<size_t>((n + devs.size() - 1) / devs.size(), 1));
  if (!getenv("SIGA_BATCH_READS")) per = std::max(per, std::min<size_t>(std::max<size_t>(threads, 1) * std::max<size_t>(batch, 1), hint));
  per = std::max<size_t>(per, 1);
  const size_t nbatch = (n + per - 1) / per;
  Pipeline pl;
  pl.out.resize(nbatch);
  auto worker = [&](size_t w) {
    sigax_index* ix = idx[w];
    sigax_batch* bt[2] = {nullptr, nullptr};
    void* st[2] = {nullptr, nullptr};
    size_t cur[2] = {0, 0};
    bool busy[2] = {false, false};
    std::vector<uint64_t> loffs[2];
    auto cleanup = [&] {
      for (int k = 0; k < 2; ++k) {
        if (bt[k]) sigax_batch_destroy(bt[k]);
        if (st[k]) sigax_stream_destroy(devs[w], st[k]);
      }
    };
    for (int k = 0; k < 2; ++k) {
      if (sigax_stream_create(devs[w], &st[k]) != SIGAX_OK || sigax_batch_create(ix, (uint32_t)per, 0, maxLen, &bt[k]) != SIGAX_OK) {
        pl.fail(std::string("overlap failed: ") + sigax_last_error());
        cleanup();
        return;
      }
      // with two runs in flight per device the finder of one overlaps the filter/extract of the other: one launch chain
      // per run is then faster than the library's sub-batches (bench.py: 112 vs 106 M reads/s at BASELINE configs[1])
      if (nbatch >= 2 * devs.size() && !getenv("SIGAX_SUBBATCHES")) sigax_batch_set_subbatches(bt[k], 1);
    }
    const bool timing = getenv("SIGA_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    auto submit = [&](int k) -> bool {
      const size_t b = pl.next.fetch_add(1);
      if (b >= nbatch || pl.failed) return false;
      const auto t0 = now();
      const size_t lo = b * per, cnt = std::min(per, n - lo);
      loffs[k].resize(cnt + 1);
      const uint64_t base = reads.offs[lo];
      for (size_t i = 0; i <= cnt; ++i) loffs[k][i] = reads.offs[lo + i] - base;
      int rc = sigax_batch_upload(bt[k], reads.seqs.data() + base, loffs[k].data(), (uint32_t)cnt, st[k]);
      if (rc == SIGAX_OK) rc = sigax_batch_run(bt[k], (uint32_t)lo, (uint32_t)minOverlap, flags, st[k]);
      if (rc != SIGAX_OK) {
        pl.fail(std::string("overlap failed: ") + sigax_last_error());
        return false;
      }
      cur[k] = b;
      busy[k] = true;
      if (timing) fprintf(stderr, "[siga]   gpu %d batch %zu: upload + enqueue %.3f s\n", devs[w], b, secs(t0, now()));
      return true;
    };
    auto collect = [&](int k) -> bool {
      BatchOut r;
      const size_t lo = cur[k] * per, cnt = std::min(per, n - lo);
      r.substring.resize(cnt);
      const auto t0 = now();
      int rc = sigax_batch_finish(bt[k], st[k], nullptr);
      const auto t1 = now();
      if (rc == SIGAX_OK) rc = sigax_batch_download_edges(bt[k], r.substring.data(), &r.edges, &r.n_edges);
      if (timing) fprintf(stderr, "[siga]   gpu %d batch %zu: wait %.3f s, download %.3f s\n", devs[w], cur[k], secs(t0, t1), secs(t1, now()));
      busy[k] = false;
      if (rc != SIGAX_OK) {
        pl.fail(std::string("overlap failed: ") + sigax_last_error());
        return false;
      }
      r.ready = true;
      {
        std::lock_guard<std::mutex> g(pl.mu);
        pl.out[cur[k]] = std::move(r);
      }
      pl.cv.notify_all();
      return true;
    };
    submit(0);
    submit(1);
    int k = 0;  // the older of the two runs
    while (busy[k] && !pl.failed) {
      if (!collect(k)) break;
      submit(k);
      k ^= 1;
    }
    cleanup();
  };
  std::vector<std::thread> workers;
  for (size_t w = 0; w < devs.size(); ++w) workers.emplace_back(worker, w);
  // ordered post-processing (OverlapPostProcess, src/overlap_builder.cpp:291-329): VT lines of batch b
  std::vector<std::pair<sigax_edge*, uint64_t>> edges;
  // The ED lines come after the last VT line, but their TEXT does not have to wait: while the GPUs work on the batches
  // that follow, the host threads have time, so a batch's edge records are formatted as soon as its VT lines are out
  // (up to 4 GiB of text held; beyond that the records wait and are formatted at the end, 16 bytes against ~45 each).
  std::vector<std::vector<std::string>> ed_text;
  size_t ed_held = 0;
  const char* env_hold = getenv("SIGA_ED_HOLD_BYTES");  // (tests: 0 = every batch's records wait for the end)
  const size_t ed_hold_max = env_hold ? (size_t)strtoull(env_hold, nullptr, 10) : (size_t)4 << 30, ed_chunk = 16384;
  const uint32_t* read_len = pre->lengths.data();
  uint32_t max_name = 0;
  for (uint32_t l : reads.name_len) max_name = std::max(max_name, l);
  auto format_edges = [&](const sigax_edge* e, uint64_t cnt, std::vector<std::string>* parts) {
    format_edge_text(reads, read_len, e, cnt, nt, max_name, ed_chunk, parts);
  };
  const size_t vt_chunk = 4096;
  edges.reserve(nbatch);    // (the edge-text job of a batch keeps its slots while the next batch is taken)
  ed_text.reserve(nbatch);
  std::vector<std::string> vt_parts;
  std::thread ed_job;
  auto join_ed = [&] {
    if (ed_job.joinable()) ed_job.join();
  };
  const bool ed_inline = getenv("SIGA_ED_INLINE") != nullptr;
  std::vector<uint8_t> sub_all;  // (VT lines ahead) the substring flags of the reads whose chunk is not out yet
  size_t ahead_from = 0;
  if (ahead) sub_all.assign(n, 0);
  for (size_t b = 0; b < nbatch; ++b) {
    BatchOut r;
    const auto tw0 = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> g(pl.mu);
      pl.cv.wait(g, [&] { return pl.out[b].ready || pl.failed; });
      if (!pl.out[b].ready) break;
      r = std::move(pl.out[b]);
      pl.out[b].edges = nullptr;  // owned by `edges` from here on
    }
    const size_t lo = b * per, cnt = std::min(per, n - lo);
    const auto tv0 = std::chrono::steady_clock::now();
    const double wait_s = std::chrono::duration<double>(tv0 - tw0).count();
    auto tv1 = tv0;
    if (ahead) {
      // every chunk of text whose reads are all back (the chunks do not know about batches)
      memcpy(sub_all.data() + lo, r.substring.data(), cnt);
      const size_t to = lo + cnt == n ? ahead->chunks() : (lo + cnt) / VtAhead::kChunk;
      if (to > ahead_from) {
        std::vector<std::string> parts;
        ahead->take(ahead_from, to, sub_all.data(), &parts);
        tv1 = std::chrono::steady_clock::now();
        out.write_parts(parts, ahead->blocks());
        ahead->taken(to);
        ahead_from = to;
      }
    } else {
      // (the strings keep their room from batch to batch: 165 MB of fresh 4 KB pages per batch otherwise)
      std::vector<std::string>& parts = vt_parts;
      parts.resize((cnt + vt_chunk - 1) / vt_chunk);
      parallel_for(parts.size(), nt, [&](size_t c) {
        const size_t cb = c * vt_chunk, ce = std::min(cnt, cb + vt_chunk);
        std::string& o = parts[c];
        o.clear();
        o.reserve((ce - cb) * (maxLen + 32));
        for (size_t i = cb; i < ce; ++i) write_vertex(o, reads.name(lo + i), reads.comment(lo + i), reads.seq(lo + i), r.substring[i] != 0);
      });
      tv1 = std::chrono::steady_clock::now();
      out.write_parts(parts);
    }
    if (pt.on) fprintf(stderr, "[siga]   batch %zu: VT text %.3f s, deflate + write %.3f s\n", b, std::chrono::duration<double>(tv1 - tv0).count(),
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - tv1).count());
    // the reference's progress line (OverlapPostProcess, src/overlap_builder.cpp:319-321: every threads x batch reads)
    if (pt.on) {
      const size_t stepn = std::max<size_t>(threads, 1) * std::max<size_t>(batch, 1);
      for (size_t at = (lo / stepn + 1) * stepn; at <= lo + cnt; at += stepn)
        if (at % (stepn * 64) == 0 || at + stepn > n) fprintf(stderr, "processed %zu sequences\n", at);  // (every 64th: a device batch is a million reads)
    }
    edges.emplace_back(r.edges, r.n_edges);
    ed_text.emplace_back();
    const auto tj0 = std::chrono::steady_clock::now();
    join_ed();  // (one batch's edge text at a time; ed_held is the job's to update, ours to read after the join)
    if (pt.on) fprintf(stderr, "[siga]   batch %zu: waited %.3f s for the batch, %.3f s for the ED text before it\n", b, wait_s,
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - tj0).count());
    if (ed_held < ed_hold_max) {
      const size_t k = edges.size() - 1;
      auto job = [&, k] {
        const auto te0 = std::chrono::steady_clock::now();
        format_edges(edges[k].first, edges[k].second, &ed_text[k]);
        if (pt.on) fprintf(stderr, "[siga]   batch %zu: ED text %.3f s\n", k, std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count());
        for (const std::string& p : ed_text[k]) ed_held += p.size();
        if (ed_text[k].empty()) ed_text[k].emplace_back();  // "formatted, and nothing to say"
        sigax_free(edges[k].first);
        edges[k].first = nullptr;
      };
      // beside the next batch's VT lines (SIGA_ED_INLINE=1: before them, on this thread)
      if (ed_inline) job();
      else ed_job = std::thread(job);
    }
  }
  join_ed();
  for (auto& t : workers) t.join();
  drop_replicas();
  auto free_edges = [&] {
    for (auto& e : edges) sigax_free(e.first);
    for (auto& o : pl.out)
      if (o.edges) sigax_free(o.edges);
  };
  if (pl.failed) {
    _error = pl.error;
    free_edges();
    return false;
  }
  pt.lap("GPU batches + VT lines");
  // ED lines in hits order (Hit2OverlapConverter, src/overlap_builder.cpp:345-375 + :474-483)
  for (size_t b = 0; b < edges.size(); ++b) {
    if (ed_text[b].empty()) format_edges(edges[b].first, edges[b].second, &ed_text[b]);
    out.write_parts(ed_text[b]);
    std::vector<std::string>().swap(ed_text[b]);
  }
  free_edges();
  if (!out.close()) {
    _error = "Failed to write ASQG " + output;
    return false;
  }
  pt.lap("ED lines + close");
  if (_keep_reads) _pre = pre;
  return true;
}

bool OverlapBuilder::rmdup(const std::string& input, const std::string& output, const std::string& duplicates, size_t threads,
                           size_t* processed) const {
  (void)processed;
  _error.clear();
  if (!_fmi || !_fmi->handle()) {
    _error = "FMIndex not loaded";
    return false;
  }
  // the chunk-parallel loader of build() (the reference reads record by record: src/overlap_builder.cpp:511-530); reads stay
  // packed, a piece of them goes to the device at a time
  const unsigned nt = (unsigned)std::max<size_t>(threads, 1);
  ReadStore reads;
  if (!LoadReads(input, &reads, nt)) {
    _error = "Failed to create DNASeqReader " + input;
    return false;
  }
  OutFile fasta(output), dups(duplicates);
  if (!fasta.ok() || !dups.ok()) {
    _error = "Failed to create FASTA " + output;
    return false;
  }
  const size_t n = reads.size();
  {
    std::vector<uint32_t> lengths, ranks;
    name_ranks(reads, nt, &lengths, &ranks);
    if (n > 0 && sigax_index_set_reads(_fmi->handle(), lengths.data(), ranks.data(), n) != SIGAX_OK) {
      _error = std::string("failed to load suffix array index: ") + sigax_last_error();
      return false;
    }
  }
  const size_t per = 1u << 20;
  std::string text;
  std::vector<uint64_t> offs;
  for (size_t base = 0; base < n; base += per) {
    const size_t cnt = std::min(per, n - base);
    offs.resize(cnt + 1);
    for (size_t i = 0; i <= cnt; ++i) offs[i] = reads.offs[base + i] - reads.offs[base];
    sigax_result res;
    if (sigax_overlap_batch(_fmi->handle(), reads.seqs.data() + reads.offs[base], offs.data(), (uint32_t)cnt, (uint32_t)base, 0,
                            SIGAX_DUPLICATE | SIGAX_EDGES, &res) != SIGAX_OK) {
      _error = std::string("rmdup failed: ") + sigax_last_error();
      return false;
    }
    // Hits2FastaConverter::convert (src/overlap_builder.cpp:578-616).  A kept overlap of a duplicate block is a
    // containment of the query with containedIdx() == 0 (both reads contained and id[0] > id[1], coord.h:185-194).
    std::vector<uint8_t> hasEdge(cnt, 0);
    for (uint64_t e = 0; e < res.n_edges; ++e) hasEdge[res.edges[e].query - base] = 1;
    for (size_t i = 0; i < cnt; ++i) {
      const std::string_view name = reads.name(base + i), seq = reads.seq(base + i);
      uint64_t numCopies = 0;
      for (uint64_t k = res.block_offs[i]; k < res.block_offs[i + 1]; ++k)
        numCopies += res.blocks[k].capped0_hi - res.blocks[k].capped0_lo + 1;
      bool contained = res.substring[i] != 0 || hasEdge[i] != 0;
      text.clear();
      text += '>';
      text.append(name.data(), name.size());
      if (contained) {
        text += ",seqrank=";
        append_u64(text, base + i);
      }
      text += ' ';
      text.append(name.data(), name.size());
      text += " NumDuplicates=";
      append_u64(text, numCopies);
      text += '\n';
      text.append(seq.data(), seq.size());
      text += '\n';
      (contained ? dups : fasta).write(text);
    }
    sigax_result_free(&res);
  }
  bool ok1 = fasta.close(), ok2 = dups.close();
  if (!ok1 || !ok2) {
    _error = "Failed to write rmdup output";
    return false;
  }
  return true;
}

bool CorrectProcessor::process(const FMIndex& index, const std::string& input, const std::string& output, size_t threads,
                               size_t* processed) const {
  (void)threads;
  (void)processed;
  _error.clear();
  if (!index.handle()) {
    _error = "FMIndex not loaded";
    return false;
  }
  DNASeqList reads;
  if (!ReadDNASequences(input, reads)) {
    _error = "Failed to create DNASeqReader " + input;
    return false;
  }
  OutFile out(output);
  if (!out.ok()) {
    _error = "Failed to create DNASeqWriter " + output;
    return false;
  }
  const size_t n = reads.size(), per = 262144;
  std::string seqs, quals, corrected, text;
  std::vector<uint64_t> offs;
  std::vector<uint8_t> valid;
  for (size_t base = 0; base < n; base += per) {
    size_t cnt = std::min(per, n - base);
    seqs.clear();
    quals.clear();
    offs.assign(1, 0);
    bool anyQual = false;
    for (size_t i = 0; i < cnt; ++i) anyQual = anyQual || !reads[base + i].quality.empty();
    for (size_t i = 0; i < cnt; ++i) {
      const DNASeq& rd = reads[base + i];
      seqs += rd.seq;
      if (anyQual) {  // a read without qualities scores 15 per base (src/kseq.h:34-40): '0' is phred 15
        if (rd.quality.empty()) quals.append(rd.seq.size(), (char)(15 + 33));
        else quals += rd.quality;
      }
      offs.push_back(seqs.size());
    }
    corrected.assign(seqs.size(), '\0');
    valid.assign(cnt, 0);
    if (sigax_correct_batch(index.handle(), seqs.data(), anyQual ? quals.data() : nullptr, offs.data(), (uint32_t)cnt,
                            (uint32_t)_options.kmerSize, (int32_t)_options.kmerThreshold, (uint32_t)_options.kmerRounds,
                            (uint32_t)_options.kmerCountOffset, &corrected[0], valid.data()) != SIGAX_OK) {
      _error = std::string("correct failed: ") + sigax_last_error();
      return false;
    }
    text.clear();
    for (size_t i = 0; i < cnt; ++i) {  // PostCorrector (src/correct_processor.cpp:247-253) + DNASeq << (src/kseq.cpp:106-126)
      if (valid[i] != 1) continue;
      const DNASeq& rd = reads[base + i];
      text += rd.quality.empty() ? '>' : '@';
      text += rd.name;
      if (!rd.comment.empty()) {
        text += ' ';
        text += rd.comment;
      }
      text += '\n';
      text.append(corrected, offs[i], offs[i + 1] - offs[i]);
      text += '\n';
      if (!rd.quality.empty()) {
        text += "+\n";
        text += rd.quality;
        text += '\n';
      }
    }
    out.write(text);
  }
  if (!out.close()) {
    _error = "Failed to write " + output;
    return false;
  }
  return true;
}

}  // namespace sigah

// ------------------------------------------------------------------------------------------------------
// C entry points for tests/bench (ctypes)
// ------------------------------------------------------------------------------------------------------
extern "C" {

// `siga index` for in-memory reads: writes <prefix>.{bwt,sai,rbwt,rsai}; returns 0 or -1 (message in err)
int sigah_index_build(const char* seqs, const uint64_t* offs, uint64_t n_reads, const char* prefix, int threads, char* err,
                      uint64_t errcap) {
  sigah::StrandIndex fwd, rev;
  std::string e1, e2;
  bool ok1 = false, ok2 = false;
  if (threads == 2) {  // one SA-IS per strand, side by side
    std::thread t([&] { ok2 = sigah::BuildStrandIndex(seqs, offs, n_reads, true, &rev, &e2); });
    ok1 = sigah::BuildStrandIndex(seqs, offs, n_reads, false, &fwd, &e1);
    t.join();
  } else {  // 1 thread: SA-IS; more: the multi-threaded bucket sort, one strand after the other
    ok1 = sigah::BuildStrandIndex(seqs, offs, n_reads, false, &fwd, &e1, (unsigned)std::max(threads, 1));
    ok2 = sigah::BuildStrandIndex(seqs, offs, n_reads, true, &rev, &e2, (unsigned)std::max(threads, 1));
  }
  std::string p(prefix);
  if (ok1 && ok2) {
    ok1 = fwd.writeSAI(p + ".sai") && fwd.writeBWT(p + ".bwt");
    ok2 = rev.writeSAI(p + ".rsai") && rev.writeBWT(p + ".rbwt");
    if (!ok1 || !ok2) e1 = "cannot write index files with prefix " + p;
  }
  if (!(ok1 && ok2)) {
    if (err && errcap) snprintf(err, errcap, "%s", (e1.empty() ? e2 : e1).c_str());
    return -1;
  }
  return 0;
}

// `siga index` on the GPU: both strands through sigax_build_strand; an input too repetitive for the device sort is
// built by the host SA-IS instead (said on stderr).  device < 0: host builder only.
int sigah_index_build_dev(const char* seqs, const uint64_t* offs, uint64_t n_reads, const char* prefix, int device, int threads,
                          int do_fwd, int do_rev, char* err, uint64_t errcap) {
  if (device < 0) {
    if (do_fwd && do_rev) return sigah_index_build(seqs, offs, n_reads, prefix, threads, err, errcap);
  }
  std::string p(prefix), e;
  // the files of one strand are written on a side thread while the other strand is sorted; the second sort works in the
  // device memory of the first
  struct Session {
    Session() { sigax_build_session(1); }
    ~Session() { sigax_build_session(0); }
  } session;
  std::thread writer;
  bool write_ok = true;
  auto join_writer = [&] {
    if (writer.joinable()) writer.join();
  };
  for (int rev = 0; rev < 2; ++rev) {
    if ((rev == 0 && !do_fwd) || (rev == 1 && !do_rev)) continue;
    auto ix = std::make_shared<sigah::StrandIndex>();
    int rc = 0;
    bool ok = device >= 0 && sigah::BuildStrandIndexGPU(seqs, offs, n_reads, rev != 0, device, ix.get(), &e, &rc);
    if (!ok && (device < 0 || rc == SIGAX_E_CAPACITY)) {
      if (device >= 0) fprintf(stderr, "siga index: %s; using the host suffix sorter\n", e.c_str());
      ok = sigah::BuildStrandIndex(seqs, offs, n_reads, rev != 0, ix.get(), &e, (unsigned)std::max(threads, 1));
    }
    if (!ok) {
      join_writer();
      if (err && errcap) snprintf(err, errcap, "%s", e.c_str());
      return -1;
    }
    join_writer();
    writer = std::thread([ix, p, rev, &write_ok] {  // the strand's two files side by side
      bool ok_sai = true;
      std::thread sai([&] { ok_sai = ix->writeSAI(p + (rev ? ".rsai" : ".sai")); });
      const bool ok_bwt = ix->writeBWT(p + (rev ? ".rbwt" : ".bwt"));
      sai.join();
      if (!(ok_sai && ok_bwt)) write_ok = false;
    });
  }
  join_writer();
  if (!write_ok) {
    if (err && errcap) snprintf(err, errcap, "cannot write index files with prefix %s", p.c_str());
    return -1;
  }
  return 0;
}

// `siga index READS` with the device builder (device < 0: host builder)
int sigah_index_file_dev(const char* reads_path, const char* prefix, int device, int threads, int do_fwd, int do_rev, char* err,
                         uint64_t errcap) {
  sigah::PhaseTimer pt;
  sigah::ReadStore rs;
  if (!sigah::LoadReads(reads_path, &rs, sigah::host_threads((size_t)std::max(threads, 1)))) {
    if (err && errcap) snprintf(err, errcap, "Failed to open input file %s", reads_path);
    return -1;
  }
  pt.lap("parse reads");
  int rc = sigah_index_build_dev(rs.seqs.data(), rs.offs.data(), rs.size(), prefix, device, threads, do_fwd, do_rev, err, errcap);
  pt.lap("suffix sort + index files");
  return rc;
}

// `siga index -a sais READS`: the SAISBuilder order (src/suffix_array_builder.cpp:31-172), host suffix sorter only
int sigah_index_file_sais(const char* reads_path, const char* prefix, int threads, int do_fwd, int do_rev, char* err, uint64_t errcap) {
  sigah::ReadStore rs;
  if (!sigah::LoadReads(reads_path, &rs, sigah::host_threads((size_t)std::max(threads, 1)))) {
    if (err && errcap) snprintf(err, errcap, "Failed to open input file %s", reads_path);
    return -1;
  }
  const std::string p(prefix);
  for (int rev = 0; rev < 2; ++rev) {
    if ((rev == 0 && !do_fwd) || (rev == 1 && !do_rev)) continue;
    sigah::StrandIndex ix;
    std::string e;
    if (!sigah::BuildStrandIndex(rs.seqs.data(), rs.offs.data(), rs.size(), rev != 0, &ix, &e, (unsigned)std::max(threads, 1), true)) {
      if (err && errcap) snprintf(err, errcap, "%s", e.c_str());
      return -1;
    }
    if (!(ix.writeSAI(p + (rev ? ".rsai" : ".sai")) && ix.writeBWT(p + (rev ? ".rbwt" : ".bwt")))) {
      if (err && errcap) snprintf(err, errcap, "cannot write index files with prefix %s", p.c_str());
      return -1;
    }
  }
  return 0;
}

// `siga index READS`
int sigah_index_file(const char* reads_path, const char* prefix, int threads, char* err, uint64_t errcap) {
  sigah::DNASeqList reads;
  if (!sigah::ReadDNASequences(reads_path, reads, 0)) {
    if (err && errcap) snprintf(err, errcap, "Failed to open input file %s", reads_path);
    return -1;
  }
  std::string seqs;
  std::vector<uint64_t> offs(1, 0);
  for (auto& r : reads) {
    seqs += r.seq;
    offs.push_back(seqs.size());
  }
  return sigah_index_build(seqs.data(), offs.data(), reads.size(), prefix, threads, err, errcap);
}

// `siga overlap`: FMIndex::load + OverlapBuilder::build, reads sharded over `gpus` GPUs starting at `device`
int sigah_overlap_file_gpus(const char* reads_path, const char* prefix, uint64_t min_overlap, const char* output, int irreducible,
                            int rc, uint64_t threads, uint64_t batch, int device, int gpus, char* err, uint64_t errcap) {
  sigah::FMIndex fmi;
  if (!sigah::FMIndex::load(prefix, fmi, device)) {
    if (err && errcap) snprintf(err, errcap, "Failed to load FMIndex from %s: %s", prefix, sigax_last_error());
    return -1;
  }
  sigah::OverlapBuilder builder(&fmi, prefix, irreducible != 0, rc != 0);
  builder.setGPUs(gpus);
  if (!builder.build(reads_path, min_overlap, output, threads, batch)) {
    if (err && errcap) snprintf(err, errcap, "%s", builder.error().c_str());
    return -1;
  }
  return 0;
}

// `siga overlap`: FMIndex::load + OverlapBuilder::build
int sigah_overlap_file(const char* reads_path, const char* prefix, uint64_t min_overlap, const char* output, int irreducible,
                       int rc, uint64_t threads, uint64_t batch, int device, char* err, uint64_t errcap) {
  sigah::FMIndex fmi;
  if (!sigah::FMIndex::load(prefix, fmi, device)) {
    if (err && errcap) snprintf(err, errcap, "Failed to load FMIndex from %s: %s", prefix, sigax_last_error());
    return -1;
  }
  sigah::OverlapBuilder builder(&fmi, prefix, irreducible != 0, rc != 0);
  if (!builder.build(reads_path, min_overlap, output, threads, batch)) {
    if (err && errcap) snprintf(err, errcap, "%s", builder.error().c_str());
    return -1;
  }
  return 0;
}

// `siga rmdup`: FMIndex::load + OverlapBuilder::rmdup
int sigah_rmdup_file(const char* reads_path, const char* prefix, const char* output, const char* duplicates, int device,
                     char* err, uint64_t errcap) {
  sigah::FMIndex fmi;
  if (!sigah::FMIndex::load(prefix, fmi, device)) {
    if (err && errcap) snprintf(err, errcap, "Failed to load FMIndex from %s: %s", prefix, sigax_last_error());
    return -1;
  }
  sigah::OverlapBuilder builder(&fmi, prefix);
  if (!builder.rmdup(reads_path, output, duplicates)) {
    if (err && errcap) snprintf(err, errcap, "%s", builder.error().c_str());
    return -1;
  }
  return 0;
}

// `siga correct`: FMIndex::load(prefix.bwt) + CorrectProcessor::process
int sigah_correct_file(const char* reads_path, const char* prefix, const char* output, uint64_t k, uint64_t threshold,
                       uint64_t rounds, uint64_t offset, int device, char* err, uint64_t errcap) {
  sigah::FMIndex fmi;
  if (!sigah::FMIndex::loadForward(prefix, fmi, device)) {
    if (err && errcap) snprintf(err, errcap, "Failed to load FMIndex from %s: %s", prefix, sigax_last_error());
    return -1;
  }
  sigah::CorrectProcessor::Options o;
  o.kmerSize = k; o.kmerThreshold = threshold; o.kmerRounds = rounds; o.kmerCountOffset = offset;
  sigah::CorrectProcessor proc(o);
  if (!proc.process(fmi, reads_path, output)) {
    if (err && errcap) snprintf(err, errcap, "%s", proc.error().c_str());
    return -1;
  }
  return 0;
}

// test hook: parse a reads file with the parallel loader (mode 0) or the record-at-a-time DNASeqReader (mode 1) and dump
// "name\tcomment\tseq\tquality\n" per read; mode 2: the parallel loader and the edge converter's read table, "rank\tlength\n"
// per read; returns the number of reads or -1
int64_t sigah_parse_file(const char* path, int mode, const char* out_path, int threads) {
  FILE* f = fopen(out_path, "wb");
  if (!f) return -1;
  int64_t n = -1;
  if (mode == 3) {  // parse only (timing aid): nothing written
    sigah::ReadStore rs;
    if (sigah::LoadReads(path, &rs, (unsigned)std::max(threads, 1))) n = (int64_t)rs.size();
  } else if (mode == 2) {
    sigah::ReadStore rs;
    if (sigah::LoadReads(path, &rs, (unsigned)std::max(threads, 1))) {
      std::vector<uint32_t> lengths, ranks;
      sigah::name_ranks(rs, (unsigned)std::max(threads, 1), &lengths, &ranks);
      n = (int64_t)rs.size();
      for (size_t i = 0; i < rs.size(); ++i) fprintf(f, "%u\t%u\n", ranks[i], lengths[i]);
    }
  } else if (mode == 0) {
    sigah::ReadStore rs;
    if (sigah::LoadReads(path, &rs, (unsigned)std::max(threads, 1))) {
      n = (int64_t)rs.size();
      for (size_t i = 0; i < rs.size(); ++i) {
        std::string line;
        line.append(rs.name(i)); line += '\t'; line.append(rs.comment(i)); line += '\t'; line.append(rs.seq(i)); line += '\t';
        line.append(rs.quality(i)); line += '\n';
        fwrite(line.data(), 1, line.size(), f);
      }
    }
  } else {
    sigah::DNASeqList reads;
    if (sigah::ReadDNASequences(path, reads)) {
      n = (int64_t)reads.size();
      for (auto& r : reads) {
        std::string line = r.name + "\t" + r.comment + "\t" + r.seq + "\t" + r.quality + "\n";
        fwrite(line.data(), 1, line.size(), f);
      }
    }
  }
  fclose(f);
  return n;
}

// Test hook (CPU tests, sanitizer builds): the text side of OverlapBuilder::build without a GPU -- the reads of `path` through
// the loader, VT lines (substring flags given) and the ED lines of the given edge records through the same formatters and
// the same output stream (gz by name) build() uses.  Returns the number of reads, -1 on failure.
int64_t sigah_format_asqg(const char* path, const uint8_t* substring, const sigax_edge* edges, uint64_t n_edges, uint64_t min_overlap,
                          const char* out_path, int threads) {
  const unsigned nt = (unsigned)std::max(threads, 1);
  sigah::ReadStore rs;
  if (!sigah::LoadReads(path, &rs, nt)) return -1;
  std::vector<uint32_t> lengths, ranks;
  sigah::name_ranks(rs, nt, &lengths, &ranks);
  for (uint64_t i = 0; i < n_edges; ++i)
    if (edges[i].query >= rs.size() || edges[i].target >= rs.size()) return -1;
  sigah::OutFile out(out_path, nt);
  if (!out.ok()) return -1;
  const std::string header = sigah::asqg_header((size_t)min_overlap);
  out.write(header);
  uint32_t maxLen = 0, max_name = 0;
  for (uint32_t l : lengths) maxLen = std::max(maxLen, l);
  for (uint32_t l : rs.name_len) max_name = std::max(max_name, l);
  if (sigah::vt_ahead_wanted() && rs.size() > 0) {
    // the way build() takes them: in pieces, as if batches of SIGA_BATCH_READS reads came back one by one
    sigah::VtAhead ahead(nullptr, &rs, nt, header, out.gz());
    const char* env = getenv("SIGA_BATCH_READS");
    const size_t per = env ? std::max<size_t>(strtoull(env, nullptr, 10), 1) : rs.size();
    size_t from = 0;
    for (size_t hi = std::min(per, rs.size());; hi = std::min(hi + per, rs.size())) {
      const size_t to = hi == rs.size() ? ahead.chunks() : hi / sigah::VtAhead::kChunk;
      if (to > from) {
        std::vector<std::string> parts;
        ahead.take(from, to, substring, &parts);
        out.write_parts(parts, ahead.blocks());
        ahead.taken(to);
        from = to;
      }
      if (hi == rs.size()) break;
    }
  } else {
    const size_t vt_chunk = 4096;
    std::vector<std::string> parts((rs.size() + vt_chunk - 1) / vt_chunk);
    sigah::parallel_for(parts.size(), nt, [&](size_t c) {
      const size_t cb = c * vt_chunk, ce = std::min(rs.size(), cb + vt_chunk);
      std::string& o = parts[c];
      o.reserve((ce - cb) * (maxLen + 32));
      for (size_t i = cb; i < ce; ++i) sigah::write_vertex(o, rs.name(i), rs.comment(i), rs.seq(i), substring && substring[i] != 0);
    });
    out.write_parts(parts);
  }
  std::vector<std::string> ed;
  sigah::format_edge_text(rs, lengths.data(), edges, n_edges, nt, max_name, 1000, &ed);
  out.write_parts(ed);
  return out.close() ? (int64_t)rs.size() : -1;
}

// Utils::ofstream as used for <prefix>.asqg.gz: write `n` bytes in `pieces` write() calls (gz when the name ends with .gz)
int sigah_write_file(const char* path, const char* data, uint64_t n, uint64_t pieces) {
  sigah::OutFile out(path);
  if (!out.ok()) return -1;
  uint64_t step = pieces ? (n + pieces - 1) / pieces : n;
  for (uint64_t b = 0; b < n; b += step ? step : 1) out.write(data + b, (size_t)std::min<uint64_t>(step, n - b));
  return out.close() ? 0 : -1;
}

void sigah_stem(const char* path, char* out, uint64_t cap) { snprintf(out, cap, "%s", sigah::Utils::stem(path).c_str()); }

}  // extern "C"
