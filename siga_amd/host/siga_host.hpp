// siga_amd/host/siga_host.hpp -- host side of the drop-in: the reference's classes for this path, re-implemented
// over the C-ABI (include/sigax.h).  Same names, argument meaning and error behaviour as the reference so that a
// maintainer can swap them in: DNASeq / readers (src/kseq.h), FMIndex (src/fmindex.h, here a handle pair living
// on the GPU), SuffixArray + BWT writers (src/suffix_array.cpp, src/bwt.cpp), OverlapBuilder
// (src/overlap_builder.h:19-45), Utils::stem (src/utils.cpp:128-135).
#ifndef SIGA_AMD_HOST_SIGA_HOST_HPP_
#define SIGA_AMD_HOST_SIGA_HOST_HPP_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/sigax.h"

namespace sigah {

// src/kseq.h: DNASeq
struct DNASeq {
  std::string name, comment, seq, quality;
};
typedef std::vector<DNASeq> DNASeqList;

// gz-aware line source (Utils::ifstream, src/utils.cpp:50-90; bz2 is not supported by this build)
class LineSource;

// DNASeqReader / FASTAReader / FASTQReader (src/kseq.h:60-150, src/kseq.cpp:127-228)
class DNASeqReader {
 public:
  static DNASeqReader* create(const std::string& path);  // DNASeqReaderFactory::create; nullptr on failure
  ~DNASeqReader();
  bool read(DNASeq& sequence);
  bool failed() const;  // the input could not be read to its end (corrupt .gz); read() then returned false early
  void reset();

 private:
  DNASeqReader();
  std::unique_ptr<LineSource> _src;
  bool _fastq;
  std::string _name;
};

enum { kSeqWithQuality = 1, kSeqWithComment = 2 };
bool ReadDNASequences(const std::string& file, DNASeqList& sequences, uint32_t flags = kSeqWithQuality | kSeqWithComment);

namespace Utils {
std::string stem(const std::string& filename);
}

// One strand's index as `siga index` writes it (src/indexer.cpp:80-104): RL-BWT + the full-read SA rows.
struct StrandIndex {
  std::vector<uint8_t> runs;   // RLUnit bytes (src/rlstring.h:10-63), 31-cap (src/bwt.cpp:17)
  std::vector<uint32_t> sai;   // read ids of the j==0 suffixes in SA order (src/suffix_array.cpp:17-44)
  uint64_t nStrings, nSymbols;
  bool writeBWT(const std::string& path) const;  // src/bwt.cpp:121-178
  bool writeSAI(const std::string& path) const;  // src/suffix_array.cpp:17-44
};

// SuffixArrayBuilder "sais2" + BWT(sa, reads) (src/suffix_array_builder.cpp:472-674, src/bwt.cpp:7-32)
// threads >= 2 selects a multi-threaded bucket sort (same suffix order), 1 the SA-IS
// own_sentinels: the suffix order of `siga index -a sais` (every read's own '$', ordered by read index) instead of the
// default "sais2" order (one shared '$', comparisons running on into the next read); host only, ACGT-only reads
bool BuildStrandIndex(const char* seqs, const uint64_t* offs, uint64_t nReads, bool reverse, StrandIndex* out,
                      std::string* error, unsigned threads = 1, bool own_sentinels = false);

// the same on the GPU (sigax_build_strand); *rc = the library's code (SIGAX_E_CAPACITY: use BuildStrandIndex)
bool BuildStrandIndexGPU(const char* seqs, const uint64_t* offs, uint64_t nReads, bool reverse, int device, StrandIndex* out,
                         std::string* error, int* rc = nullptr);

// FMIndex pair resident on a GPU (FMIndex::load x2, src/overlap.cpp:41-42)
class FMIndex {
 public:
  FMIndex() : _h(nullptr) {}
  ~FMIndex();
  static bool load(const std::string& prefix, FMIndex& fmi, int device = 0);
  static bool loadForward(const std::string& prefix, FMIndex& fmi, int device = 0);  // <prefix>.bwt alone (`siga correct`)
  sigax_index* handle() const { return _h; }
  uint64_t length() const;

 private:
  FMIndex(const FMIndex&);
  FMIndex& operator=(const FMIndex&);
  sigax_index* _h;
};

// src/overlap_builder.h:19-45.  The reference takes (fmi, rfmi); here one FMIndex object holds both strands.
class OverlapBuilder {
 public:
  OverlapBuilder(const FMIndex* fmi, const std::string& prefix = "default", bool irreducible = true, bool rc = true)
      : _fmi(fmi), _prefix(prefix), _irreducible(irreducible), _rc(rc), _gpus(1) {}
  // reads shard over `n` GPUs of the node, starting at the index's device; the index is replicated device to device
  void setGPUs(int n) { _gpus = n < 1 ? 1 : n; }
  // build() leaves the parsed reads with the builder instead of unmapping them (a CLI about to exit: its pages go with the process)
  void keepReads(bool on) { _keep_reads = on; }
  // parse `input` and rank its names now (host threads only), e.g. while FMIndex::load is busy on another thread;
  // the next build() of the same file uses the result
  void preload(const std::string& input, size_t threads = 1, long minOverlap = -1, const std::string& output = std::string()) const;

  // HT, VT (input order), ED (hits order) to `output` (gz when the name ends with .gz).  `threads` is accepted for
  // signature compatibility (the GPU replaces the OpenMP loop); `batch` = reads per device batch.
  bool build(const std::string& input, size_t minOverlap, const std::string& output, size_t threads = 1,
             size_t batch = 1000, size_t* processed = nullptr) const;

  // `siga rmdup` (src/overlap_builder.cpp:562-704): reads without an identical / reverse-complement-identical twin of
  // smaller name and that are no substring go to `output`, the others to `duplicates`, headers as the reference
  // writes them ("<name> <name> NumDuplicates=<n>" / "<name>,seqrank=<idx> <name> NumDuplicates=<n>").
  bool rmdup(const std::string& input, const std::string& output, const std::string& duplicates, size_t threads = 1,
             size_t* processed = nullptr) const;

  const std::string& error() const { return _error; }

 private:
  const FMIndex* _fmi;
  std::string _prefix;
  bool _irreducible;
  bool _rc;
  int _gpus;
  mutable std::string _error;
  struct Preloaded;
  mutable std::shared_ptr<Preloaded> _pre;
  bool _keep_reads = false;
};

// src/correct_processor.h:25-42 (k-mer algorithm only; the reference's "overlap" algorithm is an empty stub)
class CorrectProcessor {
 public:
  struct Options {
    size_t kmerSize, kmerThreshold, kmerRounds, kmerCountOffset;  // -k -x -i -O, defaults src/correct_processor.h:15-20
    Options() : kmerSize(31), kmerThreshold(3), kmerRounds(10), kmerCountOffset(1) {}
  };
  explicit CorrectProcessor(const Options& options) : _options(options) {}
  // reads that became all-solid are written to `output` in the input's format (FASTA/FASTQ), the rest are dropped
  bool process(const FMIndex& index, const std::string& input, const std::string& output, size_t threads = 1,
               size_t* processed = nullptr) const;
  const std::string& error() const { return _error; }

 private:
  Options _options;
  mutable std::string _error;
};

}  // namespace sigah

#endif
