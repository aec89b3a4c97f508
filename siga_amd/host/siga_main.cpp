// siga_amd/host/siga_main.cpp -- `siga index` / `siga overlap` command line, option for option as the reference
// (src/main.cpp:17-83, src/indexer.cpp:119-156, src/overlap.cpp:66-105).  Exit codes follow the reference:
// a runner returning -1 exits 255; printing help returns 256, i.e. exit status 0.
#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <thread>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "siga_host.hpp"

extern "C" int sigah_index_file_dev(const char*, const char*, int, int, int, int, char*, uint64_t);
extern "C" int sigah_index_file_sais(const char*, const char*, int, int, int, char*, uint64_t);

// -s, --ini=FILE (src/main.cpp:62-77): boost::property_tree::read_ini fills the option tree, then the command line's options
// are put over it.  Here: the file's top-level `key=value` lines (`;` comments; keys under a [section] have dotted names no
// option carries) become "--key=value" / "--key" arguments in front of the command line's own, for the keys this
// sub-command's option table knows, so that what the command line says wins.  Returns 1 (the reference's exit status) when
// the file cannot be read or a line has no '='.
static int apply_ini(int argc, char** argv, const option* longopts, std::vector<std::string>* store, std::vector<char*>* out) {
  std::string path;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--") break;
    if ((a == "-s" || a == "--ini") && i + 1 < argc) path = argv[i + 1];
    else if (a.compare(0, 6, "--ini=") == 0) path = a.substr(6);
    else if (a.size() > 2 && a[0] == '-' && a[1] == 's') path = a.substr(2);
  }
  out->assign(argv, argv + argc);
  if (path.empty()) return 0;
  FILE* f = fopen(path.c_str(), "r");
  if (!f) {
    fprintf(stderr, "load %s failed(cannot open file).\n", path.c_str());
    return 1;
  }
  auto trim = [](std::string x) {
    const char* ws = " \t\r\n";
    const size_t b = x.find_first_not_of(ws);
    if (b == std::string::npos) return std::string();
    return x.substr(b, x.find_last_not_of(ws) - b + 1);
  };
  char line[4096];
  bool in_section = false;
  int rc = 0;
  while (fgets(line, sizeof(line), f)) {
    const std::string l = trim(line);
    if (l.empty() || l[0] == ';') continue;
    if (l[0] == '[') {
      in_section = true;
      continue;
    }
    const size_t eq = l.find('=');
    if (eq == std::string::npos) {
      fprintf(stderr, "load %s failed('=' character not found in line).\n", path.c_str());
      rc = 1;
      break;
    }
    if (in_section) continue;
    const std::string key = trim(l.substr(0, eq)), val = trim(l.substr(eq + 1));
    for (const option* o = longopts; o->name; ++o)
      if (key == o->name && key != "ini") store->push_back(o->has_arg == no_argument ? "--" + key : "--" + key + "=" + val);
  }
  fclose(f);
  if (rc) return rc;
  out->clear();
  out->push_back(argv[0]);
  for (std::string& x : *store) out->push_back(&x[0]);
  for (int i = 1; i < argc; ++i) out->push_back(argv[i]);
  return 0;
}

static int usage() {
  printf("siga [index|correct|overlap|rmdup] [OPTION] ... READSFILE\n"
         "  index     build the FM-index (.sai/.bwt/.rsai/.rbwt) of READSFILE\n"
         "  overlap   compute pairwise overlaps between all the sequences in READSFILE (GPU)\n"
         "  rmdup     remove duplicated reads (GPU)\n"
         "  correct   k-mer based error correction (GPU)\n"
         "common options: -s, --ini=FILE (options from FILE, the command line goes over them);\n"
         "                -c, --log4cxx=FILE is accepted and ignored (this build logs to stderr; SIGA_TIMING=1 prints phase\n"
         "                times and the reference's \"processed N sequences\" progress lines)\n");
  return 256;
}

static int index_help() {
  printf("siga index [OPTION] ... READSFILE\n"
         "Index the reads in READSFILE using a suffixarray/bwt\n"
         "\n"
         "      -h, --help                       display this help and exit\n"
         "\n"
         "      -a, --algorithm=STR              BWT construction algorithm. STR can be:\n"
         "                                       sais - induced sort algorithm's suffix order (host sorter, ACGT-only reads)\n"
         "                                       sais2 - very fast and works for very long sequences (default; GPU sorter)\n"
         "      -t, --threads=NUM                use NUM threads to construct the index (default: 1)\n"
         "      -p, --prefix=PREFIX              write index to file using PREFIX instead of prefix of READSFILE\n"
         "          --no-reverse                 suppress construction of the reverse BWT\n"
         "          --no-forward                 suppress construction of the forward BWT\n"
         "          --device=NUM                 GPU that sorts the suffixes (default: 0)\n"
         "          --cpu                        sort on the host (-t threads) instead; also taken when no GPU is visible\n"
         "\n");
  return 256;
}

static int overlap_help() {
  // help text of src/overlap.cpp:66-82 (the defaults it prints differ from the code defaults 10 / 10000, :44)
  printf("siga overlap [OPTION] ... READSFILE\n"
         "Compute pairwise overlap between all the sequences in READS\n"
         "\n"
         "      -h, --help                       display this help and exit\n"
         "\n"
         "      -t, --threads=NUM                use NUM threads to construct the index (default: 1)\n"
         "          --batch-size=NUM             use NUM batches for each thread (default: 1000)\n"
         "      -m, --min-overlap=LEN            minimum overlap required between two reads (default: 45)\n"
         "      -p, --prefix=PREFIX              write index to file using PREFIX instead of prefix of READSFILE\n"
         "      -x, --exhaustive                 output all overlaps, including transitive edges\n"
         "          --no-opposite-strand         treat all reads as forward strand\n"
         "          --device=NUM                 first GPU to use (default: 0)\n"
         "          --gpus=NUM                   shard the reads over NUM GPUs of this node, index replicated on each (default: 1)\n"
         "\n");
  return 256;
}

static int run_index(int argc, char** argv) {
  enum { OPT_NO_REVERSE = 1, OPT_NO_FORWARD, OPT_DEVICE, OPT_CPU };
  static const option longopts[] = {{"log4cxx", required_argument, nullptr, 'c'},  {"ini", required_argument, nullptr, 's'},
                                    {"prefix", required_argument, nullptr, 'p'},   {"algorithm", required_argument, nullptr, 'a'},
                                    {"threads", required_argument, nullptr, 't'},  {"no-reverse", no_argument, nullptr, OPT_NO_REVERSE},
                                    {"no-forward", no_argument, nullptr, OPT_NO_FORWARD}, {"device", required_argument, nullptr, OPT_DEVICE},
                                    {"cpu", no_argument, nullptr, OPT_CPU},        {"help", no_argument, nullptr, 'h'},
                                    {nullptr, 0, nullptr, 0}};
  std::string prefix, algorithm = "sais2";
  int threads = 1, c, device = 0;
  bool help = false, nofwd = false, norev = false, cpu = false;
  std::vector<std::string> ini_store;
  std::vector<char*> ini_argv;
  if (apply_ini(argc, argv, longopts, &ini_store, &ini_argv) != 0) return 1;
  argc = (int)ini_argv.size();
  argv = ini_argv.data();
  while ((c = getopt_long(argc, argv, "c:s:a:t:p:h", longopts, nullptr)) != -1) {
    switch (c) {
      case 'p': prefix = optarg; break;
      case 'a': algorithm = optarg; break;
      case 't': threads = atoi(optarg); break;
      case OPT_NO_REVERSE: norev = true; break;
      case OPT_NO_FORWARD: nofwd = true; break;
      case OPT_DEVICE: device = atoi(optarg); break;
      case OPT_CPU: cpu = true; break;
      case 'h': help = true; break;
      default: break;
    }
  }
  if (help || argc - optind != 1) return index_help();
  std::string input = argv[optind];
  if (prefix.empty()) prefix = sigah::Utils::stem(input);
  // src/suffix_array_builder.cpp:684-692: "sais" and "sais2" (case-insensitive); anything else fails to create a builder
  for (char& ch : algorithm) ch = (char)tolower((unsigned char)ch);
  if (algorithm == "sais") {  // SAISBuilder's order (every read's own sentinel, by read index): the host sorter only
    char err[512] = "";
    if (sigah_index_file_sais(input.c_str(), prefix.c_str(), threads, nofwd ? 0 : 1, norev ? 0 : 1, err, sizeof(err)) != 0) {
      fprintf(stderr, "%s\n", err);
      return -1;
    }
    return 0;
  }
  if (algorithm != "sais2") {
    fprintf(stderr, "Failed to create suffix array builder algorithm %s\n", algorithm.c_str());
    return -1;
  }
  if (!cpu) {
    int ndev = 0;
    if (sigax_device_count(&ndev) != SIGAX_OK || ndev <= 0) cpu = true;  // `siga index` also runs on a host without a GPU
  }
  char err[512] = "";
  if (sigah_index_file_dev(input.c_str(), prefix.c_str(), cpu ? -1 : device, threads, nofwd ? 0 : 1, norev ? 0 : 1, err, sizeof(err)) != 0) {
    fprintf(stderr, "%s\n", err);
    return -1;
  }
  return 0;
}

static int run_overlap(int argc, char** argv) {
  enum { OPT_BATCH_SIZE = 1, OPT_NO_RC, OPT_DEVICE, OPT_GPUS };
  static const option longopts[] = {{"log4cxx", required_argument, nullptr, 'c'},     {"ini", required_argument, nullptr, 's'},
                                    {"prefix", required_argument, nullptr, 'p'},      {"threads", required_argument, nullptr, 't'},
                                    {"batch-size", required_argument, nullptr, OPT_BATCH_SIZE},
                                    {"min-overlap", required_argument, nullptr, 'm'}, {"exhaustive", no_argument, nullptr, 'x'},
                                    {"no-opposite-strand", no_argument, nullptr, OPT_NO_RC},
                                    {"device", required_argument, nullptr, OPT_DEVICE}, {"gpus", required_argument, nullptr, OPT_GPUS},
                                    {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
  std::string prefix;
  size_t threads = 1, batch = 10000, minOverlap = 10;  // code defaults of src/overlap.cpp:44
  bool exhaustive = false, norc = false, help = false;
  int device = 0, gpus = 1, c;
  std::vector<std::string> ini_store;
  std::vector<char*> ini_argv;
  if (apply_ini(argc, argv, longopts, &ini_store, &ini_argv) != 0) return 1;
  argc = (int)ini_argv.size();
  argv = ini_argv.data();
  while ((c = getopt_long(argc, argv, "c:s:t:p:m:xh", longopts, nullptr)) != -1) {
    switch (c) {
      case 'p': prefix = optarg; break;
      case 't': threads = strtoull(optarg, nullptr, 10); break;
      case 'm': minOverlap = strtoull(optarg, nullptr, 10); break;
      case 'x': exhaustive = true; break;
      case OPT_BATCH_SIZE: batch = strtoull(optarg, nullptr, 10); break;
      case OPT_NO_RC: norc = true; break;
      case OPT_DEVICE: device = atoi(optarg); break;
      case OPT_GPUS: gpus = atoi(optarg); break;
      case 'h': help = true; break;
      default: break;
    }
  }
  if (help || argc - optind != 1) return overlap_help();
  std::string input = argv[optind];
  if (prefix.empty()) prefix = sigah::Utils::stem(input);
  // (not destroyed: the process ends when this command returns, and handing tens of GB of tables back to the driver one
  // hipFree at a time was 0.2 s of a 2 s run at BASELINE configs[2])
  sigah::FMIndex& fmi = *new sigah::FMIndex;
  sigah::OverlapBuilder& builder = *new sigah::OverlapBuilder(&fmi, prefix, !exhaustive, !norc);
  builder.setGPUs(gpus);
  builder.keepReads(true);
  const auto t_load = std::chrono::steady_clock::now();
  // the index goes to the GPU while the host threads parse the reads
  bool loaded = false;
  std::string load_error;
  std::thread loader([&] {
    loaded = sigah::FMIndex::load(prefix, fmi, device);
    if (!loaded) load_error = sigax_last_error();
  });
  builder.preload(input, threads, (long)minOverlap, prefix + ".asqg.gz");
  loader.join();
  if (!loaded) {
    fprintf(stderr, "Failed to load FMIndex from %s: %s\n", input.c_str(), load_error.c_str());
    return -1;
  }
  if (getenv("SIGA_TIMING"))
    fprintf(stderr, "[siga] %-28s %8.3f s\n", "FMIndex::load || parse", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count());
  if (!builder.build(input, minOverlap, prefix + ".asqg.gz", threads, batch)) {
    fprintf(stderr, "Failed to build overlaps from reads %s: %s\n", input.c_str(), builder.error().c_str());
    return -1;
  }
  return 0;
}

static int rmdup_help() {
  printf("siga rmdup [OPTION] ... READSFILE\n"
         "Remove duplicated reads from the data set\n"
         "\n"
         "      -h, --help                       display this help and exit\n"
         "      -p, --prefix=PREFIX              use PREFIX instead of the prefix of the reads filename for the input/output files\n"
         "      -t, --threads=N                  use N threads (default: 1)\n"
         "          --device=NUM                 GPU to use (default: 0)\n"
         "\n");
  return 256;
}

// src/rmdup.cpp:22-49: outputs <prefix>.rmdup.fa and <prefix>.rmdup.dups.fa
static int run_rmdup(int argc, char** argv) {
  enum { OPT_DEVICE = 1 };
  static const option longopts[] = {{"log4cxx", required_argument, nullptr, 'c'}, {"ini", required_argument, nullptr, 's'},
                                    {"prefix", required_argument, nullptr, 'p'},  {"threads", required_argument, nullptr, 't'},
                                    {"sample-rate", required_argument, nullptr, 'd'}, {"device", required_argument, nullptr, OPT_DEVICE},
                                    {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
  std::string prefix;
  size_t threads = 1;
  bool help = false;
  int device = 0, c;
  std::vector<std::string> ini_store;
  std::vector<char*> ini_argv;
  if (apply_ini(argc, argv, longopts, &ini_store, &ini_argv) != 0) return 1;
  argc = (int)ini_argv.size();
  argv = ini_argv.data();
  while ((c = getopt_long(argc, argv, "c:s:t:p:d:h", longopts, nullptr)) != -1) {
    switch (c) {
      case 'p': prefix = optarg; break;
      case 't': threads = strtoull(optarg, nullptr, 10); break;
      case OPT_DEVICE: device = atoi(optarg); break;
      case 'h': help = true; break;
      default: break;
    }
  }
  if (help || argc - optind != 1) return rmdup_help();
  std::string input = argv[optind];
  if (prefix.empty()) prefix = sigah::Utils::stem(input);
  sigah::FMIndex fmi;
  if (!sigah::FMIndex::load(prefix, fmi, device)) {
    fprintf(stderr, "Failed to load FMIndex from %s: %s\n", input.c_str(), sigax_last_error());
    return -1;
  }
  sigah::OverlapBuilder builder(&fmi, prefix);
  if (!builder.rmdup(input, prefix + ".rmdup.fa", prefix + ".rmdup.dups.fa", threads)) {
    fprintf(stderr, "Failed to remove duplicates from reads %s: %s\n", input.c_str(), builder.error().c_str());
    return -1;
  }
  return 0;
}

static int correct_help() {
  printf("siga correct [OPTION] ... READSFILE\n"
         "Correct sequencing errors in all the reads in READSFILE\n"
         "\n"
         "      -h, --help                       display this help and exit\n"
         "\n"
         "      -p, --prefix=PREFIX              use PREFIX instead of prefix of READSFILE for the names of the index files\n"
         "      -o, --outfile=FILE               write the corrected reads to FILE (default READFILE.ec.fa)\n"
         "      -t, --threads=NUM                use NUM threads for the computation (default: 1)\n"
         "      -a, --algorithm=STR              specify the correction algorithm to use. Only kmer is built. (default: kmer)\n"
         "\n"
         "      -k, --kmer-size=N                the length of the kmer to user (default: 31)\n"
         "      -x, --kmer-threshold=N           attempt to correct kmers that are seen less than N times (default: 3)\n"
         "      -i, --kmer-rounds=N              perform up to N rounds of kmer correction (default: 10)\n"
         "      -O, --kmer-count-offset=N        when correcting a kmer, require the count of the new kmer is at least +N higher than the count of the old kmer. (default: 1)\n"
         "          --device=NUM                 GPU to use (default: 0)\n"
         "\n");
  return 256;
}

// src/correct.cpp:22-60
static int run_correct(int argc, char** argv) {
  enum { OPT_DEVICE = 1 };
  static const option longopts[] = {{"log4cxx", required_argument, nullptr, 'c'},   {"ini", required_argument, nullptr, 's'},
                                    {"prefix", required_argument, nullptr, 'p'},    {"outfile", required_argument, nullptr, 'o'},
                                    {"threads", required_argument, nullptr, 't'},   {"algorithm", required_argument, nullptr, 'a'},
                                    {"kmer-size", required_argument, nullptr, 'k'}, {"kmer-threshold", required_argument, nullptr, 'x'},
                                    {"kmer-rounds", required_argument, nullptr, 'i'}, {"kmer-count-offset", required_argument, nullptr, 'O'},
                                    {"device", required_argument, nullptr, OPT_DEVICE}, {"help", no_argument, nullptr, 'h'},
                                    {nullptr, 0, nullptr, 0}};
  std::string prefix, outfile, algorithm = "kmer";
  sigah::CorrectProcessor::Options o;
  size_t threads = 1;
  bool help = false;
  int device = 0, c;
  std::vector<std::string> ini_store;
  std::vector<char*> ini_argv;
  if (apply_ini(argc, argv, longopts, &ini_store, &ini_argv) != 0) return 1;
  argc = (int)ini_argv.size();
  argv = ini_argv.data();
  while ((c = getopt_long(argc, argv, "c:s:p:o:t:a:k:x:i:O:h", longopts, nullptr)) != -1) {
    switch (c) {
      case 'p': prefix = optarg; break;
      case 'o': outfile = optarg; break;
      case 't': threads = strtoull(optarg, nullptr, 10); break;
      case 'a': algorithm = optarg; break;
      case 'k': o.kmerSize = strtoull(optarg, nullptr, 10); break;
      case 'x': o.kmerThreshold = strtoull(optarg, nullptr, 10); break;
      case 'i': o.kmerRounds = strtoull(optarg, nullptr, 10); break;
      case 'O': o.kmerCountOffset = strtoull(optarg, nullptr, 10); break;
      case OPT_DEVICE: device = atoi(optarg); break;
      case 'h': help = true; break;
      default: break;
    }
  }
  if (help || argc - optind != 1) return correct_help();
  std::string input = argv[optind];
  std::string stem = sigah::Utils::stem(input);
  if (outfile.empty()) outfile = stem + ".ec.fa";
  if (prefix.empty()) prefix = stem;
  if (algorithm != "kmer") {  // AbstractCorrector::create returns NULL / the overlap corrector is an empty stub
    fprintf(stderr, "Failed to do error correction for reads %s: algorithm %s is not built\n", input.c_str(), algorithm.c_str());
    return -1;
  }
  sigah::FMIndex fmi;
  if (!sigah::FMIndex::loadForward(prefix, fmi, device)) {
    fprintf(stderr, "Failed to load FMIndex from %s: %s\n", prefix.c_str(), sigax_last_error());
    return -1;
  }
  sigah::CorrectProcessor proc(o);
  if (!proc.process(fmi, input, outfile, threads)) {
    fprintf(stderr, "Failed to do error correction for reads %s: %s\n", input.c_str(), proc.error().c_str());
    return -1;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return usage();
  const auto t0 = std::chrono::steady_clock::now();
  std::string cmd = argv[1];
  int rc = 256;
  if (cmd == "index") rc = run_index(argc - 1, argv + 1);
  else if (cmd == "rmdup") rc = run_rmdup(argc - 1, argv + 1);
  else if (cmd == "correct") rc = run_correct(argc - 1, argv + 1);
  else if (cmd == "overlap") rc = run_overlap(argc - 1, argv + 1);
  else return usage();
  if (getenv("SIGA_TIMING"))
    fprintf(stderr, "[siga] %-28s %8.3f s\n", "main() total", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  // every output file is closed by now: leave without tearing the HIP runtime down (tens of milliseconds of nothing)
  // (SIGA_CLEAN_EXIT=1: the ordinary way out, for profilers that write their report from an exit handler)
  fflush(nullptr);
  if (getenv("SIGA_CLEAN_EXIT")) return rc & 0xFF;
  _exit(rc & 0xFF);
}
