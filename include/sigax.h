/* include/sigax.h -- C-ABI of the MI355X-native `siga overlap` hot path.
 *
 * The reference (chungongyu/siga) has no FFI layer: the seam this library replaces is the C++ class
 * OverlapBuilder (src/overlap_builder.h:19-45) as used by Overlapping::run (src/overlap.cpp:41-47) and the
 * per-read call OverlapBuilder::overlap() that parallel::foreach drives (src/overlap_builder.cpp:269-280,
 * 453-457; src/parallel_framework.h:16-59).  The batch entry points below are "overlap() for a batch of
 * reads"; index lifetime replaces FMIndex::load (src/fmindex.cpp:353-366) + SuffixArray::load
 * (src/suffix_array.cpp:104-118).  Plain pointers and sizes only; nothing throws across this boundary;
 * every function returns SIGAX_OK (0) or a negative error code and sigax_last_error() gives the text.
 *
 * All compute runs in hand-written HIP kernels for gfx950.  There is no CPU fallback: if no GPU is
 * visible the calls fail with SIGAX_E_DEVICE.
 */
#ifndef SIGAX_H_
#define SIGAX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIGAX_OK           0
#define SIGAX_E_ARG       -1  /* bad argument */
#define SIGAX_E_IO        -2  /* file missing / malformed (.bwt magic, .sai header) */
#define SIGAX_E_DEVICE    -3  /* HIP error (no device, out of memory, launch failure) */
#define SIGAX_E_CAPACITY  -4  /* a device arena overflowed and could not be grown */
#define SIGAX_E_SUBSTRING -5  /* reserved */
#define SIGAX_E_STATE     -6  /* call order (e.g. edges requested before read metadata was set) */

/* flags of the overlap calls: OverlapBuilder ctor arguments (src/overlap_builder.h:21-24) */
#define SIGAX_IRREDUCIBLE  1u  /* irreducible=true  (CLI: absence of -x/--exhaustive, src/overlap.cpp:43) */
#define SIGAX_RC           2u  /* rc=true           (CLI: absence of --no-opposite-strand) */
#define SIGAX_EDGES        4u  /* also materialise edge records (Hit2OverlapConverter, src/overlap_builder.cpp:345-375) */
#define SIGAX_DUPLICATE    8u  /* OverlapBuilder::duplicate instead of overlap (src/overlap_builder.cpp:1184-1195): only the
                                  seq/fmi and complement(seq)/rfmi finds, minOverlap = read length, blocks = the containment
                                  blocks; min_overlap and the other mode flags are ignored.  Used by `siga rmdup`. */

typedef struct sigax_index sigax_index; /* both FM-indexes + both .sai tables, resident on one GPU */
typedef struct sigax_batch sigax_batch; /* device workspace for batches of reads */

/* One OverlapBlock in the order it is serialised to the hits text (src/overlap_builder.cpp:138-141,198-201):
 * capped pair, raw pair, length, AlignFlags (bit0 QUERYREV, bit1 TARGETREV, bit2 QUERYCOMP; :42-44). */
typedef struct sigax_block {
  uint64_t capped0_lo, capped0_hi, capped1_lo, capped1_hi;
  uint64_t raw0_lo, raw0_hi, raw1_lo, raw1_hi;
  uint32_t length;
  uint32_t af;
  uint64_t reserved;  /* pads the record to 80 bytes = 5 x 16 so device vector accesses stay aligned */
} sigax_block;

/* One kept overlap of Hit2OverlapConverter::convert (src/overlap_builder.cpp:345-375): read indices are
 * positions in the indexed read set; coordinates follow from (length, af, read lengths) exactly as
 * OverlapBlock::overlap computes them (src/overlap_builder.cpp:158-175). */
typedef struct sigax_edge {
  uint32_t query, target;
  uint32_t length;
  uint32_t af;
} sigax_edge;

typedef struct sigax_stats {
  uint64_t n_reads;
  uint64_t n_candidate_blocks; /* blocks pushed by OverlapBlockFinder::find, containments included */
  uint64_t n_blocks;           /* blocks returned */
  uint64_t n_edges;
  uint64_t n_occ_find;         /* Occ(i) evaluations made by the block finder (N_occ_min share) */
  uint64_t n_occ_extract;      /* ... by sub-maximal filter + irreducible extraction */
  uint64_t n_substring;        /* reads flagged substring (SS:i:1) */
  uint64_t n_slow_reads;       /* reads that needed the general (serial) filter/extract kernel */
  uint64_t n_extract_errors;   /* reads where extract() hit "substring read found" (src/overlap_builder.cpp:754-757) */
  uint64_t n_sectors_find;     /* distinct 64-byte sectors of the rank tables the finder's formulation touches: per step one
                                  or two granules (one-step) or both halves of one or two 128-byte lines (two-step table) */
  uint64_t n_sectors_extract;  /* the same for sub-maximal filter + irreducible extraction, per round and block */
} sigax_stats;

typedef struct sigax_result {
  uint32_t     n_reads;
  uint64_t*    block_offs;  /* n_reads+1; blocks of read r are blocks[block_offs[r] .. block_offs[r+1]) in the
                               reference's list order */
  sigax_block* blocks;
  uint8_t*     substring;   /* n_reads; OverlapResult::substring (src/overlap_builder.cpp:211-216) */
  uint64_t     n_edges;
  sigax_edge*  edges;       /* only with SIGAX_EDGES; in hits order = the ED order of the reference at -t 1 */
  sigax_stats  stats;
} sigax_result;

typedef struct sigax_index_info {
  uint64_t n_symbols;   /* BWT length (= sum(len+1)) per strand */
  uint64_t n_strings;   /* reads */
  uint64_t device_bytes;
  uint64_t pred[5];     /* C[] of the forward index: FMIndex::getPC (src/fmindex.h:129-131) */
  int      device;
  int      wide;        /* 1 if positions need 64 bits */
} sigax_index_info;

const char* sigax_last_error(void);           /* thread-local text of the last failure */
int  sigax_device_count(int* n);
/* A HIP stream (non-blocking) for callers that do not link HIP themselves, e.g. to keep two batch objects in flight
 * (sigax_batch_run).  The handle is a hipStream_t and may be passed wherever this header takes a `stream`. */
int  sigax_stream_create(int device, void** stream);
void sigax_stream_destroy(int device, void* stream);

/* FMIndex::load x2 + SuffixArray::load x2 (src/overlap.cpp:41-42, src/overlap_builder.cpp:466).  The .sai
 * paths may be NULL when SIGAX_EDGES is never requested.  rbwt_path NULL or "": the forward strand alone -- what
 * `siga index --no-reverse` writes and `siga correct` loads (src/correct.cpp:41-47) -- serving sigax_occ_batch (which = 0),
 * sigax_kmer_count_batch and sigax_correct_*; overlap runs on such an index fail with SIGAX_E_STATE. */
int  sigax_index_open(const char* bwt_path, const char* rbwt_path, const char* sai_path, const char* rsai_path,
                      int device, sigax_index** out);
/* Same from memory: RL units exactly as in the .bwt payload (src/rlstring.h:10-63), read ids of the .sai lines
 * (rruns NULL with n_rruns 0: forward strand only). */
int  sigax_index_open_mem(const uint8_t* runs, uint64_t n_runs, const uint8_t* rruns, uint64_t n_rruns,
                          uint64_t n_symbols, uint64_t n_strings, const uint32_t* sai, const uint32_t* rsai,
                          int device, sigax_index** out);
/* A replica of an open index (tables, .sai rows, read metadata set so far) on another GPU of the node, copied device to
 * device -- over xGMI between peers -- instead of being read, decoded and uploaded again. */
int  sigax_index_clone(const sigax_index* src, int device, sigax_index** out);
void sigax_index_close(sigax_index*);
int  sigax_index_info_get(const sigax_index*, sigax_index_info* out);
/* Per-read metadata Hit2OverlapConverter keeps (ReadInfo{name,length}, src/overlap_builder.cpp:333-343).
 * name_rank[i] = rank of read i's name among all names under std::string operator< (equal names, equal rank):
 * the dedup rule of src/overlap_builder.cpp:358,365 needs only equality and order of names. */
int  sigax_index_set_reads(sigax_index*, const uint32_t* lengths, const uint32_t* name_rank, uint64_t n);

/* `siga index` for one strand on the GPU: SuffixArrayBuilder "sais2" + BWT(sa, reads) + the .sai rows
 * (src/indexer.cpp:80-104, src/suffix_array_builder.cpp:472-674, src/bwt.cpp:7-32, src/suffix_array.cpp:17-44).
 * seqs/offs as in sigax_overlap_batch; reverse != 0 indexes the reversed reads (src/indexer.cpp:60-64: the .rbwt/.rsai
 * pair).  On success *runs holds *n_runs RL units exactly as the .bwt payload stores them (31-cap of src/bwt.cpp:17),
 * *sai the n_reads read ids of the full-read suffixes in suffix order, *n_symbols = sum(len + 1).  Both arrays are
 * malloc'd; release with sigax_free.  SIGAX_E_CAPACITY = input too repetitive for the device sort (the host falls
 * back to its own SA-IS). */
int  sigax_build_strand(const char* seqs, const uint64_t* offs, uint64_t n_reads, int reverse, int device,
                        uint8_t** runs, uint64_t* n_runs, uint32_t** sai, uint64_t* n_symbols);
void sigax_free(void* p);
/* Brackets several builds (the two strands of one `siga index`): between open (1) and close (0) the builders' device
 * workspace is kept and handed from one call to the next instead of going back to the driver -- hipMalloc hands out
 * cleared memory, and clearing the tens of GB a strand's sort works in again for the next strand cost more than its
 * kernels.  Optional; nests; without it every call returns its workspace when it ends. */
void sigax_build_session(int open);

/* The index is here to stay (a service, a benchmark: many more reads than it holds will be asked of it): builds the
 * extractor's row tables now and returns when they are in place.  Without this call they are built in the background once
 * the index has been asked for as many reads as it holds -- one pass of `siga overlap` over the indexed reads
 * (src/overlap.cpp:41-47) never pays their build.  No-op when they are there already or do not fit. */
int  sigax_index_prepare(sigax_index*);
/* The same for an index that will serve overlap runs with this minimum overlap (the CLI's -m, src/overlap.cpp:44): besides
 * the row tables, the block finder's DEEP START TABLE for K = min(min_overlap, 56) -- every distinct K-mer of the indexed
 * reads with the state OverlapBlockFinder::find (src/overlap_builder.cpp:846-871) holds after consuming it, so that a
 * chain starts where its output starts (nothing is pushed below minOverlap, :861) instead of walking there.  Serves every
 * later run whose min_overlap is at least that K; runs with a smaller one, and chains whose K-mer is not in the reads,
 * walk as before -- same bytes out.  Without this call the table is built in the background, for the min_overlap of the
 * run at hand, once the index has been asked for as many reads as it holds.  No-op when it does not fit the free memory
 * (32 bytes x 2 per distinct K-mer and strand) or min_overlap < 16. */
int  sigax_index_prepare_overlap(sigax_index*, uint32_t min_overlap);

/* Self-check of an open index: are the BWT rows of strand `which` (0 forward, 1 reverse) in the suffix order `siga index`
 * produces (SuffixArrayBuilder "sais2": src/suffix_array_builder.cpp:472-674; SURVEY.md App. C: one '$' smaller than
 * ACGT, comparisons continuing past it, end of text smallest)?  Every pair of adjacent rows is compared on the device.
 * *n_bad = pairs out of order (0 = the index is in order), *first_bad = the first such row, *n_undecided = pairs still
 * tied after 4096 reads' worth of symbols.  Needs the .sai tables and sigax_index_set_reads(); SIGAX_E_STATE when the index
 * holds non-ACGT bases or has no row tables. */
int  sigax_index_check_order(sigax_index*, int which, uint64_t* n_bad, uint64_t* first_bad, uint64_t* n_undecided);

/* FMIndex::getOcc(i) for many positions (src/fmindex.cpp:320-323): which = 0 forward, 1 reverse index;
 * counts5[5*k..] = Occ($,A,C,G,T) inclusive of positions[k]; position 2^64-1 gives zeros. Host buffers. */
int  sigax_occ_batch(sigax_index*, int which, const uint64_t* positions, uint64_t n, uint64_t* counts5);
/* FMIndex::Interval::occurrences (src/fmindex.h:80-86) for n k-mers of length k on the forward index. */
int  sigax_kmer_count_batch(sigax_index*, const char* kmers, uint32_t k, uint64_t n, uint64_t* counts);

/* KmerCorrector::process for a batch of reads (src/correct_processor.cpp:72-229, `siga correct -a kmer`).  quals may be
 * NULL (FASTA input: every base scores 15).  out_seqs has the layout of seqs and receives the corrected sequence of
 * reads that became all-solid and the unchanged sequence otherwise; valid[r] = CorrectResult::validQC (only those
 * reads are written by PostCorrector, :242-265).  Parameters as the CLI's -k/-x/-i/-O (defaults 31/3/10/1,
 * src/correct_processor.h:15-20).  Forward index only, as the reference. */
int  sigax_correct_batch(sigax_index*, const char* seqs, const char* quals, const uint64_t* offs, uint32_t n_reads,
                         uint32_t kmer_size, int32_t kmer_threshold, uint32_t kmer_rounds, uint32_t count_offset,
                         char* out_seqs, uint8_t* valid);

/* The same with every buffer in device memory, asynchronous on `stream` (a hipStream_t or NULL): d_offs u64[n_reads+1],
 * d_quals may be NULL, d_stat4 = 4 u64 of device scratch that receive {reads longer than the kernel supports (their
 * valid[] is 2), rank-table sectors asked for, k-mer lookups made, reserved}.  What a batching runtime and bench.py use.
 * The first correction call on an index allocates the table of all 13-mer intervals (537 MB, 1 GB with 64-bit positions:
 * the one synchronous step; sigax_batch_size_hint leaves room for it) and builds it on `stream`, ahead of the call's own
 * kernel; calls on other streams wait for that build through an event. */
int  sigax_correct_device(sigax_index*, const void* d_seqs, const void* d_quals, const void* d_offs, uint64_t n_reads,
                          uint32_t kmer_size, int32_t kmer_threshold, uint32_t kmer_rounds, uint32_t count_offset,
                          void* d_out_seqs, void* d_valid, void* d_stat4, void* stream);

/* OverlapBuilder::overlap for a batch (host buffers in, host buffers out).  seqs = concatenated read bytes,
 * offs[n_reads+1]; read r of the batch is read `read_base + r` of the indexed set (only used for edges).
 * The result is filled with malloc'd arrays; release with sigax_result_free. */
int  sigax_overlap_batch(sigax_index*, const char* seqs, const uint64_t* offs, uint32_t n_reads, uint32_t read_base,
                         uint32_t min_overlap, uint32_t flags, sigax_result* out);
void sigax_result_free(sigax_result*);

/* Device-resident pipeline (what a batching runtime and bench.py use). */
int  sigax_batch_create(sigax_index*, uint32_t max_reads, uint64_t max_bases, uint32_t max_read_len, sigax_batch** out);
void sigax_batch_destroy(sigax_batch*);
/* Copy reads to the device (async on `stream`, a hipStream_t or NULL). */
int  sigax_batch_upload(sigax_batch*, const char* seqs, const uint64_t* offs, uint32_t n_reads, void* stream);
/* Use reads that already sit in device memory (d_seqs bytes, d_offs u64[n_reads+1]); max_len = longest read. */
int  sigax_batch_set_device_reads(sigax_batch*, const void* d_seqs, const void* d_offs, uint32_t n_reads,
                                  uint64_t n_bases, uint32_t max_len);
/* The batch's reads need not be consecutive reads of the indexed set: ids[r] = read r's place in the index's read table
 * (what an ED record's query field then says; blocks and substring flags do not depend on it).  This is what lets a caller
 * shard its reads by anything but position in the file -- e.g. by a locality key, so that one GPU's reads cover one part
 * of the genome deeply instead of all of it thinly (bench.py --gpus N, siga_amd/sharding.py).  The ids stay with the batch
 * object for every later run until set again; NULL = read_base + r as before.  A run whose reads are not as many as the ids
 * fails with SIGAX_E_STATE; an id beyond the indexed reads fails the upload (host form) or the run's
 * sigax_batch_finish (device form) with SIGAX_E_ARG, without touching memory outside the tables.
 * No counterpart in the reference (one process, reads in file order: src/overlap_builder.cpp:1113-1182 takes the id from
 * the read's position in the file). */
int  sigax_batch_upload_read_ids(sigax_batch*, const uint32_t* ids, uint32_t n_reads, void* stream);
int  sigax_batch_set_device_read_ids(sigax_batch*, const void* d_ids, uint32_t n_reads);
/* Locality keys of reads in device memory (d_seqs bytes, d_offs u64[n_reads+1]) into d_keys u64[n_reads], queued on `stream`
 * (a hipStream_t or NULL): key = min over the read's 16-mers of hash(canonical 16-mer) << 16 | start offset (csrc/sigax_keys.hip).
 * Reads of one stretch of the genome, either strand, get neighbouring keys: a caller that sorts its reads by key and gives each
 * GPU a contiguous range of that order (with sigax_batch_*_read_ids above) has every GPU walk a part of the index instead of
 * all of it.  Needs no index.  No counterpart in the reference. */
int  sigax_locality_keys(int device, const void* d_seqs, const void* d_offs, uint32_t n_reads, void* d_keys, void* stream);
/* How many sub-batches a run is cut into (0 = automatic).  With more than one, sub-batch i's filter/extract kernels run on
 * an internal stream beside sub-batch i+1's block finder (the first is VALU-bound, the second memory-request-bound). */
int  sigax_batch_set_subbatches(sigax_batch*, uint32_t n);
/* Enqueue the whole path: find -> filter/extract -> order -> edges.  Asynchronous.  The kernels run on three internal
 * streams owned by the index (finder, filter/extract, tail) and ordered against `stream`: work already on `stream` is
 * waited for, and `stream` waits for the run's last kernel.  Several batch objects of one index may be in flight at
 * once (each with its own `stream`): their finder launches queue behind each other, so batch k+1's finder starts beside
 * batch k's filter/extract tail.  A batch object must be finished before it is run again. */
int  sigax_batch_run(sigax_batch*, uint32_t read_base, uint32_t min_overlap, uint32_t flags, void* stream);
/* Wait for the stream, check arena overflow flags (growing arenas and re-running if needed), fill stats. */
int  sigax_batch_finish(sigax_batch*, void* stream, sigax_stats* stats);
/* Device pointers of the finished batch's outputs (valid until the next run on this batch). */
int  sigax_batch_device_outputs(sigax_batch*, const sigax_block** d_blocks, const uint64_t** d_block_offs,
                                const uint8_t** d_substring, const sigax_edge** d_edges);
int  sigax_batch_download(sigax_batch*, sigax_result* out);
/* Only what the ASQG writer needs (OverlapPostProcess + Hit2OverlapConverter, src/overlap_builder.cpp:291-375): the
 * substring flags (n_reads bytes, caller's buffer, may be NULL) and the edge records (malloc'd; release with sigax_free). */
int  sigax_batch_download_edges(sigax_batch*, uint8_t* substring, sigax_edge** edges, uint64_t* n_edges);
/* Reads of up to max_read_len bases one batch object can take when `in_flight` batch objects share the device's free
 * memory now (the reference's threads x batch-size is a host notion; the device batch is sized from HBM). */
int  sigax_batch_size_hint(sigax_index*, uint32_t max_read_len, uint32_t min_overlap, uint32_t flags, uint32_t in_flight,
                           uint32_t* max_reads);
/* Device time of the kernels of the last finished run, measured with HIP events on the streams they were launched on,
 * summed over the run's sub-batch launches: ms[0] find, ms[1] filter/extract (32- and 64-lane launches),
 * ms[2] filter/extract (general), ms[3] order, ms[4] edges.  *n_sub = finder launches of the run (sub-batches, times
 * two when the finder runs once per strand's two-step table); the other kernels run once per sub-batch. */
int  sigax_batch_kernel_ms(sigax_batch*, float ms[5], uint32_t* n_sub);

/* How the last finished run of a batch object was carried out: what a caller needs to turn sigax_stats and
 * sigax_batch_kernel_ms into requests and bytes per launch without re-deriving the library's choices. */
typedef struct sigax_run_info {
  uint32_t n_sub;         /* sub-batches of the run */
  uint32_t find_per_sub;  /* finder launches per sub-batch (2: one per strand's table) */
  uint32_t two_step;      /* the finder gathered 128-byte two-step lines (two backward steps each); 0: 64-byte granules */
  uint32_t coop;          /* ... cooperatively through LDS (indexes of 2^31 symbols and more) */
  uint32_t read_order;    /* the finder walked the batch in its locality order */
  uint32_t cap;           /* candidate slots per chain of the run */
  uint32_t worst_cap;     /* ... had they been sized for the worst case (one per overlap length + 1) */
  uint32_t row_bits;      /* bits per entry of the index's row tables, 0 = none */
  uint32_t row_syms;      /* symbols an entry carries */
  uint32_t row_text;      /* the stretch text exists (extension rounds are read off it) */
  uint32_t row_direct;    /* ... reached through the direct maps (8 bytes per read) instead of a row table */
  uint32_t deep_k;        /* chains started from the deep start table with this K (0: from the 12-mer table or the first symbol) */
  uint64_t arena_bytes;   /* candidate arena of this batch object */
  uint64_t workspace_bytes; /* all device buffers of this batch object */
  uint64_t reruns;        /* runs of this batch object repeated so far because an arena was too small */
  float    order_ms;      /* device time of the locality ordering inside this run (0: the order of an earlier run was reused) */
} sigax_run_info;
int  sigax_batch_run_info(sigax_batch*, sigax_run_info* out);

/* ---- Multi-GPU: the one exchange step -----------------------------------------------------------------
 * Reads shard across the GPUs of a node and the index is replicated (sigax_index_clone), so the path has exactly one
 * exchange: the variable-length gather of a step's fixed-size edge records to one rank -- what replaces the serial
 * hits -> ASQG pass of src/overlap_builder.cpp:466-483 when one process drives each GPU.  Over RCCL (xGMI between the
 * MI355X of a node): ncclAllGather of the per-rank counts, then grouped ncclSend / ncclRecv of the 16-byte records.
 * Gathered in rank order = read order: the ED order of a one-GPU run.  RCCL is bound at run time; without it these calls
 * fail with SIGAX_E_DEVICE and nothing else in the library is affected.  (The C++ host's `siga overlap --gpus N` drives
 * all GPUs from one process and copies each GPU's records to the host over that GPU's own PCIe link instead.) */
typedef struct sigax_comm sigax_comm;
#define SIGAX_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and ships it to the other ranks by whatever means the launcher has */
int  sigax_comm_unique_id(uint8_t id[SIGAX_COMM_ID_BYTES]);
/* collective over all `world` ranks (ncclCommInitRank), each on its own `device` */
int  sigax_comm_create(int device, int rank, int world, const uint8_t id[SIGAX_COMM_ID_BYTES], sigax_comm** out);
void sigax_comm_destroy(sigax_comm*);
/* The gather, two collectives.  sigax_gather_counts: every rank says how many records it holds; counts[world] (host)
 * receives all ranks' counts before the call returns (ncclAllGather + one host wait on `stream`), so that the root can size
 * its buffer.  sigax_gather_edges: with those counts, the transfer of the records is enqueued on `stream` (grouped ncclSend /
 * ncclRecv); d_local = this rank's counts[rank] records in device memory (e.g. sigax_batch_device_outputs' d_edges), on
 * `root` d_out (device, room for the sum of the counts) receives rank 0's records first; other ranks pass d_out = NULL. */
int  sigax_gather_counts(sigax_comm*, uint64_t n_local, uint64_t* counts, void* stream);
int  sigax_gather_edges(sigax_comm*, const sigax_edge* d_local, const uint64_t* counts, int root, sigax_edge* d_out,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGAX_H_ */
