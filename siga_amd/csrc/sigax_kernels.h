// siga_amd/csrc/sigax_kernels.h -- argument blocks and launch wrappers shared by sigax_kernels.hip and sigax_api.cpp
#ifndef SIGA_AMD_SIGAX_KERNELS_H_
#define SIGA_AMD_SIGAX_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigax.h"
#include "fm_layout.h"

static_assert(sizeof(sigax_block) == 80, "sigax_block must be 5 x 16 bytes");
static_assert(sizeof(sigax_edge) == 16, "sigax_edge must be 16 bytes");

// slots of the per-batch device statistics / allocator block (u64 each, zeroed before every run)
enum {
  DS_OCC_FIND = 0,
  DS_CAND_BLOCKS,
  DS_FIND_OVERFLOW,
  DS_OCC_EXTRACT,
  DS_EXTRACT_ERRORS,
  DS_POOL_OVERFLOW,
  DS_SUBSTRING,
  DS_FIN_TOP,      // bump allocator of the unordered final-block arena (may run past fin_cap: needed size)
  DS_TOTAL_BLOCKS, // written by the scan: sum of per-read block counts
  DS_TOTAL_EDGES,  // written by the edge scan
  DS_SLOW_BASE,    // [DS_SLOW_BASE + i]: reads sub-batch i queued for the general kernel (i < SIGAX_MAX_SUB)
  DS_W64_BASE = DS_SLOW_BASE + 8,  // [DS_W64_BASE + i]: (read, side) items sub-batch i queued for the 64-lane launch
  DS_SEC_FIND = DS_W64_BASE + 8,   // distinct 64-byte sectors of the rank tables the finder asked for, step by step
  DS_SEC_EXTRACT,                  // ... filter/extract, round by round
  DS_MAX_CHAIN,                    // most candidate blocks any chain of the run pushed (also when its slots ran out)
  DS_BAD_IDS,                      // (read, side) items whose read id (sigax_batch_set_read_ids) is beyond the indexed read set
  DS_PROF_BASE = 32,               // 32 diagnostic counters (builds with -DSIGAX_FX_PROFILE only)
  DS_W64B_BASE = 64,               // [DS_W64B_BASE + i], [DS_W64C_BASE + i]: items in the second and third queue between
  DS_W64C_BASE = 72,               // sub-batch i's filter/extract launches
  DS_W64D_BASE = 80,               // ... in the fourth (what the 16-lane launch hands to the 32-lane one)
  DS_COUNT = 88
};
#define SIGAX_MAX_SUB 8

struct Ent;  // defined in sigax_kernels.hip (48 bytes)
#define SIGAX_ENT_BYTES 48

struct FindArgs {
  FmStrand fwd, rev;
  const unsigned char* seqs;
  const unsigned long long* offs;
  uint32_t n_reads, minov, chain_mask, cap;  // chain_mask bit o = find o runs; cap = slots per chain, last = containment
  uint32_t max_seen;                 // chains no longer than this need not report their length (DS_MAX_CHAIN)
  uint32_t start_ok;                 // both strands carry the start table and min-overlap >= 12: chains may start twelve symbols in
  uint32_t deep_k;                   // both strands carry the deep start table with this K <= min-overlap (fm_layout.h), or 0
  uint32_t max_len;                  // the batch's longest read (sizes the per-lane finder's LDS budget)
  uint32_t read_begin, read_end;     // this launch's sub-batch
  uint32_t stage_bytes;              // dynamic LDS per workgroup that may hold the workgroup's reads (set by launch_find)
  uint32_t two_step;                 // both strands carry the two-step table (u32 positions only)
  uint32_t mask_upper;               // per-lane two-step finder: lanes whose upper position lies in the lower one's line do not load it again
  uint32_t coop, coop_stage_bytes;   // cooperative two-step finder (k_find_c2): lines staged through LDS; bytes for 64 reads
  uint32_t coop_grid;                // ... its grid cap (persistent workgroups walk the tiles), 0 = one workgroup per tile
  const uint32_t* perm;              // locality order of the batch: slot -> read (a permutation inside every sub-batch), or NULL
  uint32_t stage_stride;             // with perm: bytes per read in the LDS copy (>= the longest read, a multiple of 4)
  uint32_t chain_base, chains_per_wg; // a workgroup walks chains [chain_base, chain_base + chains_per_wg) (4, or 2: one strand's
                                     // table per launch) of 256 / chains_per_wg reads; chains outside are another launch's
  void* arena;                       // [n_reads][4][cap] candidate records of cand_bytes(wide) each
  uint32_t* chain_cnt;               // [n_reads][4]
  unsigned long long* dstat;
};

struct FxArgs {
  FmStrand fwd, rev;
  const unsigned long long* offs;
  uint32_t n_reads, cap, irreducible;
  uint32_t no_lean;   // skip the lean (single-group only) launches: the previous run of this batch object sent most items on
  const void* arena;
  const uint32_t* chain_cnt;
  unsigned long long n_map;  // entries of the strands' direct maps (= reads of the index)
  Ent* pool;          // [lanes][pool_cap]
  uint32_t pool_cap;
  const uint32_t* work;  // optional list of read ids; NULL = all reads
  unsigned long long n_work;
  const unsigned long long* n_work_ptr;  // if set, the list length is read from device memory
  Ent* wpool;         // fast kernel: [waves][fast_pool_entries_per_wave()]
  uint32_t* work_out; // fast kernel: reads queued for the general kernel, counted in *slow_counter
  unsigned long long* slow_counter;
  // fast kernels: three queues of (read, side) items between the launches of launch_filter_extract_fast, each counted
  // in its *counter; the first launch's count also tells the caller how many items needed more than the strict lean form
  uint32_t* work64;
  unsigned long long* w64_counter;
  uint32_t* work64b;
  unsigned long long* w64b_counter;
  uint32_t* work64c;
  unsigned long long* w64c_counter;
  uint32_t* work64d;
  unsigned long long* w64d_counter;
  // set per launch by launch_filter_extract_fast: input queue (NULL: the sub-batch's read range) and output queue (NULL:
  // the last launch, which queues reads for the general kernel)
  const uint32_t* q_in;
  const unsigned long long* q_in_n;
  uint32_t* q_out;
  unsigned long long* q_out_n;
  uint32_t* q_wide;  // optional: items with more blocks than the launch's lane group has lanes skip the next launch
  unsigned long long* q_wide_n;
  uint32_t read_begin, read_end;  // fast kernel: this launch's sub-batch
  sigax_block* fin;   // unordered final blocks, allocated in per-wave / per-lane chunks
  unsigned long long* item_base;  // [n_reads][2]: where a (read, side) item's blocks start in `fin`
  unsigned long long fin_cap;
  uint32_t* fin_cnt;  // [n_reads][2]: blocks of the suffix side (+ containments) and of the prefix side
  uint32_t* occ_side; // [n_reads][2]: what a completed fast side added to the statistics (taken back on redo)
  uint32_t* slow_flag;// [n_reads]: read queued for the general kernel
  uint8_t* substring; // [n_reads]
  unsigned long long* dstat;
};

struct OrderArgs {
  const sigax_block* fin;
  const unsigned long long* item_base;  // [n_items]
  const uint32_t* fin_cnt;              // [n_items]
  unsigned long long n_items;           // 2 * n_reads
  unsigned long long fin_cap;
  const unsigned long long* offs2;      // [n_items+1] per-(read, side) offsets in the ordered output
  sigax_block* out;
  unsigned long long out_cap;
  const void* arena;   // candidate records: a block whose `reserved` has bit 63 set carries (chain, slot) there instead
  uint32_t cap, wide;  // of its raw intervals, length and flags, which the scatter then takes from the candidate record
  // edge records are counted on the way (per item), unless item_edges is NULL
  uint32_t* item_edges;  // [n_items]
  uint32_t read_base;
  // the batch's reads in the index's read table: read_base + i, or read_ids[i] (a rank's share of a key-range sharding:
  // any subset in any order); an id >= n_index_reads makes no edge and is counted in *bad_ids
  const uint32_t* read_ids;
  uint32_t n_index_reads;
  unsigned long long* bad_ids;
  const uint32_t* sai;
  const uint32_t* rsai;
  unsigned long long n_sai;
  const uint32_t* read_len;
  const uint32_t* name_rank;
};

struct EdgeArgs {
  const sigax_block* blocks;             // ordered output
  unsigned long long blocks_cap;
  const unsigned long long* offs2;       // [n_items+1] first block of each (read, side) item
  const uint32_t* fin_cnt;               // [n_items] its blocks
  unsigned long long n_items;
  uint32_t read_base;
  const uint32_t* read_ids;  // as in OrderArgs (items of out-of-range ids have no records: k_order_scatter counted none)
  const uint32_t* sai;
  const uint32_t* rsai;
  unsigned long long n_sai;
  const uint32_t* read_len;
  const uint32_t* name_rank;
  const unsigned long long* edge_offs;   // [n_items+1] scan of the per-item counts of k_order_scatter
  sigax_edge* edges;
  unsigned long long edge_cap;
};

struct CorrectArgs {
  FmStrand fwd;
  const unsigned char* seqs;
  const unsigned char* quals;  // NULL: every base scores Quality::Phred::DEFAULT_SCORE (15)
  const unsigned long long* offs;
  unsigned long long n_reads;
  uint32_t k, low, high, cutoff, rounds, offset;  // CorrectThreshold: required support low / high (phred >= cutoff)
  unsigned char* out;    // corrected sequences, same layout as seqs
  unsigned char* valid;  // CorrectResult::validQC; 2 = read longer than the kernel supports
  unsigned long long* dstat;  // [0] reads too long, [1] rank-table sectors asked for, [2] k-mer lookups (4 x u64)
  const void* ktab;           // hash of the reads' distinct k-mers with their counts (fm_layout.h: deep start table, K = k), or NULL
  unsigned long long ktab_slots;
  const void* ptab;           // intervals of all pk-mers (launch_prefix_build), or NULL
  uint32_t pk;
  uint32_t max_len;           // upper bound of the batch's read lengths, 0 = unknown (launch_correct picks the kernel form)
  uint32_t only_deferred;     // set by launch_correct: this launch takes the reads the small form marked (valid = 3)
};
void launch_correct(const CorrectArgs& a, bool wide, hipStream_t st);
// table of the intervals of all pk-mers on strand s: prefix_table_bytes(wide, pk) bytes
unsigned long long prefix_table_bytes(bool wide, uint32_t pk);
void launch_prefix_build(const FmStrand& s, bool wide, void* tab, uint32_t pk, hipStream_t st);

void launch_occ_batch(const FmStrand& s, bool wide, const unsigned long long* pos, unsigned long long n,
                      unsigned long long* out, hipStream_t st);
void launch_kmer_count(const FmStrand& s, bool wide, const unsigned char* kmers, uint32_t k, unsigned long long n,
                       unsigned long long* out, hipStream_t st);
void launch_find(const FindArgs& a, bool wide, hipStream_t st);
// rows of strand s in suffix order?  bad3 (device, zeroed except [1] = ~0): pairs out of order, first such row, undecided pairs
void launch_suffix_order_check(const FmStrand& s, const uint32_t* sai, uint32_t* isai_tmp, const uint32_t* read_len,
                               unsigned long long n_strings, unsigned long long* bad3, hipStream_t st);
// start table of the finder for chains whose primary index is `prim` (fm_layout.h): start_table_bytes(wide) bytes
unsigned long long start_table_bytes(bool wide);
void launch_start_build(const FmStrand& prim, const FmStrand& other, bool wide, void* tab, hipStream_t st);
// deep start table of strand `prim` as primary index (fm_layout.h).  launch_deep_scan: *count (zeroed by the caller) += the
// distinct K-mers of s's text = the runs of rows with equal K-symbol prefixes; with list != NULL each run's first row goes
// to list[] (any order, at most list_cap).  launch_deep_fill: their states into tab (nslots entries of deep_entry_bytes(),
// zeroed by the caller); *err (zeroed) counts K-mers whose walk did not come out at their own rows.  slen = the stretches'
// lengths by '$' rank (launch_rows_fill).
unsigned long long deep_entry_bytes();
void launch_deep_scan(const FmStrand& s, const uint32_t* slen, unsigned long long n_stretch, uint32_t K, unsigned long long* count,
                      unsigned long long* list, unsigned long long list_cap, hipStream_t st);
void launch_deep_fill(const FmStrand& prim, const FmStrand& other, bool wide, const uint32_t* slen, unsigned long long n_stretch, uint32_t K,
                      const unsigned long long* list, unsigned long long n_list, void* tab, unsigned long long nslots, unsigned long long* err,
                      hipStream_t st);
unsigned long long find_stage_capacity(unsigned max_len);  // bytes of reads a finder workgroup can stage in LDS (batch's longest read given)
void launch_filter_extract(const FxArgs& a, bool wide, unsigned grid, hipStream_t st);
// qhint: items the four queues held last time (per sub-batch; ~0 = unknown), or NULL
void launch_filter_extract_fast(const FxArgs& a, bool wide, unsigned grid32, unsigned grid64, const unsigned long long* qhint, hipStream_t st);
unsigned long long fast_pool_entries_per_wave();
void launch_scan(const uint32_t* cnt, unsigned long long n, unsigned long long* partial, unsigned long long* offs,
                 unsigned long long* total_out, hipStream_t st);
unsigned long long scan_partials_needed(unsigned long long n);
// Two-step table of one strand (fm_layout.h): gran2 = (n/64+1) x 32 u32; cnt = 20 x (n/64+1) u32, offs = (n/64+2) u64,
// partial = scan_partials_needed(n/64+1) u64, total = 1 u64 of scratch.
void launch_build2(const FmStrand& s, bool wide, uint32_t* gran2, unsigned long long* super2, uint32_t* cnt, unsigned long long* offs, unsigned long long* partial,
                   unsigned long long* total, hipStream_t st);
// Row table + stretch text of one strand (fm_layout.h), two walks over every stretch (n_stretch = C['A'] of them):
// launch_stretch_scan fills info[n_stretch] and *maxlen (zeroed by the caller) = the longest stretch, from which the caller
// sizes the entries; launch_rows_fill writes the table (zeroed by the caller, n * sa_bits bits + 16 bytes; or sa = NULL), unless
// NULL the text rows, and unless NULL the stretches' lengths by their rank among the '$' rows.
void launch_stretch_scan(const FmStrand& s, bool wide, unsigned long long n_stretch, unsigned long long* info, uint32_t* maxlen, hipStream_t st);
void launch_rows_fill(const FmStrand& s, bool wide, unsigned long long n_stretch, const unsigned long long* info, unsigned char* sa,
                      uint32_t sa_bits, uint32_t ld_bits, uint32_t t_bits, unsigned char* text, uint32_t text_stride, uint32_t* slen,
                      hipStream_t st);
// direct map of strand X as extension index (fm_layout.h) from the other strand's .sai ids (sai_y), X's own (sai_x; isai_tmp =
// n u32 of scratch for its inverse) and X's stretch lengths by '$' rank (slen_x, written by launch_rows_fill)
void launch_xmap(const uint32_t* sai_y, const uint32_t* sai_x, uint32_t* isai_tmp, const uint32_t* slen_x, unsigned long long n,
                 unsigned long long* xmap, hipStream_t st);
void launch_order_scatter(const OrderArgs& a, hipStream_t st);
unsigned long long fast_fin_chunk();
unsigned long long cand_bytes(bool wide);
void launch_pick_read_offsets(const unsigned long long* offs2, unsigned long long n_reads, unsigned long long* block_offs,
                              hipStream_t st);
void launch_edges_fill(const EdgeArgs& a, hipStream_t st);
// sigax_index_build.hip: locality order of a batch's reads (a permutation inside each of the nsub slot ranges bounds[0..nsub]),
// queued on `st` without a host wait.  keys = n u32 of scratch, vals = n u32 = the order once the stream gets there
// (*result = vals), tmp = sigax_order_reads_tmp_bytes(nsub) bytes for the class counters.
size_t sigax_order_reads_tmp_bytes(uint32_t nsub);
int sigax_order_reads(const unsigned char* d_seqs, const unsigned long long* d_offs, uint32_t n, uint32_t max_len, const uint32_t* bounds, uint32_t nsub,
                      uint32_t* keys, uint32_t* vals, void* tmp, size_t tmp_bytes, const uint32_t** result, hipStream_t st);

#endif
