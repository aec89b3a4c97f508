// siga_amd/csrc/fm_layout.h -- HBM layout of one FM-index strand and of the per-batch arenas.
//
// On disk the index stays the reference's 1-byte run-length units (src/rlstring.h:10-63, src/bwt.cpp:34-190).
// In HBM it is decoded once into a fixed-rate, bit-sliced rank structure so that FMIndex::getOcc(i)
// (src/fmindex.cpp:188-231: 48 B large marker + 12 B small marker + a serial walk over <=64 symbols of
// runs) becomes ONE aligned 64-byte granule and a handful of v_bcnt instructions:
//
//   granule g (64 B) covers BWT symbols [128 g, 128 g + 128) as 4 chunks of 16 B;
//   chunk j = { u32 cnt_j, u32 p0, u32 p1, u32 p2 }
//     cnt_0..3 = number of A, C, G, T in BWT[0, 128 g)   (relative to the superblock in wide mode)
//     p0,p1,p2 = bit planes of the 3-bit symbol codes of symbols [128 g + 32 j, +32); bit k = symbol k
//     codes: $=0 A=1 C=2 G=3 T=4 (src/alphabet.h:14) so  A = p0&~p1, C = p1&~p0, G = p0&p1, T = p2.
//   Occ('$') follows from the position: Occ($, p) = p - (A + C + G + T).
//
// Each chunk holds its own 32 symbols + one of the four counters, so a granule can be consumed by one lane
// (4 x global_load_dwordx4, what the kernels do) or by a quad of lanes (one dwordx4 each + a quad reduction).
// Two adjacent granules form one 128-byte line with counters at symbol 0 and symbol 128.
// 4 bits per symbol: C2 (1.51e8 symbols) = 75.5 MB per strand, C5 (1.255e10) = 6.3 GB per strand.
#ifndef SIGA_AMD_FM_LAYOUT_H_
#define SIGA_AMD_FM_LAYOUT_H_

#include <stdint.h>

#define SIGAX_GRANULE_SYMS 128
#define SIGAX_GRANULE_BYTES 64
#ifndef SIGAX_SUPER_SHIFT
#define SIGAX_SUPER_SHIFT 32  /* wide mode: u64 counters every 2^32 symbols (tests build a variant with a small shift) */
#endif

/* AlignFlags of the four finds of OverlapBuilder::overlap (src/overlap_builder.cpp:52-55,1124-1132), by chain:
 * 0 seq on fmi (SuffixPrefix 000), 1 revcomp(seq) on fmi (PrefixPrefix qr,qc = 101b),
 * 2 reverse(seq) on rfmi (PrefixSuffix qr,tr = 011b), 3 complement(seq) on rfmi (SuffixSuffix tr,qc = 110b). */
#define SIGAX_AF_CHAIN0 0u
#define SIGAX_AF_CHAIN1 5u
#define SIGAX_AF_CHAIN2 3u
#define SIGAX_AF_CHAIN3 6u

/* Two-step table (u32-position indexes below 2^31 symbols; built on the device at open time from the granules above):
 * one 128-byte granule per 64 BWT rows j, holding for each row its BWT symbol c1(j) and, for c1 in ACGT, the symbol
 * c2(j) = BWT[LF(j)] that precedes it in the text (the "second" backward step):
 *   words 0..3    number of A, C, G, T among c1 of rows [0, 64 g)
 *   words 4..19   [c = A..T][x = A..T] number of rows in [0, 64 g) with c1 = c and c2 = x
 *   words 20..31  bit planes y1 z1 w1 y2 z2 w2 (64 bits each, low word first) of the codes of c1 and c2
 * With R2(x,c,p) = rows j < p with c1 = c, c2 = x:  Occ(e, C[c] + Occ(c, p)) = Occ(e, C[c]) + R2(e, c, p), because LF
 * keeps the rows with c1 = c in order -- so two consecutive backward steps (c, then e) need only the two positions
 * of the first one.  A 128-byte line costs the memory system what a 64-byte one does (profiles/r01_gather_probe3.txt). */
#define SIGAX_GRAN2_SYMS 64
#define SIGAX_GRAN2_WORDS 32

/* 64-bit-position indexes: the 20 counters of a two-step line are relative to the line's superblock of 2^SIGAX_SUPER_SHIFT
 * rows; super2[sb][20] holds the absolute values at the superblock's first row (same order as the line's words 0..19). */

/* Row table (built on the device, optional): the suffix array of the index in the form the irreducible extractor needs it,
 * bit-packed.  For BWT row p, whose suffix starts at offset t of its stretch (a read, or the piece of a read between two
 * symbols of rank 0: terminators and non-ACGT bases, src/alphabet.h:19-39), the entry is (t << ld_bits) | ld with
 * ld = Occ('$') at the row reached after t backward steps -- the row whose BWT symbol has rank 0, i.e. the index of the
 * stretch among all stretches in suffix order (what the .sai table is indexed by).  ld_bits + t_bits (= bits of the longest
 * stretch) per row: BASELINE configs[1] 28 bits, [2] 33, [4] 34 (the round-2 layout spent 128 bits per row on the same
 * information and could not hold configs[4]).  When memory allows, an entry also carries the first K symbols on the backward
 * path from its row, 2 bits each (rank - 1: inside a stretch only A, C, G, T occur, and t says where it ends), above those
 * bits, first symbol highest, as many as keep the entry within 57 bits (at most 14): sa_bits = ld_bits + t_bits + 2 K --
 * 56 bits with 14 symbols at configs[1], 57 with 12 at configs[2].  Entry p sits at bit p * sa_bits: one unaligned 8-byte
 * load.
 *
 * Stretch text (same walk): for stretch ld its symbols as rank - 1, offset k at bits 2 k, 2 k + 1 of the row
 * text + ld * text_stride.  A single-row block's right extension IS the backward path from its row (IntervalPair::updateR
 * with the one symbol at its row = LF; capped[0] never moves): the symbols at offsets t-1, t-2, ..., 0 and then rank 0, and
 * the block is emitted with Occ('$') = ld when its read ends (src/overlap_builder.cpp:747-766).  So the rounds of the
 * extractor are one row-table lookup, one text load per 28 rounds (whose address depends on nothing but registers) and
 * no rank arithmetic. */
/* Direct map (round 3, what indexes with .sai tables and ACGT-only reads get INSTEAD of the row table): a single-row block
 * names its target read itself -- capped[0] is the read's rank among the '$' rows of the finder's primary index, the
 * index into that strand's .sai -- and the block's length says how far into the target the overlap reaches.  For strand X
 * as the EXTENSION index (OverlapBlock::index, src/overlap_builder.cpp:177-179) xmap[r] = (stretch of that read in X's
 * order) | (its length << 32), r = the rank in the OTHER strand's '$' rows: the block's path starts at offset
 * length(target) - block.length of that stretch's text.  8 bytes per read instead of 34-57 bits per BWT symbol, and the
 * lookup hits a table the caches hold instead of a random row of a table of tens of gigabytes. */
#define SIGAX_TEXT_WINDOW 28  /* symbols per text load that every lane can count on (an unaligned 8-byte load holds 57 to 64 bits of them) */

/* Start table of the block finder (optional, built on the device at open): for every 12-mer, what a chain whose first twelve
 * symbols (in the order the chain consumes them, first one in the highest two bits of the code, rank - 1 each) are that
 * 12-mer holds after consuming them with THIS strand as its primary index: { capped/raw [0].lower, [1].lower, size, steps }
 * -- steps = 12, or where the range emptied (the chain then reports the same number of rank evaluations as the walk).
 * 16 bytes per entry (32 with 64-bit positions), 268 MB per strand: one gather instead of eleven dependent steps. */
#define SIGAX_START_K 12

/* Deep start table of the block finder (round 4; optional, built on the device once the index is known to stay open and the
 * minimum overlap it is asked for is known): the same idea carried to where the finder's OUTPUT starts.  Below min-overlap
 * symbols a chain emits nothing (src/overlap_builder.cpp:861: the '$' probe only counts from minOverlap on), so all a chain
 * needs from its first K = min(min-overlap, 56) symbols is the state after them -- and a K-mer that occurs in the reads at
 * all is one of at most n of them, not one of 4^K.  So: a hash table per strand (as primary index) of every distinct K-mer
 * of that strand's text (found as the runs of equal K-symbol prefixes among adjacent rows of the row table: a K-mer's rows
 * ARE its interval) with the state a chain holds after consuming it, computed by the same IntervalPair::init + updateL walk
 * the finder does (k_deep_fill).  BASELINE configs[1], min-overlap 45: a chain starts 45 symbols in after ONE lookup
 * instead of 12 symbols in and seventeen double steps; chains whose K-mer is not in the table (reads with errors: the
 * reverse-complement chains' K-mers exist only if another read has them) start from the 12-mer table as before.
 * Entry (32 bytes, 4 per 128-byte line, linear probing, load <= 0.5 when memory allows):
 *   u64 k0   symbols 0..31 in the order the chain consumes them, symbol i at bits 2i (rank - 1)
 *   u64 k1   symbols 32..55 likewise in bits 0..47; bits 48..63 = 0x8000 | K (never 0: an empty slot has k1 == 0)
 *   16 bytes of payload: u32 positions { [0].lower, [1].lower, size, 0 }; u64 positions (below 2^40, what 288 GB hold):
 *   [0].lower | [1].lower << 40 in the first 8 bytes and the next, [1].lower >> 24 | size << 16. */
#define SIGAX_DEEP_KMAX 56
#define SIGAX_DEEP_KMIN 16

struct FmStrand {
  const uint32_t* granules;  /* n_granules x 16 u32 */
  const unsigned char* sa;    /* row table: n entries of sa_bits bits (+ 8 bytes of padding), or NULL */
  const unsigned char* text;  /* stretch text: C['A'] rows of text_stride bytes, or NULL (then only countdowns use `sa`) */
  const unsigned long long* xmap;  /* direct map of this strand as extension index (n_strings entries), or NULL */
  unsigned int sa_bits, ld_bits, t_bits, text_stride;  /* symbols per entry = (sa_bits - ld_bits - t_bits) / 2 */
  const uint32_t* gran2;     /* (n / 64 + 1) x 32 u32, or NULL */
  const unsigned long long* super2;  /* [n_super][20], wide mode with two-step tables; else NULL */
  const unsigned long long* super;    /* [n_super][4] absolute A,C,G,T counts at each superblock start (wide mode) */
  const void* start;                  /* start table of the finder, or NULL */
  const void* deep;                   /* deep start table (hash of this strand's K-mers), or NULL */
  unsigned long long deep_slots;      /* its slots (32 bytes each) */
  unsigned int deep_k, deep_pad;      /* its K */
  unsigned long long n;             /* symbols */
  unsigned long long C[5];           /* FMIndex::_pred (src/fmindex.cpp:156-160) */
  unsigned long long total[5];         /* symbol totals = Occ(c, n-1) */
};

/* chain_cnt word written by the block finder for every (read, chain) */
#define SIGAX_CC_COUNT_MASK 0x0FFFFFFFu
#define SIGAX_CC_CONTAIN    0x40000000u  /* a containment block sits in the chain's last slot */
#define SIGAX_CC_SUBSTRING  0x80000000u  /* this find set OverlapResult::substring */

#endif
