"""64-bit position path (indexes with >= 2^32 symbols: BASELINE configs[4]) on small inputs: SIGAX_FORCE_WIDE=1 selects
the WIDE kernels, and a library variant built with superblocks every 2^12 symbols exercises the superblock counters.
Runs the parity suite in a subprocess because the library is chosen at load time."""
import os
import subprocess
import sys

import pytest

from tests.fixtures import ROOT

pytestmark = pytest.mark.gpu


def _run_parity(env_extra, select, seeds=None):
    env = dict(os.environ, **env_extra)
    if seeds:  # cases of the randomised suite by seed
        what = [os.path.join(ROOT, "tests", "test_gpu_random.py") + "::test_random_case_bit_exact[%d]" % k for k in seeds]
    else:
        what = [os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", select]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q"] + what,
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_wide_kernels_bit_exact():
    _run_parity({"SIGAX_FORCE_WIDE": "1"}, "occ or toy or rep or dup or corner or kmer")


def test_wide_kernels_with_many_superblocks():
    from siga_amd import build as sbuild
    lib = sbuild.build_libsigax(out=os.path.join(ROOT, "build", "libsigax_super12.so"), defines=("SIGAX_SUPER_SHIFT=12",))
    _run_parity({"SIGAX_FORCE_WIDE": "1", "SIGAX_LIB": lib}, "occ or toy or dup or kmer")


def test_one_step_finder_and_extractor_bit_exact():
    """Indexes of 1.6 G symbols and more carry no two-step table: the one-step finder (k_find_n) and the extractor without
    double rounds are what they run.  SIGAX_TWO_STEP=0 selects that path on the small fixtures."""
    _run_parity({"SIGAX_TWO_STEP": "0"}, "hits_and_asqg or non_acgt or duplicate or in_flight")


def test_two_step_table_limit_env():
    """SIGAX_TWO_STEP_MAX_SYMBOLS below the fixture size must fall back to the one-step path, above it use the table:
    both give the same answers (above it is what every other test runs)."""
    _run_parity({"SIGAX_TWO_STEP_MAX_SYMBOLS": "1000"}, "toy or tiny")


def test_arena_regrow_and_rerun():
    """Tiny first sizes for the final-block arena, the edge buffer and the general kernel's pool: sigax_batch_finish
    must grow each and repeat the run, and the repeated run must give the oracle's bytes (k_edges stays inside its
    buffers on the overflowing run)."""
    _run_parity({"SIGAX_TEST_FIN_CAP": "64", "SIGAX_TEST_EDGE_CAP": "16"}, "hits_and_asqg and (toy or dup or rep)")
    _run_parity({"SIGAX_TEST_POOL_CAP": "24", "SIGAX_GENERAL_ONLY": "1", "SIGAX_TEST_FIN_CAP": "100"}, "hits_and_asqg and (toy or rep or corner)")


def test_candidate_slots_sized_by_the_batch_and_overlap_in_pieces():
    """The candidate arena gives a chain the slots the longest chain so far needed, not one per overlap length: a first try
    of two slots (SIGAX_CAND_CAP=2) makes every run overflow once and repeat itself with what its chains reported;
    SIGAX_CAND_CAP=worst is the round-2 sizing.  sigax_overlap_batch in pieces of 37 reads (what a read set beyond one
    device workspace gets) joins the pieces' results in read order.  All give the oracle's bytes."""
    _run_parity({"SIGAX_CAND_CAP": "2"}, "hits_and_asqg or non_acgt or duplicate or deep or in_flight")
    _run_parity({"SIGAX_CAND_CAP": "2", "SIGAX_FIND_COOP": "1", "SIGAX_FORCE_WIDE": "1"}, "hits_and_asqg and (toy or dup or rep or ragged)")
    _run_parity({"SIGAX_TEST_PIECE": "37", "SIGAX_CAND_CAP": "worst"}, "hits_and_asqg or non_acgt or duplicate or deep")
    _run_parity({"SIGAX_TEST_PIECE": "37", "SIGAX_CAND_CAP": "4"}, None, seeds=(1, 3, 8))


def test_cooperative_finder_bit_exact():
    """k_find_c2 (two-step lines fetched eight lanes per line through LDS; what indexes of 2^30 symbols and more run) forced
    on the small fixtures: same bytes as the oracle, ragged reads, non-ACGT bases, duplicates and deep coverage included."""
    _run_parity({"SIGAX_FIND_COOP": "1"}, "hits_and_asqg or non_acgt or duplicate or in_flight or deep or mid")


def test_finder_start_table_bit_exact():
    """Chains of the block finder may start twelve symbols in, from a table of all 12-mers per strand (built at open for
    indexes of 2^22 symbols and more; forced here): same blocks, same hits order, same count of rank evaluations as the walk --
    also where the range empties inside the first twelve symbols, reads hold non-ACGT bases or are shorter than twelve, and
    with min-overlap below twelve (no table then)."""
    _run_parity({"SIGAX_FIND_START": "1"}, "hits_and_asqg or non_acgt or duplicate or deep or in_flight")
    _run_parity({"SIGAX_FIND_START": "1", "SIGAX_FIND_COOP": "1"}, "hits_and_asqg or non_acgt or deep")
    _run_parity({"SIGAX_FIND_START": "1", "SIGAX_TWO_STEP": "0"}, None, seeds=(1, 2, 3, 5, 8, 13, 21))
    _run_parity({"SIGAX_FIND_START": "1", "SIGAX_FIND_COOP": "1", "SIGAX_FORCE_WIDE": "1", "SIGAX_READ_ORDER": "1"}, None, seeds=(2, 3, 8, 21))


def test_deep_start_table_bit_exact():
    """The deep start table (fm_layout.h; every resident index of the suite gets one for the min-overlap it is asked) in the
    forms the suite does not reach by itself: switched off (SIGAX_FIND_DEEP=0: every chain walks from the 12-mer table or
    the first symbol, as round 3 did), tiny K with short min-overlaps (SIGAX_DEEP_K), a table filled to 90 % (long probe
    sequences), the cooperative and the one-step finder, 64-bit positions (40-bit packed entries, superblocks every 2^12
    symbols), reads with substitutions, duplicates and non-ACGT bases (chains whose K-mer is not in the table walk)."""
    _run_parity({"SIGAX_FIND_DEEP": "0"}, "hits_and_asqg or non_acgt or deep")
    _run_parity({"SIGAX_DEEP_K": "5"}, "hits_and_asqg or non_acgt")
    _run_parity({"SIGAX_DEEP_K": "13", "SIGAX_DEEP_LOAD": "90"}, None, seeds=(1, 2, 3, 8, 13, 21))
    _run_parity({"SIGAX_DEEP_LOAD": "90", "SIGAX_FIND_COOP": "1"}, "hits_and_asqg or non_acgt or in_flight")
    _run_parity({"SIGAX_FORCE_WIDE": "1", "SIGAX_FIND_COOP": "1", "SIGAX_DEEP_K": "33"}, None, seeds=(2, 8, 21))
    _run_parity({"SIGAX_TWO_STEP": "0"}, "deep_start")
    from siga_amd import build as sbuild
    lib = sbuild.build_libsigax(out=os.path.join(ROOT, "build", "libsigax_super12.so"), defines=("SIGAX_SUPER_SHIFT=12",))
    _run_parity({"SIGAX_FORCE_WIDE": "1", "SIGAX_LIB": lib}, "deep_start or toy or non_acgt")


def test_sixteen_lane_groups_bit_exact():
    """SIGAX_FX_16=1: branching items of at most 16 blocks run four to a wave (a launch of its own between the strict and the
    branching 32-lane one; an option, off by default: DESIGN.md 4.3).  Read sets with substitutions, duplicates, repeats."""
    _run_parity({"SIGAX_FX_16": "1"}, None, seeds=(1, 2, 3, 5, 8, 13, 21, 22, 23, 24))
    _run_parity({"SIGAX_FX_16": "1", "SIGAX_FORCE_WIDE": "1"}, "hits_and_asqg or deep")


def test_filter_extract_grid_does_not_change_results():
    """A batch object sizes the filter/extract grid by its last run (sigax_batch_finish: three workgroups per CU once
    filter/extract outlasts the finder, two and a half otherwise): persistent waves draw items from one
    counter, so any grid gives the same bytes.  One workgroup, and far more workgroups than items' waves, on read sets
    with substitutions, duplicates and repeats, and on batch objects that run several times."""
    _run_parity({"SIGAX_FX_GRID": "1"}, None, seeds=(1, 3, 8, 21))
    _run_parity({"SIGAX_FX_GRID": "4096", "SIGAX_FX_GRID64": "4096"}, None, seeds=(2, 5, 13, 22))
    _run_parity({"SIGAX_FX_GRID": "3"}, "in_flight or deep_coverage or duplicate")


def test_finder_measured_by_the_batch_object():
    """SIGAX_COOP_TUNE_MIN=0 puts the small fixtures in the range where a batch object times both finders and keeps the
    faster (sigax_api.cpp: want_coop): the runs of one batch object go per lane, per lane, cooperative, cooperative, then
    the choice -- same bytes throughout; batch objects in flight tune side by side; a batch whose reads carry ids starts
    over."""
    _run_parity({"SIGAX_COOP_TUNE_MIN": "0"}, "finder_choice or in_flight or any_order or deep_coverage")
    _run_parity({"SIGAX_COOP_TUNE_MIN": "0", "SIGAX_SUBBATCHES": "3"}, None, seeds=(1, 8, 21))


def test_correct_without_the_kmer_prefix_table():
    """`siga correct`'s k-mer lookups start from the interval of their last twelve bases (a table of all 12-mers, built on
    first use); SIGAX_KMER_PREFIX=0 walks every step as the reference does.  Same files either way, 32- and 64-bit positions."""
    _run_parity({"SIGAX_KMER_PREFIX": "0", "SIGAX_KMER_TABLE": "0"}, "correct")
    _run_parity({"SIGAX_FORCE_WIDE": "1"}, "correct")
    # ... and without the table of the reads' distinct k-mers (one lookup per k-mer: what every other correction test runs),
    # the prefix table + walk of round 3
    _run_parity({"SIGAX_KMER_TABLE": "0"}, "correct")
    _run_parity({"SIGAX_KMER_TABLE": "0", "SIGAX_FORCE_WIDE": "1"}, "correct_matches")


def test_locality_order_of_the_batch_bit_exact():
    """The finder may walk a batch in its locality order (minimizer keys + one radix sort per batch, sigax_order_reads; on by
    default from 2^30 symbols): forced on for the small fixtures, per-lane and cooperative finder, 32- and 64-bit positions,
    several sub-batches.  Same bytes as the oracle."""
    _run_parity({"SIGAX_READ_ORDER": "1"}, "hits_and_asqg or non_acgt or duplicate or deep or in_flight")
    _run_parity({"SIGAX_READ_ORDER": "1", "SIGAX_FIND_COOP": "1", "SIGAX_FORCE_WIDE": "1", "SIGAX_SUBBATCHES": "3"}, None, seeds=(1, 2, 8, 21))


def test_extractor_forms_without_the_row_tables():
    """The extractor's tables (fm_layout.h) are accelerators.  With the bit-packed row table + the stretch text,
    filter/extract reads a block's next symbols off its read's text -- the first ones from the row-table entry itself when
    memory allows (what every other test runs), with bare entries (SIGAX_ROW_SYMS=0: what BASELINE configs[4] at full size
    gets) from the text alone.  An index too big for a row table gets DIRECT MAPS instead (SIGAX_XMAP=1 forces them: a
    single-row block names its target read itself through the .sai tables; reads with non-ACGT bases fall back to the row
    table).  With SIGAX_LOOKAHEAD=0 there is no text: the rounds come from two-step lines (or one-step granules) and only
    the countdown uses the row table; with SIGAX_ROWEND=0 nothing exists and the extractor walks as the reference does.
    Every form gives the oracle's bytes on read sets with substitutions, duplicates and substrings, 32- and 64-bit
    positions."""
    _run_parity({"SIGAX_XMAP": "1"}, None, seeds=(1, 2, 3, 5, 8, 13, 21, 22, 23, 24))
    _run_parity({"SIGAX_XMAP": "1", "SIGAX_FORCE_WIDE": "1", "SIGAX_FIND_COOP": "1"}, "hits_and_asqg or non_acgt or duplicate or deep or in_flight")
    _run_parity({"SIGAX_ROW_SYMS": "0"}, None, seeds=(1, 2, 3, 5, 8, 13, 21))
    _run_parity({"SIGAX_ROW_SYMS": "0", "SIGAX_FORCE_WIDE": "1"}, "hits_and_asqg or non_acgt or duplicate or deep")
    _run_parity({"SIGAX_ROW_SYMS": "3"}, None, seeds=(1, 2, 5, 8))
    _run_parity({"SIGAX_LOOKAHEAD": "0"}, None, seeds=(1, 2, 5, 8, 13, 21))
    _run_parity({"SIGAX_LOOKAHEAD": "0", "SIGAX_TWO_STEP": "0", "SIGAX_FORCE_WIDE": "1"}, None, seeds=(2, 3, 8))
    _run_parity({"SIGAX_ROWEND": "0"}, None, seeds=(1, 2, 5, 8, 13))
    _run_parity({"SIGAX_ROWEND": "0", "SIGAX_TWO_STEP": "0"}, "hits_and_asqg or non_acgt or duplicate or deep")
    _run_parity({"SIGAX_FORCE_WIDE": "1"}, None, seeds=(2, 3, 8, 21))
