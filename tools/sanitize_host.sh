#!/bin/bash
# Host sanitizer builds ("ASan/UBSan/TSan host builds"), CPU container only: libsiga_host.so and the `siga` CLI
# built with g++ -fsanitize=address,undefined and again with -fsanitize=thread, and the CPU tests of the host side
# (tests/test_cpu_boundary.py: mmap chunk loader, sample sort, .sai parser/writer slices, gzip writer + line deflate coder,
# SA-IS and the bucket sorter, CLI option handling) run on each.  The GPU entry points are not reached without a GPU
# (sigax_index_open fails with SIGAX_E_DEVICE), so the HIP library itself stays the ordinary build.
# strict_memcmp=0: the suffix comparators hand memcmp the whole rest of the text as length (it stops at the first
# difference); the sanitizers' interceptor would otherwise touch-check all of it on every comparison (2 s -> hours).
# The native libraries are built BEFORE the runtime is preloaded (a compiler under LD_PRELOAD=libasan crawls).
#   bash tools/sanitize_host.sh [asan|tsan|all]   -> profiles/r04_sanitize_{asan,tsan}.log
set -u
cd "$(dirname "$0")/.."
what=${1:-all}
ROOT=$PWD
LIBDIR=$ROOT/siga_amd/lib
GCCDIR=$(dirname "$(gcc -print-file-name=libasan.so)")
python -c "
from siga_amd import build; build.build_all()
from oracle import pyoracle; pyoracle.build()" || exit 1
one() { # name, -fsanitize flags, runtime to preload, options env
  name=$1; flags=$2; rt=$3
  O=$ROOT/build/san_$name; mkdir -p $O
  cp -f $LIBDIR/libsigax.so $O/   # -rpath $ORIGIN of the sanitized host library finds the HIP library beside it
  COMMON="g++ -O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -Wall -Wno-sign-compare -pthread $flags"
  $COMMON -shared -o $O/libsiga_host.so siga_amd/host/siga_host.cpp -L$O -lsigax -lz -ldl -Wl,-rpath,'$ORIGIN' || return 1
  $COMMON -o $O/siga siga_amd/host/siga_main.cpp -L$O -lsiga_host -lsigax -lz -ldl -Wl,-rpath,'$ORIGIN' || return 1
  log=$ROOT/profiles/r04_sanitize_$name.log
  {
    echo "== $name: $COMMON"
    echo "== $(g++ --version | head -1); $(date -u +%F)"
    # python itself is not instrumented: the runtime is preloaded for the ctypes-loaded library; the CLI binary carries its own
    env LD_PRELOAD=$rt SIGA_HOST_LIB=$O/libsiga_host.so SIGA_CLI=$O/siga \
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:strict_memcmp=0:log_path=$O/asan \
        UBSAN_OPTIONS=print_stacktrace=1:log_path=$O/ubsan \
        TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 strict_memcmp=0 log_path=$O/tsan second_deadlock_stack=1" \
        timeout -k 10 1500 python -m pytest tests/test_cpu_boundary.py -q -p no:cacheprovider --timeout 600 2>&1 | tail -15
    echo "== reports:"
    ls $O | grep -E "^(asan|ubsan|tsan)\." || echo "(none)"
    for f in $O/asan.* $O/ubsan.* $O/tsan.*; do [ -f "$f" ] && { echo "---- $f"; head -80 "$f"; }; done
  } > $log 2>&1
  tail -25 $log
}
rm -f build/san_*/asan.* build/san_*/ubsan.* build/san_*/tsan.* 2>/dev/null
if [ "$what" = asan ] || [ "$what" = all ]; then one asan "-fsanitize=address,undefined" "$GCCDIR/libasan.so:$GCCDIR/libubsan.so"; fi
# TSan: a preloaded libtsan under CPython hangs (it wants to be first in the process), so the threaded paths are driven by a
# small C++ program instead (tools/tsan_driver.cpp: loader, name ranks, VT/ED formatters, gz writer, index builders, each on
# several thread counts with the outputs compared) plus the CLI itself
tsan() {
  O=$ROOT/build/san_tsan; mkdir -p $O; rm -f $O/tsan.*
  cp -f $LIBDIR/libsigax.so $O/
  COMMON="g++ -O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -Wall -Wno-sign-compare -pthread -fsanitize=thread"
  $COMMON -shared -o $O/libsiga_host.so siga_amd/host/siga_host.cpp -L$O -lsigax -lz -ldl -Wl,-rpath,'$ORIGIN' || return 1
  $COMMON -o $O/siga siga_amd/host/siga_main.cpp -L$O -lsiga_host -lsigax -lz -ldl -Wl,-rpath,'$ORIGIN' || return 1
  $COMMON -o $O/tsan_driver tools/tsan_driver.cpp -L$O -lsiga_host -lsigax -lz -ldl -Wl,-rpath,'$ORIGIN' || return 1
  log=$ROOT/profiles/r04_sanitize_tsan.log
  D=$(mktemp -d)
  {
    echo "== tsan: $COMMON"
    echo "== $(g++ --version | head -1); $(date -u +%F)"
    export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 strict_memcmp=0 log_path=$O/tsan second_deadlock_stack=1"
    timeout -k 10 1200 $O/tsan_driver $D; echo "tsan_driver rc=$?"
    cp $D/tsan_reads.fa $D/cli.fa
    (cd $D && timeout -k 10 600 $O/siga index --cpu -t 4 cli.fa; echo "siga index --cpu -t 4 rc=$?"; timeout -k 10 600 $O/siga index --cpu -a sais -t 4 -p cli_sais cli.fa; echo "siga index -a sais rc=$?")
    echo "== reports:"
    ls $O | grep -E "^tsan\." || echo "(none)"
    for f in $O/tsan.*; do [ -f "$f" ] && { echo "---- $f"; head -120 "$f"; }; done
  } > $log 2>&1
  rm -rf $D
  tail -25 $log
}
if [ "$what" = tsan ] || [ "$what" = all ]; then tsan; fi
