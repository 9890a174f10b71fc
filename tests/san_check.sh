#!/bin/bash
# Everything that runs on the CPU under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md 5 "Race detection / sanitizers";
# CPU builds only -- the GPU side is never sanitized on this pool):
#   * oracle/*.c, oracle/kpn_baseline.cpp      make -C oracle SAN=1      -> oracle/_build_san/
#   * tests/emu (the kernels' lane programs)   make -C tests/emu SAN=1   -> tests/_build/libemu_san.so
#   * tests/cpp/kpn_tests.cpp + include/kpn.hpp make -C tests/cpp SAN=1  -> tests/_build/kpn_tests_san (the CPU blocks: `plumbing`,
#     which includes a block that throws mid-stream -- the reference's panic cascade, kpn.rs:17-29 -- and the bounded channels)
#   * the same program under ThreadSanitizer  make -C tests/cpp TSAN=1 -> tests/_build/kpn_tests_tsan (channels, block threads)
# then the CPU test suite (pytest -m "not gpu") with the sanitized oracle and lane programs loaded (libasan preloaded into python;
# leak checking off: the interpreter never frees its own arenas).  usage: bash tests/san_check.sh [logfile]
set -e -o pipefail   # a sanitizer abort or a failing test ends the script non-zero (tail / tee no longer hide the status)
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/profiles/r06_sanitizers.txt}
cd $R
make -C oracle -s SAN=1
make -C tests/emu -s SAN=1
make -C libredio_amd/csrc -s
make -C tests/cpp -s SAN=1
make -C tests/cpp -s TSAN=1
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
{
  echo "# tests/san_check.sh: $(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1)"
  echo "# flags: -fsanitize=address,undefined -fno-sanitize-recover=undefined (any report aborts the process: a clean log is a clean run)"
  echo "== kpn_tests_san plumbing (include/kpn.hpp CPU blocks, a block that throws mid-stream, bounded channels; the rings and the stream ordering of include/kpn_dev.hpp on stand-in device calls)"
  ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 tests/_build/kpn_tests_san plumbing 2>&1 | tail -5
  echo "== kpn_tests_tsan plumbing (ThreadSanitizer: the channels and every block thread; exit code 66 on a report)"
  TSAN_OPTIONS="exitcode=66" tests/_build/kpn_tests_tsan plumbing 2>&1 | grep -E "ThreadSanitizer|plumbing ok" | sort | uniq -c
  echo "== pytest -m 'not gpu' with oracle/_build_san and tests/_build/libemu_san.so"
  REDIO_ORACLE_SAN=1 LD_PRELOAD="$ASAN_LIB $UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python3 -m pytest tests -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -6
} | tee $LOG
