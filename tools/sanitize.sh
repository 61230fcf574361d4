#!/bin/bash
# Sanitizer runs of the host-only C++ (csrc/leaderboard.cpp: worker-thread pre-filter of the bounded scan; csrc/bpe.cpp: parses a user-supplied
# vocabulary file, per-word cache shared by threads).  CPU only -- GPU AddressSanitizer is not available on this pool; the device code is covered by
# the parity tests.  1. `make sanitize`: g++ -fsanitize=thread and -fsanitize=address,undefined builds + tests/native/sanitize_driver.cpp under both.
# 2. the Python tests of the same code (tests/test_refine_scan.py incl. the threaded pre-filter cases, tests/test_tokenizer.py) against the
# ASan + UBSan library (GRIP_HOST_LIB swaps the host-only symbols; libasan preloaded into the interpreter).  Log: profiles/r06_sanitize.txt (r05: profiles/r05_sanitize.txt).
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/profiles/r06_sanitize.txt}
{
  echo "## $(date -u +%F) $(g++ --version | head -1)"
  echo "## make -C menghini-neurips23-code_amd/csrc sanitize"
  make -C "$R/menghini-neurips23-code_amd/csrc" sanitize 2>&1 | grep -v "^make\|^g++" ; echo "driver status: ${PIPESTATUS[0]}"
  echo "## pytest tests/test_refine_scan.py tests/test_tokenizer.py against libgrip_host_asan.so (ASan + UBSan)"
  cd "$R"
  LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    GRIP_HOST_LIB="$R/menghini-neurips23-code_amd/_san/libgrip_host_asan.so" GRIP_SCAN_THREADS=8 \
    python -m pytest tests/test_refine_scan.py tests/test_tokenizer.py -q -x -m "not gpu" -p no:cacheprovider 2>&1 | tail -4
  echo "pytest status: ${PIPESTATUS[0]}"
} 2>&1 | tee "$LOG"
