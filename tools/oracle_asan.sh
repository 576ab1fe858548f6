#!/bin/bash
# The CPU suite against an AddressSanitizer + UBSan build of the oracle's C sources (sanitizers run
# on the CPU build only; the GPU pool refuses sanitizer runs).  Usage: tools/oracle_asan.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
make -C oracle asan
export REART_ORACLE_LIB=$PWD/oracle/_build/liboracle_asan.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
exec python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
