#!/bin/bash
# CPU code under AddressSanitizer + UBSan.  The oracle (test infrastructure): builds a sanitized copy of liboracle.so,
# runs the oracle's own tests against it, and puts the regular build back.  (GPU sanitizers are not available on the pool.)
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -march=x86-64-v3 -std=gnu11 -shared -fPIC -pthread \
    -o /tmp/liboracle_asan.so oracle/fbg_oracle.c
cp oracle/liboracle.so /tmp/liboracle_orig.so
trap 'cp /tmp/liboracle_orig.so oracle/liboracle.so' EXIT
cp /tmp/liboracle_asan.so oracle/liboracle.so
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_oracle.py tests/test_wide_chain_model.py -x -q
cp /tmp/liboracle_orig.so oracle/liboracle.so
# the host program's FASTA reader and xGFA writer (no GPU needed) under the same sanitizers
make -C founderblockgraphs_amd/csrc -B ../fbg_host_selftest SANITIZE="-fsanitize=address,undefined -fno-omit-frame-pointer" > /dev/null
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 python -m pytest tests/test_host_io.py -x -q
make -C founderblockgraphs_amd/csrc -B ../fbg_host_selftest > /dev/null
