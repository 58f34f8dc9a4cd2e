#!/bin/bash
# A/B of library builds on one box: tools/ab_pass.sh [pass_bench args --] lib_dir... — tools/pass_bench.py for each (SCANRS_AMD_LIB), twice round
ARGS="1000000 100 0"
for L in "$@" "$@"; do
  echo "== $L"
  SCANRS_AMD_LIB=$PWD/$L/libscanrs_amd.so timeout 250 python3 tools/pass_bench.py $ARGS 2>&1 | tail -2
done
