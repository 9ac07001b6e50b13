#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L="main variants/libksa_fsmall.so"
for c in 4 5 2; do CFG=$c BENCH_ARGS=--no-secondary tools/cfg_ab.sh $L; done
for lib in $L; do
  for shape in "64 0.5 hanning 512 262144" "128 0.5 hanning 1024 131072" "32 0.5 hanning 256 524288" "256 0.5 hanning 2048 65536" "512 0.5 hanning 4096 32768" "16 0.5 hanning 128 1048576"; do
    echo -n "$lib $shape : "; tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $shape 2>&1 | tail -1
  done
done
