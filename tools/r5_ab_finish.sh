#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L="variants/libksa_fbase.so variants/libksa_fslow.so variants/libksa_ffastonly.so variants/libksa_ffast.so"
for c in 2 3 4 5; do CFG=$c BENCH_ARGS=--no-secondary tools/cfg_ab.sh $L; done
for lib in $L; do
  for shape in "1024 0.5 hanning 8192 65536" "2048 0.5 hanning 16384 32768" "4096 0.25 hanning 32768 16384" "4096 0.1 hanning 32768 16384" "8192 0.5 hanning 65536 8192"; do
    echo -n "$lib $shape : "; tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $shape 2>&1 | tail -1
  done
done
