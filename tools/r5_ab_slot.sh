#!/bin/bash
# GPU box: output-stage slot combine variants at config 4 (N = 64) and at small-N zeroSpan shapes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
CFG=4 BENCH_ARGS=--no-secondary tools/cfg_ab.sh "$@"
for lib in "$@"; do
  for shape in "64 0.5 hanning 512 262144" "128 0.5 hanning 1024 131072" "32 0.5 hanning 256 524288" "256 0.5 hanning 2048 65536"; do
    echo -n "$lib $shape : "; tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $shape 2>&1 | tail -1
  done
done
