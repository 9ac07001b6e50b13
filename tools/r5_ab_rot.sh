#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for f in c64 u8; do BENCH_ARGS="--no-secondary --fmt $f" CFG=4 tools/cfg_ab.sh variants/libksa_norot.so main; done
for lib in variants/libksa_norot.so main; do
  for shape in "64 0.5 hanning 512 262144" "64 0.25 hanning 512 262144" "128 0.1 hanning 1024 131072" "128 0.5 hanning 1024 131072" "32 0.1 hanning 256 524288" "32 0.5 hanning 256 524288" "256 0.1 hanning 2048 65536" "512 0.1 hanning 4096 32768" "16 0.1 hanning 128 1048576"; do
    echo -n "$lib $shape : "; tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $shape 2>&1 | tail -1
  done
done
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -4
