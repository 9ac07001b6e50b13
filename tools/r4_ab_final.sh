#!/bin/bash
# GPU box, round 4: the round-3 kernel shape (variants/libksa_pad16.so: one pad per 16, plain exchange loads, middle twiddles
# from LDS, no prefetch) against the round-4 default, every BASELINE configuration, c64 and u8, same box, twice each.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L=gpurun_out/r4_ab_final.txt
: > $L
for c in 2 3 4 5; do CFG=$c tools/cfg_ab.sh main variants/libksa_pad16.so >> $L 2>&1; done
CFG=2 BENCH_ARGS="--fmt u8" tools/cfg_ab.sh main variants/libksa_pad16.so >> $L 2>&1
cat $L
