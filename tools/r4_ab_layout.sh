#!/bin/bash
# GPU box, round 4: LDS layout of the exchanges and the form of the exchange reads, same box, every library twice.
#   main     transposed / natural layout (Plan<N>), exchange reads as single ds_read_b64 (relaxed atomic loads)
#   xnoatom  the same layout, plain loads (hipcc merges them into ds_read2_b64 / ds_read2st64_b64)
#   pad16    one pad per 16 (rounds 1-3), plain loads;  pad16a: one pad per 16, single ds_read_b64
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L=gpurun_out/r4_ab_layout.txt
: > $L
LIBS="main variants/libksa_xnoatom.so variants/libksa_pad16.so variants/libksa_pad16a.so"
for c in 2 4; do CFG=$c tools/cfg_ab.sh $LIBS >> $L 2>&1; done
for a in "1024 0.5 hanning 8192 65536" "512 0.5 hanning 4096 131072" "128 0.5 hanning 1024 262144" "64 0.5 hanning 512 524288" \
         "32 0.5 hanning 256 524288" "2048 0.5 hanning 16384 32768" "4096 0.25 hanning 32768 16384" "4096 0.1 hanning 32768 16384"; do
  for rep in 1 2; do for lib in $LIBS; do
    tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $a 2>&1 | tail -1 >> $L
  done; done
done
CFG=5 tools/cfg_ab.sh main variants/libksa_pad16.so >> $L 2>&1
cat $L
