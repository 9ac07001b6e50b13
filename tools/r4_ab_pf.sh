#!/bin/bash
# GPU box, round 4: config 2 with the next window's loads prefetched (KSA_PF=1 now fits: 155 VGPRs, 0 spills), the middle-pass
# twiddles in VGPRs (KSA_TWM_REGS=1: 165 VGPRs) and both (168 VGPRs, 2 spills), same box, every library twice.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L=gpurun_out/r4_ab_pf2.txt
: > $L
CFG=2 tools/cfg_ab.sh main variants/libksa_pf1.so variants/libksa_twm.so variants/libksa_twmpf.so >> $L 2>&1
for a in "4096 0.25 hanning 32768 16384" "4096 0.1 hanning 32768 16384" "2048 0.5 hanning 16384 32768" "1024 0.5 hanning 8192 2048"; do
  for rep in 1 2; do for lib in main variants/libksa_pf1.so variants/libksa_twm.so; do
    tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $a 2>&1 | tail -1 >> $L
  done; done
done
CFG=5 tools/cfg_ab.sh main variants/libksa_twm.so >> $L 2>&1
cat $L
