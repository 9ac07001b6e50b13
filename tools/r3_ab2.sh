#!/bin/bash
# GPU box, round 3 session 2: window multiply folded into the first radix-4 level (main) vs the plain form (nowf).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
L=gpurun_out/r3_ab2.log
: > $L
echo "== parity of main (fused window multiply)" >> $L
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x >> $L 2>&1 || echo "parity FAILED" >> $L
for c in 2 4 3 5; do
  echo "== config $c: main vs nowf" >> $L
  CFG=$c tools/cfg_ab.sh main variants/libksa_nowf.so >> $L 2>&1
done
echo "== N = 1024 / 64 / 16 spectrum stage" >> $L
for lib in "" variants/libksa_nowf.so; do
  if [ -z "$lib" ]; then unset KSA_LIB; else export KSA_LIB=$R/$lib; fi
  KSA_NO_PAIR=1 timeout -k 10 200 python3 tools/bench_one.py 1024 0.5 hanning 8192 65536 2>&1 | grep N= >> $L
  timeout -k 10 200 python3 tools/bench_one.py 4096 0.25 hanning 32768 16384 2>&1 | grep N= >> $L
  timeout -k 10 200 python3 tools/bench_one.py 4096 0.1 hanning 32768 16384 2>&1 | grep N= >> $L
done
cat $L
