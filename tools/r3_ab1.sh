#!/bin/bash
# GPU box, round 3 session 1: quad (DPP) exchange at N = 64 and in-loop twiddles for the 32-point kernel, A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
L=gpurun_out/r3_ab1.log
: > $L
echo "== parity of the quad-exchange build (N = 64 cases)" >> $L
KSA_LIB=$R/variants/libksa_quad.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_random.py -m gpu -q -x -k "64 or quick or every_plan or random" >> $L 2>&1 || { echo "quad parity FAILED" >> $L; }
echo "== config 4: main vs quad" >> $L
CFG=4 tools/cfg_ab.sh main variants/libksa_quad.so >> $L 2>&1
echo "== parity of the twl3 build (N = 8192 / 16384 cases)" >> $L
KSA_LIB=$R/variants/libksa_twl3.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -k "16384 or 8192 or fm_n or every_plan" >> $L 2>&1 || { echo "twl3 parity FAILED" >> $L; }
echo "== config 3: main vs twl3 vs twl6" >> $L
CFG=3 tools/cfg_ab.sh main variants/libksa_twl3.so variants/libksa_twl6.so >> $L 2>&1
echo "== N = 8192, 50 %: main vs twl3" >> $L
for lib in "" variants/libksa_twl3.so; do
  if [ -z "$lib" ]; then unset KSA_LIB; else export KSA_LIB=$R/$lib; fi
  timeout -k 10 200 python3 tools/bench_one.py 8192 0.5 hanning 65536 4096 >> $L 2>&1
done
unset KSA_LIB
echo "== config 2 / 5 today (main)" >> $L
CFG=2 tools/cfg_ab.sh main >> $L 2>&1
CFG=5 BENCH_ARGS="" tools/c5_ab.sh main >> $L 2>&1
cat $L
