#!/bin/bash
# GPU box, round 5, first call: the new GPU tests, the overlap-mode A/B at configs 2 and 5, the cycle
# stamps of the 32-point kernel at config 3 (item 6; needs variants/libksa_stamps.so: empty .gpurunignore for this call),
# and one default bench line.   tools/r5_run1.sh [tests|ab|stamps|bench|all]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
what=${1:-all}
O=gpurun_out/r5a
mkdir -p $O
line() { python3 -c "
import json,sys
d=json.load(open('$1')); r=d['roofline']
print('%-28s %.3f MFFT/s  ms/step %.4f  kern %.4f ms  frac %.4f step %.4f  clk %s  overlap %s' % ('$2', d['value']/1e6, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['frac_step'], r.get('shader_clock_ghz_live'), d['config'].get('accumulate_overlap')))"; }
if [ $what = tests ] || [ $what = all ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_round5.py -m gpu -x -q > $O/pytest_r5.log 2>&1; echo "pytest rc $?" >> $O/pytest_r5.log
  tail -15 $O/pytest_r5.log
fi
if [ $what = ab ] || [ $what = all ]; then
  : > $O/ab_overlap.txt
  for rep in 1 2 3; do
    for ov in 0 1; do
      timeout -k 10 300 python3 bench.py --config 2 --overlap $ov --steps 30 --warmup 3 --no-cpu --no-secondary > $O/c2_ov${ov}_$rep.json 2> $O/c2_ov${ov}_$rep.err || { echo "c2 ov$ov failed"; tail -3 $O/c2_ov${ov}_$rep.err; }
      line $O/c2_ov${ov}_$rep.json "config 2 overlap=$ov rep $rep" | tee -a $O/ab_overlap.txt
    done
  done
  for rep in 1 2; do
    for ov in 0 1; do
      timeout -k 10 300 python3 bench.py --config 5 --overlap $ov --steps 10 --warmup 2 --no-cpu --no-secondary > $O/c5_ov${ov}_$rep.json 2> $O/c5_ov${ov}_$rep.err || { echo "c5 ov$ov failed"; tail -3 $O/c5_ov${ov}_$rep.err; }
      line $O/c5_ov${ov}_$rep.json "config 5 overlap=$ov rep $rep" | tee -a $O/ab_overlap.txt
    done
  done
  for ov in 0 1; do
    timeout -k 10 300 python3 bench.py --config 2 --fmt u8 --overlap $ov --steps 30 --warmup 3 --no-cpu --no-secondary > $O/c2u8_ov${ov}.json 2> $O/c2u8_ov${ov}.err || echo "c2 u8 ov$ov failed"
    line $O/c2u8_ov${ov}.json "config 2 u8 overlap=$ov" | tee -a $O/ab_overlap.txt
  done
fi
if [ $what = stamps ] || [ $what = all ]; then
  if [ -f variants/libksa_stamps.so ]; then
    rm -f /tmp/stamps.txt
    KSA_STAMPS_FILE=/tmp/stamps.txt tools/with_lib.sh variants/libksa_stamps.so timeout -k 10 200 python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu --no-secondary > /dev/null 2> /tmp/stamps.err || tail -5 /tmp/stamps.err
    tail -2 /tmp/stamps.txt | python3 -c "
import sys
names=['0 IQ + tap loads (issue, wait), window multiply','1 pass 0 (radix 32)','2 barrier in front of exchange 1','3 exchange 1: 32 stores + barrier','4 exchange-1 reads + 31 twiddle reads + pass 1 (radix 32)','5 exchange 2: barrier, 32 stores, barrier','6 exchange-2 reads + pass 2 (2 x radix 16) + fold','7 -','8 per-frame output stage','9 -']
for ln in sys.stdin:
    p=ln.split(':'); v=[float(x) for x in p[1].split()]; tot=sum(v)
    print(p[0]); [print('  %-58s %6.1f %%  %10.0f clk' % (n, 100*x/tot, x)) for n,x in zip(names,v) if x]
" | tee $O/c3_stamps.txt
  else
    echo "variants/libksa_stamps.so did not travel (is variants/ still in .gpurunignore?)"
  fi
fi
if [ $what = bench ] || [ $what = all ]; then
  ( time timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
  grep real $O/bench_default.time; cut -c1-600 $O/bench_default.json
fi
echo r5_run1 $what done
