#!/bin/bash
# GPU box, round 3: tests (+ a soak of the random sweeps), every BASELINE configuration's bench line, and the rocprofv3
# passes behind profiles/r03_*.   tools/r3_final.sh [tests|bench|prof|all]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
what=${1:-all}
mkdir -p gpurun_out/r3final
if [ $what = tests ] || [ $what = all ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3final/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3final/pytest.log
  tail -5 gpurun_out/r3final/pytest.log
  KSA_RANDOM_CASES=500 KSA_RANDOM_SEED=7 timeout -k 10 900 python3 -m pytest tests/test_gpu_random.py -m gpu -q > gpurun_out/r3final/soak.log 2>&1; echo "soak rc $?" >> gpurun_out/r3final/soak.log
  tail -3 gpurun_out/r3final/soak.log
fi
if [ $what = bench ] || [ $what = all ]; then
  ( time timeout -k 10 900 python3 bench.py > gpurun_out/r3final/bench_default.json 2> gpurun_out/r3final/bench_default.err ) 2> gpurun_out/r3final/bench_default.time
  grep real gpurun_out/r3final/bench_default.time; cut -c1-300 gpurun_out/r3final/bench_default.json
  for k in 2 3 4 5; do
    timeout -k 10 400 python3 bench.py --config $k > gpurun_out/r3final/bench_c$k.json 2> gpurun_out/r3final/bench_c$k.err || echo "bench c$k failed"
    timeout -k 10 300 python3 bench.py --config $k --fmt u8 --no-cpu > gpurun_out/r3final/bench_c${k}_u8.json 2> gpurun_out/r3final/bench_c${k}_u8.err || echo "bench c$k u8 failed"
    python3 -c "
import json
for f in ('bench_c$k.json','bench_c${k}_u8.json'):
    d=json.load(open('gpurun_out/r3final/'+f)); r=d['roofline']
    print('c$k %-16s %.3f MFFT/s %.1f GS/s ms/step %.3f kern %.3f ms frac %.4f step %.4f flop %.3f traffic %s cpu %s' % (f, d['value']/1e6, d['msamples_per_s']/1e3, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['frac_step'], r['flop_frac'], r['traffic'], d.get('cpu_baseline',{}).get('value')))"
  done
  # multi-rank rehearsals on this one GPU over gloo (functional record, not a scaling measurement)
  for spec in "2 2" "2 4" "3 2" "3 4" "5 2"; do
    set -- $spec
    KSA_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --config $1 --gpus $2 --steps 3 --warmup 1 --no-cpu $( [ $1 = 2 ] && echo "--frames 4096" ) $( [ $1 = 3 ] && echo "--passes 32" ) $( [ $1 = 5 ] && echo "--frames 128" ) > gpurun_out/r3final/gloo_c$1_g$2.json 2> gpurun_out/r3final/gloo_c$1_g$2.err || echo "gloo c$1 g$2 failed"
  done
fi
if [ $what = prof ] || [ $what = all ]; then
  for k in 2 3 4 5; do tools/profile_bench.sh r03_c$k --config $k > gpurun_out/r3final/prof_c$k.log 2>&1; done
  # the 75 %-overlap reuse kernel (RM = 4) at N = 4096: kernel trace of a plain spectrum-stage run
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_rm4/stats -- python3 $R/tools/bench_one.py 4096 0.25 hanning 32768 16384 > $R/gpurun_out/r3final/rm4.log 2>&1
  cd $R; tail -2 gpurun_out/r3final/rm4.log
fi
echo r3_final $what done
