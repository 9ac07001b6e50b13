#!/bin/bash
# GPU box, round 5: tests (+ a soak of the random sweeps), every BASELINE configuration's bench line (c64 and u8), gloo
# multi-rank rehearsals and the in-process leg, the rocprofv3 passes behind profiles/r05_*, the host-pointer latencies.
#   tools/r5_final.sh [tests|bench|prof|wide|all]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
what=${1:-all}
O=gpurun_out/r5final
mkdir -p $O
if [ $what = tests ] || [ $what = all ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -4 $O/pytest.log
  KSA_RANDOM_CASES=500 KSA_RANDOM_SEED=7 timeout -k 10 900 python3 -m pytest tests/test_gpu_random.py -m gpu -q > $O/soak.log 2>&1; echo "soak rc $?" >> $O/soak.log
  tail -3 $O/soak.log
fi
if [ $what = bench ] || [ $what = all ]; then
  ( time timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
  grep real $O/bench_default.time; cut -c1-300 $O/bench_default.json
  for k in 2 3 4 5; do
    timeout -k 10 400 python3 bench.py --config $k > $O/bench_c$k.json 2> $O/bench_c$k.err || echo "bench c$k failed"
    timeout -k 10 300 python3 bench.py --config $k --fmt u8 --no-cpu > $O/bench_c${k}_u8.json 2> $O/bench_c${k}_u8.err || echo "bench c$k u8 failed"
    python3 -c "
import json
for f in ('bench_c$k.json','bench_c${k}_u8.json'):
    d=json.load(open('$O/'+f)); r=d['roofline']
    print('c$k %-16s %.3f MFFT/s %.1f GS/s ms/step %.3f kern %.3f ms frac %.4f step %.4f flop %.3f traffic %s cpu %s' % (f, d['value']/1e6, d['msamples_per_s']/1e3, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['frac_step'], r['flop_frac'], r['traffic'], d.get('cpu_baseline',{}).get('value')))"
  done
  # (multi-rank rehearsals over gloo and the in-process leg: tools/node_day.sh rehearsal)
  timeout -k 10 300 python3 tools/latency_host.py > $O/latency_host.txt 2>&1 || echo "latency_host failed"
fi
if [ $what = prof ] || [ $what = all ]; then
  for k in 2 3 4 5; do tools/profile_bench.sh r05_c$k --config $k > $O/prof_c$k.log 2>&1; done
fi
if [ $what = wide ]; then
  # the radix-32 / 64 first stages (N = 524288 / 1048576): spectrum-stage time and the kernel trace behind it
  cd /tmp && export TMPDIR=/tmp
  for n in 524288 1048576; do
    fr=$((33554432 / n))
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_wide_$n/stats -- python3 $R/tools/bench_one.py $n 0.25 hanning $((n*2)) $fr > $R/$O/wide_$n.log 2>&1
    tail -1 $R/$O/wide_$n.log
  done
  cd $R
fi
echo r5_final $what done
