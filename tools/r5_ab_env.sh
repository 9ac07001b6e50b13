#!/bin/bash
# GPU box: bench.py --config $CFG with library $LIB swapped in, alternating with and without the environment switch $SW=1
# (an -DKSA_EXPERIMENTS build reads it): $REPS times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in $(seq 1 ${REPS:-2}); do
  for on in 1 0; do
    if [ $on = 1 ]; then export $SW=1; else unset $SW; fi
    tools/with_lib.sh $LIB timeout -k 10 200 python3 bench.py --config ${CFG:-4} --steps 20 --warmup 3 --no-cpu --no-secondary $BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err || { echo "failed"; tail -3 /tmp/ab.err; continue; }
    python3 -c "
import json
d=json.load(open('/tmp/ab.json')); r=d['roofline']
print('cfg ${CFG:-4} %-24s %s=%d: %.3f MFFT/s  ms/step %.4f  kern %.4f ms frac %.4f flop %.3f clock %s' % ('$LIB', '$SW', $on, d['value']/1e6, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['flop_frac'], r.get('shader_clock_ghz_live')))"
  done
done
unset $SW
