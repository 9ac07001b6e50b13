#!/bin/bash
# GPU box: bench.py --config $CFG with each library of $LIBS ("main" = the in-tree product library) swapped in, $REPS times alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do
    if [ $lib = main ]; then pre=""; else pre="tools/with_lib.sh $lib"; fi
    $pre timeout -k 10 200 python3 bench.py --config ${CFG:-2} --steps 20 --warmup 3 --no-cpu --no-secondary $BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
    python3 -c "
import json
d=json.load(open('/tmp/ab.json')); r=d['roofline']
print('cfg ${CFG:-2} %-28s %.3f MFFT/s  ms/step %.4f  kern %.4f ms frac %.4f flop %.3f vgprs %s lds %s grid %s clock %s' % ('$lib', d['value']/1e6, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['flop_frac'], r.get('vgprs'), r.get('lds_bytes'), r.get('grid'), r.get('shader_clock_ghz_live')))"
  done
done
