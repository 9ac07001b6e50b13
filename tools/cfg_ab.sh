#!/bin/bash
# GPU box: bench.py --config $CFG (default 2), A/B over the libraries given ("main" = in-tree); every library twice
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
for lib in "$@"; do
  tools/with_lib.sh $lib timeout -k 10 200 python3 bench.py --config ${CFG:-2} --steps 20 --warmup 3 --no-cpu $BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
  python3 -c "
import json
d=json.load(open('/tmp/ab.json'))
print('cfg %s %-34s %.3f MFFT/s  ms/step %.3f  kern %.4f ms frac %.4f' % ('${CFG:-2}', '$lib', d['value']/1e6, d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
done
done
