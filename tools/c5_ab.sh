#!/bin/bash
# GPU box: config-5 bench, A/B over libraries given as arguments (paths relative to the repo root; "main" = in-tree)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lib in "$@"; do
  tools/with_lib.sh $lib timeout -k 10 200 python3 bench.py --config ${CFG:-5} --steps 10 --warmup 2 --no-cpu > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
  python3 -c "
import json
d=json.load(open('/tmp/ab.json'))
print('%-34s %.3f MFFT/s  ms/step %.3f  kern %.3f ms frac %.3f' % ('$lib', d['value']/1e6, d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
done
