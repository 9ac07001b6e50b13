#!/bin/bash
# GPU box: config-3 bench (N=16384 kaiser scan) and an N=8192 zeroSpan shape, 32-point plan vs 16-point plan (KSA_PLAN16=1),
# for each library given ("main" = in-tree; KSA_PLAN16 needs an experiments build: tools/variants.sh exp)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lib in "$@"; do
  for plan in 32 16; do
    if [ $plan = 16 ]; then export KSA_PLAN16=1; else unset KSA_PLAN16; fi
    tools/with_lib.sh $lib timeout -k 10 200 python3 bench.py --config 3 --steps 10 --warmup 2 --no-cpu > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib plan$plan failed"; tail -3 /tmp/ab.err; continue; }
    python3 -c "
import json
d=json.load(open('/tmp/ab.json'))
print('%-30s plan%s  C3 %.3f MFFT/s  kern %.3f ms  vgprs %d' % ('$lib', '$plan', d['value']/1e6, d['roofline']['avg_kernel_ms'], d['roofline']['vgprs']))"
    tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py 8192 0.5 hanning 65536 8192 2>/dev/null | tail -1
    tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py 16384 0.25 hanning 131072 4096 2>/dev/null | tail -1
  done
done
unset KSA_PLAN16
