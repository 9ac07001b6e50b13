#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for c in 4 3; do
  for rep in 1 2; do
    for tb in 256 128 64; do
      KSA_STITCH_TB=$tb tools/with_lib.sh variants/libksa_stb.so timeout -k 10 200 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu --no-secondary > /tmp/ab.json 2> /tmp/ab.err || { echo failed; tail -3 /tmp/ab.err; continue; }
      python3 -c "
import json
d=json.load(open('/tmp/ab.json'))
print('cfg $c stitch workgroup $tb: %.3f MFFT/s  ms/step %.4f  kern %.4f ms  step-kern %.4f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['ms_per_step']-d['roofline']['avg_kernel_ms']))"
    done
  done
done
