#!/bin/bash
# run on the GPU box: bench every variants/libksa_*.so given as args, print kernel ms
cd "$(dirname "$0")/.."
for v in "$@"; do
  KSA_LIB=$PWD/variants/libksa_$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print('%-8s kernel %.3f ms  step %.3f ms  %.1f MFFT/s  frac %.3f  vgpr %d grid %d' % ('$v', r['avg_kernel_ms'], d['ms_per_step'], d['value']/1e6, r['frac'], r['vgprs'], r['grid']))
"
done
