#!/bin/bash
# A/B on the GPU box: sweep2.sh "<ENV=val ...>:<variant>" ...   e.g. "KSA_NO_REUSE=1:rm"
cd "$(dirname "$0")/.."
for spec in "$@"; do
  envs=${spec%%:*}; v=${spec#*:}
  env $envs KSA_LIB=$PWD/variants/libksa_$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print('%-28s kernel %.3f ms  step %.3f ms  %.1f MFFT/s  frac %.3f  vgpr %d grid %d' % ('$spec', r['avg_kernel_ms'], d['ms_per_step'], d['value']/1e6, r['frac'], r['vgprs'], r['grid']))
"
done
