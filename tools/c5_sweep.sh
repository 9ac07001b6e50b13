#!/bin/bash
# GPU box: config-5 bench over first-stage scratch sizes (frames per chunk = MB / 14.5); experiments build (tools/variants.sh exp)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/c5sweep
for mb in ${@:-64 128 256 512 1024 4096}; do
  KSA_FS_SCRATCH_MB=$mb $R/tools/with_lib.sh variants/libksa_exp.so timeout -k 10 200 python3 $R/bench.py --config 5 --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/c5sweep/mb$mb.json 2> $R/gpurun_out/c5sweep/mb$mb.err || echo "mb $mb failed"
  python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/c5sweep/mb$mb.json'))
print('scratch %5d MB: %.2f MFFT/s  ms/step %.2f  frac %.3f' % ($mb, d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"
done
