#!/bin/bash
# GPU box: staged frames (RM = -1) against the general path, one experiments build, switched by KSA_NO_STAGE
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
LIB=${LIB:-variants/libksa_stage.so}
for rep in 1 2 3; do
  for ns in 1 0; do
    if [ $ns = 1 ]; then export KSA_NO_STAGE=1; else unset KSA_NO_STAGE; fi
    tools/with_lib.sh $LIB timeout -k 10 200 python3 bench.py --config 4 --steps 20 --warmup 3 --no-cpu --no-secondary > /tmp/ab.json 2> /tmp/ab.err || { echo failed; tail -3 /tmp/ab.err; continue; }
    python3 -c "
import json
d=json.load(open('/tmp/ab.json'))
print('cfg 4 staged=%d: %.3f MFFT/s  ms/step %.4f  kern %.4f ms frac %.4f flop %.3f' % (1-$ns, d['value']/1e6, d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['roofline']['flop_frac']))"
  done
done
for ns in 1 0; do
  if [ $ns = 1 ]; then export KSA_NO_STAGE=1; else unset KSA_NO_STAGE; fi
  for shape in "64 0.5 hanning 512 262144" "64 0.1 kaiser 512 262144" "64 0.25 hanning 512 262144"; do
    echo -n "staged=$((1-ns)) $shape : "; tools/with_lib.sh $LIB timeout -k 10 120 python3 tools/bench_one.py $shape 2>&1 | tail -1
  done
done
unset KSA_NO_STAGE
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -q -x -k "64 or random or quick" 2>&1 | tail -4
