#!/bin/bash
# GPU box: spectrum64_kernel (8 x 8 plan of N = 64) variants at bench config 4, inside -DKSA_EXPERIMENTS builds
# (tools/variants.sh): $LIBS = "lib:KSA_NO_K64 ..." pairs (KSA_NO_K64=1 runs the 4 x 16 kernel of the same library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
LIBS=${LIBS:-"variants/libksa_k64s0.so:1 variants/libksa_k64s0.so:0 variants/libksa_k64s1.so:0 variants/libksa_k64r1.so:0"}
for rep in 1 2; do
  for v in $LIBS; do
    lib=${v%%:*}; no=${v##*:}
    if [ $no = 1 ]; then export KSA_NO_K64=1; else unset KSA_NO_K64; fi
    tools/with_lib.sh $lib timeout -k 10 200 python3 bench.py --config 4 --steps 20 --warmup 3 --no-cpu --no-secondary > /tmp/ab.json 2> /tmp/ab.err || { echo failed; tail -3 /tmp/ab.err; continue; }
    python3 -c "
import json
d=json.load(open('/tmp/ab.json'))
print('cfg 4 %-26s 8x8=%d: %.3f MFFT/s  ms/step %.4f  kern %.4f ms frac %.4f flop %.3f' % ('$lib', 1-$no, d['value']/1e6, d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['roofline']['flop_frac']))"
  done
done
unset KSA_NO_K64
for lib in ${SHAPE_LIBS:-variants/libksa_k64s0.so variants/libksa_k64s1.so}; do
  for shape in "64 0.5 hanning 512 262144" "64 0.25 hanning 512 262144" "64 0.1 kaiser 512 262144"; do
    echo -n "$lib $shape : "; tools/with_lib.sh $lib timeout -k 10 120 python3 tools/bench_one.py $shape 2>&1 | tail -1
  done
done
