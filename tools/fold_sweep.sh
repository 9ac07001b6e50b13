#!/bin/bash
# run on the GPU box: spectrum-stage time per size and hop, fold mode as template constant vs run-time branch
cd "$(dirname "$0")/.."
for n in 16 32 64 128 256 512 1024 2048 4096 8192 16384; do
  for q in 0.5 0.25 0.1; do
    fr=$((16777216 / n))
    a=$(timeout -k 10 120 python tools/bench_one.py $n $q hanning $((n*8)) $fr | awk '{print $4}')
    b=$(KSA_GENERIC_FOLD=1 timeout -k 10 120 python tools/bench_one.py $n $q hanning $((n*8)) $fr | awk '{print $4}')
    echo "N=$n q=$q templated $a ms generic $b ms"
  done
done
