#!/bin/bash
# Fold mode as a template constant of spectrum_kernel vs a run-time branch in its window loop, per size and hop.
#   here:        tools/variants.sh foldconst -DKSA_FOLD_CONST_ALL ; tools/variants.sh foldgen -DKSA_FOLD_GENERIC
#   GPU box:     tools/fold_sweep.sh
cd "$(dirname "$0")/.."
for n in 16 32 64 128 256 512 1024 2048 4096 8192 16384; do
  for q in 0.5 0.25 0.1; do
    fr=$((16777216 / n))
    a=$(tools/with_lib.sh variants/libksa_foldconst.so timeout -k 10 120 python tools/bench_one.py $n $q hanning $((n*8)) $fr | awk '{print $4}')
    b=$(tools/with_lib.sh variants/libksa_foldgen.so timeout -k 10 120 python tools/bench_one.py $n $q hanning $((n*8)) $fr | awk '{print $4}')
    echo "N=$n q=$q constant $a ms run-time branch $b ms"
  done
done
