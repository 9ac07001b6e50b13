#!/bin/bash
# GPU box: two-frames-per-workgroup packed kernel vs the one-frame kernel, spectrum stage only.
# Needs the experiments build (tools/variants.sh exp): KSA_PAIR_ALL / KSA_NO_PAIR are read by it alone.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for n in 1024 2048 4096; do
  for q in 0.5 0.25 0.1; do
    fr=$((134217728 / n / 8))
    a=$(KSA_PAIR_ALL=1 tools/with_lib.sh variants/libksa_exp.so timeout -k 10 120 python3 tools/bench_one.py $n $q hanning $((n*8)) $fr | awk '{print $4" ms "$6" MFFT/s"}')
    b=$(KSA_NO_PAIR=1 tools/with_lib.sh variants/libksa_exp.so timeout -k 10 120 python3 tools/bench_one.py $n $q hanning $((n*8)) $fr | awk '{print $4" ms "$6" MFFT/s"}')
    echo "N=$n q=$q frames=$fr  pair $a | single $b"
  done
done
