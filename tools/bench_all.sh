#!/bin/bash
# Run on the GPU box: one bench line per BASELINE configuration -> gpurun_out/<tag>/bench_c<k>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-bench}
shift || true
mkdir -p $R/gpurun_out/$tag
for c in 2 3 4 5; do
  timeout -k 10 400 python3 $R/bench.py --config $c --steps 10 --warmup 2 "$@" > $R/gpurun_out/$tag/bench_c$c.json 2> $R/gpurun_out/$tag/bench_c$c.err || echo "config $c failed rc=$?"
  cut -c1-400 $R/gpurun_out/$tag/bench_c$c.json
done
