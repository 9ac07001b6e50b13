#!/bin/bash
# GPU box, round 5, third call: clock diagnostic with hardware-id keyed stamps, the whole GPU suite, node-day rehearsal.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
what=${1:-all}
O=gpurun_out/r5c
mkdir -p $O
if [ $what = clock ] || [ $what = all ]; then
  timeout -k 10 300 python3 tools/clock_diag.py 2>&1 | tee $O/clock_diag.txt
fi
if [ $what = tests ] || [ $what = all ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -6 $O/pytest.log
fi
if [ $what = node ] || [ $what = all ]; then
  ( time tools/node_day.sh rehearsal ) > $O/node_day.log 2>&1
  tail -30 $O/node_day.log
fi
echo r5_run3 $what done
