#!/bin/bash
# GPU box, round 5, second call: clock diagnostic, kernel trace of overlap mode (why it loses), general-path prefetch at
# N = 64 (variants/libksa_pf0_64.so), the whole GPU suite.   tools/r5_run2.sh [clock|trace|pf0|tests|all]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
what=${1:-all}
O=gpurun_out/r5b
mkdir -p $O
line() { python3 -c "
import json,sys
d=json.load(open('$1')); r=d['roofline']
print('%-34s %.3f MFFT/s  ms/step %.4f  kern %.4f ms  frac %.4f  flop %.3f  clk %s' % ('$2', d['value']/1e6, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['flop_frac'], r.get('shader_clock_ghz_live')))"; }
if [ $what = clock ] || [ $what = all ]; then
  timeout -k 10 300 python3 tools/clock_diag.py 2>&1 | tee $O/clock_diag.txt
fi
if [ $what = trace ] || [ $what = all ]; then
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r5_overlap -- python3 $R/bench.py --config 2 --overlap 1 --steps 4 --warmup 1 --no-cpu --no-secondary > $R/$O/trace_overlap.log 2>&1 )
  f=$(ls -t gpurun_out/prof_r5_overlap/*/*_kernel_trace.csv | head -1)
  python3 - $f <<'PY' | tee $O/trace_overlap.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
keep = [r for r in rows if any(s in r["Kernel_Name"] for s in ("spectrum_kernel", "accumulate", "commit"))]
for r in keep[-24:]:
    print("%-44s start %10.3f ms  end %10.3f ms  dur %8.3f ms  stream/queue %s" % (r["Kernel_Name"].split("(")[0][-44:], (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?")))
PY
fi
if [ $what = pf0 ] || [ $what = all ]; then
  : > $O/ab_pf0.txt
  for rep in 1 2; do
    for lib in main variants/libksa_pf0_64.so; do
      for fmt in c64 u8; do
        tools/with_lib.sh $lib timeout -k 10 300 python3 bench.py --config 4 --fmt $fmt --steps 10 --warmup 2 --no-cpu --no-secondary > $O/c4_$(basename $lib .so)_${fmt}_$rep.json 2> $O/c4_$(basename $lib .so)_${fmt}_$rep.err || echo "c4 $lib failed"
        line $O/c4_$(basename $lib .so)_${fmt}_$rep.json "config 4 $fmt $(basename $lib .so) rep $rep" | tee -a $O/ab_pf0.txt
      done
    done
  done
fi
if [ $what = tests ] || [ $what = all ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -12 $O/pytest.log
fi
echo r5_run2 $what done
