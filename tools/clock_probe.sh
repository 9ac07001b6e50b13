#!/bin/bash
# GPU box: shader clock held during the dominant kernel = GRBM_GUI_ACTIVE / 8 / kernel time, for two builds/switches
# usage: clock_probe.sh <tag> [env assignments...] -- bench args
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
out=$R/gpurun_out/clock_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu "$@" > $out/log.txt 2>&1
python3 - <<EOT
import csv, glob, collections
f = glob.glob("$out/*/*_counter_collection.csv")[0]
t = glob.glob("$out/*/*_kernel_trace.csv")[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "spectrum" not in k and "dif16_kernel" not in k: continue
    ns = sum(dur[k]) / len(dur[k])
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print("$tag %-52s %.3f ms  clock %.3f GHz  VALU instr %.3e  wave quad-cycles %.3e  VALU active %.3e wait_any %.3e wait_inst %.3e" % (k, ns / 1e6, g / 8 / ns, m["SQ_INSTS_VALU"], m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_VALU"], m["SQ_WAIT_ANY"], m["SQ_WAIT_INST_ANY"]))
EOT
