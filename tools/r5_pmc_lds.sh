#!/bin/bash
# GPU box: the LDS counters of bench config $CFG with library $1 swapped in (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the dominant kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib=$1; tag=$(basename $lib .so)
out=$R/gpurun_out/pmc_lds_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
$R/tools/with_lib.sh $lib timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out -- python3 $R/bench.py --config ${CFG:-4} --steps 3 --warmup 1 --no-cpu --no-secondary > $out/run.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in tot.items():
    if 'spectrum' in k and v.get('SQ_LDS_IDX_ACTIVE'):
        print(k, dict(v), 'conflict ratio %.4f' % (v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE']))
PY
