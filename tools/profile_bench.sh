#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 evidence for bench.py's dominant kernel(s).
#   tools/profile_bench.sh <tag> [bench.py args, e.g. --config 3]
#     -> gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,pmc_sq,pmc_sq2}
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with a
# trace domain other than --kernel-trace).  tools/summarize_profile.py <tag> then writes profiles/<tag>_*.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-run}
shift || true
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "$@" > $out/bench_args.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary "$@" > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary "$@" > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary "$@" > $out/pmc_write.log 2>&1
if [ -z "$KSA_PROF_NO_SQ" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary "$@" > $out/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary "$@" > $out/pmc_sq2.log 2>&1
fi
grep -h '^{' $out/stats.log | tail -1 > $out/bench_line.json || true
echo profiled $tag
