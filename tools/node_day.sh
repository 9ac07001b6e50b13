#!/bin/bash
# Node-day checklist: the exact order for the FIRST lease of a multi-GPU MI355X node.  RCCL with more than
# one rank, hipMemcpyPeerAsync between two devices and cross-device event waits have never executed on this code (rounds 1-5 had
# one-GPU boxes); every line below proves its own topology (`ranks`, `distinct_devices`, `state_identical_across_ranks`).
#
#   tools/node_day.sh              on a node: nccl (= RCCL) backend, N = 2 4 8 (capped at the visible devices)
#   tools/node_day.sh rehearsal    on a one-GPU box: gloo backend, ranks share the device (functional rehearsal; N <= 4: the box
#                                  admits 6 GPU processes); lines carry "value_is_rehearsal": true
#
# Writes one profiles/node_<tag>.json per bench line and profiles/node_table.txt (value, strong.value, distinct_devices,
# state_identical_across_ranks per line).  Order: (1) pytest -m gpu -- the 5 tests that skip on one GPU un-skip --,
# (2) bench.py --gpus {2,4,8} for configs 2-5 over RCCL, (3) the torch-free in-process leg at --gpus 8 (ksa_allreduce_state /
# ksa_scan_allstitch: peer copies), (4) the merge path's fixed cost at N = 1 (--force-collective), (5) the N = 1 lines of the same
# box for the scaling quotient.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mode=${1:-node}
P=profiles
L=gpurun_out/node_day
mkdir -p $P $L
ndev=$(python3 -c "import torch; print(torch.cuda.device_count())")
if [ $mode = rehearsal ]; then
  export KSA_BENCH_BACKEND=gloo
  gpus="2 4"; inproc=8; small=1; tag=rehearsal_
else
  gpus=""; for n in 2 4 8; do [ $n -le $ndev ] && gpus="$gpus $n"; done
  inproc=$ndev; small=0; tag=""
fi
echo "node_day: mode $mode, $ndev visible device(s), rank counts:$gpus, in-process engines $inproc" | tee $L/summary.txt
# bounded batches for the rehearsal (ranks time-slice one GPU through host-staged gloo); the BASELINE batches on a node
args_for() {
  if [ $small = 1 ]; then
    case $1 in 2) echo "--frames 2048 --steps 3 --warmup 1";; 3) echo "--passes 16 --steps 3 --warmup 1";;
               4) echo "--passes 32 --steps 3 --warmup 1";; 5) echo "--frames 64 --steps 3 --warmup 1";; esac
  else
    echo "--steps 20 --warmup 3"
  fi
}
run_line() {   # run_line <name> <bench args...>
  local name=$1; shift
  timeout -k 10 600 python3 bench.py "$@" --no-cpu --no-secondary > $L/$name.json 2> $L/$name.err
  local rc=$?
  if [ $rc != 0 ] || ! [ -s $L/$name.json ]; then echo "FAILED $name (rc $rc): $(tail -2 $L/$name.err | tr '\n' ' ')" | tee -a $L/summary.txt; return; fi
  cp $L/$name.json $P/node_$tag$name.json
  python3 - $L/$name.json $name <<'PY' | tee -a $L/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
rk = d.get("ranks") or {}
print("%-24s n=%d value %.4g %s  strong %s  distinct_devices %s  identical %s  rehearsal %s  backend %s" % (
    sys.argv[2], d["n_gpus"], d["value"], d["unit"], ("%.4g" % d["strong"]["value"]) if d.get("strong") else "-",
    rk.get("distinct_devices", "-"), d.get("state_identical_across_ranks", "-"), d.get("value_is_rehearsal", "-"), rk.get("backend", "-")))
PY
}
# (1) the GPU suite: on a node the nccl world = device_count tests run instead of skipping
if [ $mode = rehearsal ]; then
  timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -k "nccl or allreduce_state or band_sharded or inprocess or rccl" > $L/pytest.log 2>&1
else
  timeout -k 10 1500 python3 -m pytest tests -m gpu -q > $L/pytest.log 2>&1
fi
echo "pytest rc $? : $(tail -1 $L/pytest.log)" | tee -a $L/summary.txt
# (2) one process per GPU over RCCL: every BASELINE configuration at N = 2, 4, 8
for k in 2 3 4 5; do
  for n in $gpus; do run_line c${k}_g$n --config $k --gpus $n $(args_for $k); done
done
# (3) the torch-free form: one process, one engine per device, peer copies + events inside libksa
for k in 2 3 4 5; do run_line inproc_c${k}_g$inproc --inprocess --config $k --gpus $inproc $(args_for $k); done
# (4) what the merge path costs when there is nothing to merge (N = 1, one-rank group), and (5) the N = 1 lines of this box
run_line c2_g1_forced --config 2 --gpus 1 --force-collective $(args_for 2)
for k in 2 3 4 5; do run_line c${k}_g1 --config $k --gpus 1 $(args_for $k); done
cp $L/summary.txt $P/node_${tag}table.txt
echo node_day $mode done
