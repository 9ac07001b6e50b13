#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
tools/r5_ab_slot.sh variants/libksa_old.so variants/libksa_ch5only.so main 2>&1 | tee gpurun_out/r5_ab_slot2.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_round4.py tests/test_gpu_round2.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r5_narrow_tests.txt
