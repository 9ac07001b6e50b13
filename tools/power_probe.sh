#!/bin/bash
# GPU box: socket power, shader / memory clock and the power cap as rocm-smi reports them while bench.py --config $CFG $BENCH_ARGS runs
# (sampled about once a second; the first samples are taken before the run starts = idle)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
sample() { rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "GPU\[0\]" | grep -E "Power|sclk|mclk|fclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo; }
echo "idle:"; sample; sample
timeout -k 10 300 python3 bench.py --config ${CFG:-2} --steps ${STEPS:-2500} --warmup 3 --no-cpu --no-secondary $BENCH_ARGS > /tmp/pp.json 2> /tmp/pp.err &
pid=$!
sleep ${LEAD:-25}    # import torch + engine set-up + input generation
echo "under bench.py --config ${CFG:-2} $BENCH_ARGS:"
for i in $(seq 1 ${SAMPLES:-10}); do kill -0 $pid 2>/dev/null || break; sample; done
wait $pid
python3 -c "
import json
d=json.load(open('/tmp/pp.json')); r=d['roofline']
print('bench: %.3f MFFT/s ms/step %.4f kern %.4f ms frac %.4f live clock %s GHz' % (d['value']/1e6, d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('shader_clock_ghz_live')))"
