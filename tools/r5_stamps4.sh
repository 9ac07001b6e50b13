#!/bin/bash
# GPU box: cycle stamps of spectrum_kernel<64> at config 4 (diagnostic build, -DKSA_STAMPS): share of a wave's time per segment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -f /tmp/stamps.txt
KSA_STAMPS_FILE=/tmp/stamps.txt tools/with_lib.sh variants/libksa_stamps.so timeout -k 10 200 python3 bench.py --config ${CFG:-4} --steps 3 --warmup 1 --no-cpu --no-secondary $BENCH_ARGS > /dev/null 2> /tmp/stamps.err || tail -5 /tmp/stamps.err
tail -2 /tmp/stamps.txt | python3 -c "
import sys
names=['0 loads (issue, wait) + taps + unpack','1 pass 0 (window multiply fused)','2 wait in front of the exchange stores','3 exchange stores + wait','4 -','5 -','6 exchange reads + last pass (radix 16)','7 |X| + fold','8 per-frame output stage (slot combine, dB, stores)','9 -']
for ln in sys.stdin:
    p=ln.split(':'); v=[float(x) for x in p[1].split()]; tot=sum(v)
    print(p[0]); [print('  %-52s %6.1f %%  %10.0f clk' % (n, 100*x/tot, x)) for n,x in zip(names,v) if x]
"
