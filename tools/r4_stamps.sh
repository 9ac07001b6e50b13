#!/bin/bash
# GPU box: in-kernel cycle stamps of the config-2 kernel (diagnostic build, -DKSA_STAMPS): share of a wave's time per segment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -f /tmp/stamps.txt
KSA_STAMPS_FILE=/tmp/stamps.txt tools/with_lib.sh variants/libksa_stamps.so timeout -k 10 200 python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu --no-secondary > /dev/null 2> /tmp/stamps.err || tail -5 /tmp/stamps.err
tail -2 /tmp/stamps.txt | python3 -c "
import sys
names=['0 loads+taps+multiply','1 pass 0','2 barrier A','3 write 1 + barrier','4 read + pass 1','5 exchange 2 (barrier, write, barrier)','6 read + pass 2','7 fold','8 output stage','9 -']
for ln in sys.stdin:
    p=ln.split(':'); v=[float(x) for x in p[1].split()]; tot=sum(v)
    print(p[0]); [print('  %-42s %6.1f %%  %8.0f clk per frame' % (n, 100*x/tot, x/ (1 if tot==0 else 1))) for n,x in zip(names,v)]
"
