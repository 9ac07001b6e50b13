#!/bin/bash
# GPU box: cycle stamps of the spectrum kernel of bench config $CFG (diagnostic build variants/libksa_stamps.so, -DKSA_STAMPS): share of a
# wave's time per segment; the per-frame output stage split in three (12 segments per wave)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -f /tmp/stamps.txt
KSA_STAMPS_FILE=/tmp/stamps.txt tools/with_lib.sh variants/libksa_stamps.so timeout -k 10 200 python3 bench.py --config ${CFG:-2} --steps 3 --warmup 1 --no-cpu --no-secondary $BENCH_ARGS > /dev/null 2> /tmp/stamps.err || tail -5 /tmp/stamps.err
tail -1 /tmp/stamps.txt | python3 -c "
import sys
names=['0 loads (issue, wait) + taps + unpack / multiply','1 pass 0','2 barrier in front of exchange 1','3 exchange 1 stores + barrier','4 exchange-1 reads + pass 1','5 exchange 2 (barrier, stores, barrier)','6 exchange reads + last pass','7 |X| + fold','8 output stage 3: slot combine, dB, stores (finish_frame)','9 output stage 1: barrier behind the last window','10 output stage 2: fold -> LDS + barrier','11 window start known, loads issued']
for ln in sys.stdin:
    p=ln.split(':'); v=[float(x) for x in p[1].split()]; tot=sum(v)
    print('config ${CFG:-2}', p[0]); [print('  %-58s %6.1f %%  %10.0f clk' % (n, 100*x/tot, x)) for n,x in zip(names,v) if x]
"
