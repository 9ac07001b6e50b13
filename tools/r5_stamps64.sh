#!/bin/bash
# GPU box: cycle stamps of spectrum64_kernel (the 8 x 8 plan of N = 64) at bench config 4 (diagnostic build
# variants/libksa_stamps.so, -DKSA_STAMPS): share of a wave's time per segment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -f /tmp/stamps.txt
KSA_STAMPS_FILE=/tmp/stamps.txt tools/with_lib.sh variants/libksa_stamps.so timeout -k 10 200 python3 bench.py --config 4 --steps 3 --warmup 1 --no-cpu --no-secondary > /dev/null 2> /tmp/stamps.err || tail -5 /tmp/stamps.err
tail -1 /tmp/stamps.txt | python3 -c "
import sys
names={11:'window start known, 8 x 16-byte loads issued',0:'load wait + taps + window multiply',1:'step A (2 x radix-8 over j) + step B (14 twiddles)',3:'exchange: barrier, 16 stores, barrier',6:'16 exchange reads + step C (2 x radix-8 over m)',7:'|X| + fold',10:'output stage 1: fold -> LDS staging + barriers',8:'output stage 2: slot combine, dB, stores (finish_frame)'}
for ln in sys.stdin:
    p=ln.split(':'); v=[float(x) for x in p[1].split()]; tot=sum(v)
    print('config 4 spectrum64_kernel', p[0]); [print('  %2d %-58s %6.1f %%  %10.0f clk' % (i, names[i], 100*v[i]/tot, v[i])) for i in (11,0,1,3,6,7,10,8)]
"
