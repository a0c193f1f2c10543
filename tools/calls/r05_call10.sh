#!/bin/bash
# multi-tree mode: does the wave-size controller tuned for a lone planner (launch ~18 us) suit the lock step (tick ~50-80 us)?
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call10.txt
: > $O
run() {
  echo "-- $*" >> $O
  env "$@" timeout 300 python tools/multi_bench.py --trees 16,32 --steps 3 --per-call 16384 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d  %.3e attempts/s  mean wave %.1f  rounds/1024 %.1f' % (d['trees'], d['attempts_per_s'], d['mean_wave'], d['rounds_per_1024']))" >> $O
}
run A=0
run LQRRT_CTL_MIN=192
run LQRRT_CTL_MIN=256
run LQRRT_CTL_MIN=256 LQRRT_CTL_CUT=2
run LQRRT_CTL_MIN=256 LQRRT_CTL_LO=4 LQRRT_CTL_HI=16
run LQRRT_CTL_MIN=64
run LQRRT_CTL_LO=4 LQRRT_CTL_HI=16
run LQRRT_MULTI_NWF=2
run LQRRT_MULTI_NWF=2 LQRRT_CTL_MIN=256
cat $O
