#!/bin/bash
# slot-bound multi calls (two wavefronts per rollout): does a smaller floor of the wave-size controller (less discarded speculation) pay?
cd /root/repo
mkdir -p gpurun_out/c31
O=gpurun_out/c31/ctl.txt
: > $O
fmt() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d  %.3e attempts/s  mean wave %.1f  rounds/1024 %.1f' % (d['trees'], d['attempts_per_s'], d['mean_wave'], d['rounds_per_1024']))"; }
for cfg in "A=0" "LQRRT_CTL_MIN=96" "LQRRT_CTL_MIN=64" "LQRRT_CTL_MIN=32" "LQRRT_CTL_MIN=64 LQRRT_CTL_CUT=1.25" "A=0"; do
  echo "-- $cfg" >> $O
  env $cfg timeout 400 python tools/multi_bench.py --trees 32,64 --steps 3 --per-call 16384 2>/dev/null | fmt >> $O
done
cat $O
