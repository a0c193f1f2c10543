#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call7.txt
: > $O
for q in 4 8 16 24; do
  echo "== concurrent_planners (one Python thread + stream per planner, single-engine loops), GPU_MAX_HW_QUEUES=$q" >> $O
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/concurrent_planners.py 4 8 16 2>/dev/null | grep planners >> $O
done
for q in 4 16; do
  echo "== multi_bench with host threads (groups), GPU_MAX_HW_QUEUES=$q" >> $O
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/multi_bench.py --trees 16,32 --threads 1,2,4,8 --steps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d threads %d  %.3e attempts/s' % (d['trees'], d['host_threads'], d['attempts_per_s']))" >> $O
done
cat $O
