#!/bin/bash
# multi-tree loop: wavefronts per rollout 3 / 2 / 1 at 16, 32, 64 trees (two host-thread groups; 64 trees also on four)
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call14.txt
: > $O
fmt() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d threads %d  %.3e attempts/s  (per tree %.2e, rounds/1024 %.1f)' % (d['trees'], d['host_threads'], d['attempts_per_s'], d['per_tree'], d['rounds_per_1024']))"; }
for nwf in 3 2 1; do
  echo "-- LQRRT_MULTI_NWF=$nwf (default thread groups)" >> $O
  LQRRT_MULTI_NWF=$nwf timeout 500 python tools/multi_bench.py --trees 16,32,64 --steps 3 --per-call 16384 2>/dev/null | fmt >> $O
done
echo "-- LQRRT_MULTI_NWF=1 LQRRT_MULTI_THREADS=4" >> $O
LQRRT_MULTI_NWF=1 LQRRT_MULTI_THREADS=4 timeout 300 python tools/multi_bench.py --trees 32,64 --steps 3 --per-call 16384 2>/dev/null | fmt >> $O
echo "-- LQRRT_MULTI_NWF=1 LQRRT_MULTI_THREADS=1" >> $O
LQRRT_MULTI_NWF=1 LQRRT_MULTI_THREADS=1 timeout 300 python tools/multi_bench.py --trees 32,64 --steps 3 --per-call 16384 2>/dev/null | fmt >> $O
LQRRT_MULTI_NWF=1 timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
cat $O
