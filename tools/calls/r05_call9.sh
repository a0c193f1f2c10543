#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call9.txt
: > $O
for t in "" 1 2 4 8; do
  echo "-- LQRRT_MULTI_THREADS=$t" >> $O
  LQRRT_MULTI_THREADS=$t timeout 600 python tools/multi_bench.py --trees 4,8,16,32,64 --steps 3 --per-call 16384 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d  %.3e attempts/s  (per tree %.2e)' % (d['trees'], d['attempts_per_s'], d['per_tree']))" >> $O
done
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q 2>&1 | tail -2 >> $O
cat $O
