#!/bin/bash
# experiment: the two-wavefront steer kernel of large multi calls held to three wavefronts per SIMD (LQRRT_MULTI_OCC3=1)
cd /root/repo
mkdir -p gpurun_out/c30
O=gpurun_out/c30/occ3.txt
: > $O
fmt() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d  %.3e attempts/s  (per tree %.2e)' % (d['trees'], d['attempts_per_s'], d['per_tree']))"; }
for r in 1 2; do
for v in 0 1; do
  echo "-- LQRRT_MULTI_OCC3=$v" >> $O
  LQRRT_MULTI_OCC3=$v timeout 400 python tools/multi_bench.py --trees 24,32,64 --steps 3 --per-call 16384 2>/dev/null | fmt >> $O
done
done
LQRRT_MULTI_OCC3=1 timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x -p no:cacheprovider -k "many_engines or every_tree" 2>&1 | tail -2 >> $O
cat $O
