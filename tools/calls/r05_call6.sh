#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call6.txt
: > $O
for cfg in "LQRRT_MULTI_NWF=3 LQRRT_MULTI_ORDER=0" "LQRRT_MULTI_NWF=3 LQRRT_MULTI_ORDER=1" "LQRRT_MULTI_NWF=2 LQRRT_MULTI_ORDER=0" "LQRRT_MULTI_NWF=2 LQRRT_MULTI_ORDER=1"; do
  echo "== $cfg" >> $O
  env $cfg timeout 600 python tools/multi_bench.py --trees 8,16,32 --steps 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d  %.3e attempts/s  (per tree %.2e)' % (d['trees'], d['attempts_per_s'], d['per_tree']))" >> $O
done
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x 2>&1 | tail -2 >> $O
LQRRT_MULTI_NWF=2 timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x 2>&1 | tail -2 >> $O
cat $O
