#!/bin/bash
# A/B on one box, one library: the host's chores in every wait (LQRRT_BG_ALWAYS=1, the loop until now) against only in waits that
# probably roll out (the default since); identical trees
cd /root/repo
mkdir -p gpurun_out/c32
O=gpurun_out/c32/ab_bg.txt
: > $O
for r in 1 2 3 4; do
  for v in 1 0; do
    x=$(LQRRT_BG_ALWAYS=$v python bench.py --no-cpu --no-extras --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['waves_per_1024'], d['repair_rounds_per_1024'])")
    echo "LQRRT_BG_ALWAYS=$v $x" >> $O
  done
done
timeout 600 python -m pytest tests/test_hip_vs_coracle.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1 >> $O
cat $O
