#!/bin/bash
# A/B of the second-choice re-steers of the fused rounds (LQRRT_SECOND_CHOICE=0|1) on identical trees, + the round trace
cd /root/repo
timeout 300 python tools/round_trace.py 8192 2>&1 | tail -4
for i in 1 2 3; do
for v in 0 1; do
  echo "== LQRRT_SECOND_CHOICE=$v"
  LQRRT_SECOND_CHOICE=$v python bench.py --steps 10 --warmup 2 --repeats 1 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g rounds/1024 %.1f resteers/1024 %.1f waves/1024 %.1f' % (d['value'], d['repair_rounds_per_1024'], d['resteers_per_1024'], d['waves_per_1024']))"
done; done
