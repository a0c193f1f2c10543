#!/bin/bash
# update_plans tests; where the multi loop's host time goes with 2 / 4 / 8 groups of engines (16 trees)
cd /root/repo
mkdir -p gpurun_out/c17
timeout 900 python -m pytest tests/test_multi_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/c17/tests.txt
for t in 1 2 4 8; do
  echo "-- LQRRT_MULTI_THREADS=$t" >> gpurun_out/c17/hostprof.txt
  LQRRT_HOSTPROF=1 LQRRT_MULTI_THREADS=$t timeout 300 python tools/multi_bench.py --trees 16 --steps 2 --per-call 16384 2>&1 | grep -E "hostprof multi|^\{" | tail -12 | cut -c1-400 >> gpurun_out/c17/hostprof.txt
done
cat gpurun_out/c17/tests.txt gpurun_out/c17/hostprof.txt
