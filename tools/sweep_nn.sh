#!/bin/bash
# A/B of the scan's launch decomposition on the GPU box: wavefronts per launch x minimum nodes per wavefront.
#   bash tools/sweep_nn.sh   -> gpurun_out/sweep_nn.log  (scan alone: tools/nn_bench.py; end to end: bench.py)
cd /root/repo
O=gpurun_out/sweep_nn.log
: > $O
for w in 1024 2048 4096 8192; do for c in 16 32 64; do
  echo "== NN_WAVES=$w MIN_CHUNK=$c" >> $O
  LQRRT_NN_WAVES=$w LQRRT_NN_MIN_CHUNK=$c timeout 120 python tools/nn_bench.py --reps 100 --waves 64,256,1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   W=%4d %6.2f us  %.2f Tpairs/s' % (d['W'], d['launch_us'], d['pairs_per_s'] / 1e12))" >> $O
  LQRRT_NN_WAVES=$w LQRRT_NN_MIN_CHUNK=$c timeout 120 python bench.py --no-cpu --no-extras --steps 60 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']; print('   bench %d attempts/s, nn %.2f us' % (d['value'], r['avg_launch_us']))" >> $O
done; done
