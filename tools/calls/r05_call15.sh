#!/bin/bash
# the multi loop's automatic choice of the rollout form (two wavefronts from 24 engines on): trees unchanged, the sweep, the bench line
cd /root/repo
mkdir -p gpurun_out/c15
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_bench_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/c15/tests.txt
timeout 500 python tools/multi_bench.py --trees 8,16,24,32,48,64 --steps 3 --per-call 16384 2>/dev/null | grep "^{" > gpurun_out/c15/multi.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c15/bench.json 2> gpurun_out/c15/bench.err
cat gpurun_out/c15/tests.txt
python - <<'P'
import json
for l in open('gpurun_out/c15/multi.jsonl'):
    d = json.loads(l); print('trees %2d  %.3e attempts/s (per tree %.2e)' % (d['trees'], d['attempts_per_s'], d['per_tree']))
d = json.loads(open('gpurun_out/c15/bench.json').read().strip().splitlines()[-1])
print(d['value'], {k: (v['value'], v['trees']) for k, v in d.items() if k.startswith('multi_tree')})
P
