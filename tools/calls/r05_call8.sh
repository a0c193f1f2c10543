#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call8.txt
: > $O
echo "== pytest -m gpu (full)" >> $O
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05_call8_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r05_call8_pytest.log | tail -3 >> $O
grep -E "^FAILED|^ERROR" gpurun_out/r05_call8_pytest.log | head -20 >> $O
echo "== multi_bench, native threaded groups (default) and forced thread counts" >> $O
for t in "" 1 2 4 6; do
  echo "-- LQRRT_MULTI_THREADS=$t" >> $O
  LQRRT_MULTI_THREADS=$t timeout 600 python tools/multi_bench.py --trees 4,8,16,32,64 --steps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  trees %2d  %.3e attempts/s  (per tree %.2e)' % (d['trees'], d['attempts_per_s'], d['per_tree']))" >> $O
done
echo "== bench (driver command, no cpu)" >> $O
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/r05_call8_bench.json 2>/dev/null
python -c "
import json
d = json.loads(open('gpurun_out/r05_call8_bench.json').read().strip().splitlines()[-1])
print('cfg4 value=%d scan_us=%.2f steer_us=%.2f sync=%d multi16=%d' % (d['value'], d['roofline']['avg_launch_us'], d['steer_kernel']['avg_launch_us'], d['synchronous_mode']['value'], d['multi_tree']['value']))
print(d['config']['box'])" >> $O 2>&1
cat $O
