#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call4.txt
: > $O
timeout 900 python -m pytest tests/test_multi_gpu.py -m gpu -q -x > gpurun_out/r05_call4_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r05_call4_pytest.log | tail -3 >> $O
grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/r05_call4_pytest.log | head -20 >> $O
echo "== multi_bench" >> $O
timeout 900 python tools/multi_bench.py --trees 1,2,4,8,16,32 --steps 4 >> $O 2>gpurun_out/r05_call4_multi.err
tail -5 gpurun_out/r05_call4_multi.err >> $O
echo "== bench sanity (refactored kernels)" >> $O
timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --repeats 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg4 value=%d scan_us=%.2f steer_us=%.2f sync=%d' % (d['value'], d['roofline']['avg_launch_us'], d['steer_kernel']['avg_launch_us'], d['synchronous_mode']['value']))" >> $O
tail -40 $O
