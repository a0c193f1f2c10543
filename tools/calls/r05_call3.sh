#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call3.txt
: > $O
timeout 1500 python -m pytest tests/test_chain.py tests/test_native_sharded_gpu.py tests/test_pendulum_lqr.py tests/test_config5.py tests/test_hip_parity.py -m gpu -q -s > gpurun_out/r05_call3_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r05_call3_pytest.log | tail -3 >> $O
grep -E "^FAILED|^ERROR|two-rank parity path|^RANK" gpurun_out/r05_call3_pytest.log | head -20 >> $O
for r in 1 2; do
python bench.py --workload cfg5 --no-cpu --no-extras --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg5 value=%d scan_us=%.2f' % (d['value'], d['roofline']['avg_launch_us']))" >> $O
done
tail -30 $O
