#!/bin/bash
# Round 5, GPU call 2: CU-mask probe; GPU suite on the head2 build; the two-level scan reduction re-measured on the repaired loop
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call2.txt
: > $O
echo "== cumask probe" >> $O
timeout 90 tools/micro/cumask.bin >> $O 2>&1; echo "rc=$?" >> $O
echo "== pytest -m gpu" >> $O
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05_call2_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r05_call2_pytest.log | tail -3 >> $O
grep -E "^FAILED|^ERROR" gpurun_out/r05_call2_pytest.log | head -20 >> $O
echo "== WG4 A/B on the repaired scan" >> $O
timeout 900 bash tools/ab_wg4.sh > /dev/null 2>&1
cat gpurun_out/ab_wg4.txt >> $O
tail -40 $O
