#!/bin/bash
# Round 5, GPU call 1: (a) the scan regression r3 -> r4 -> fix, three builds on one box, interleaved; (b) CU-mask probe and sweep;
# (c) torque default A/B over 8 sample seeds; (d) the GPU test suite on the fixed build.
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call1.txt
: > $O
echo "== box" >> $O
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i partition >> $O
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, getattr(p,'gcnArchName',''))" >> $O 2>&1
echo "== cumask probe" >> $O
timeout 120 tools/micro/cumask.bin >> $O 2>&1
echo "== nn_bench (launch_us by W), three builds interleaved" >> $O
for rep in 1 2; do
  for t in variants/r3tree variants/r4tree .; do
    (cd $t && timeout 300 python tools/nn_bench.py --reps 200 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$t rep$rep ' + ' '.join('W%d=%.2f' % (x['W'], x['launch_us']) for x in r))") >> $O
  done
done
echo "== bench: exact value / live scan us / synchronous extra (units 16, steps 5), then cfg5" >> $O
for rep in 1 2; do
  for t in variants/r3tree variants/r4tree .; do
    (cd $t && timeout 300 python bench.py --no-cpu --units 16 --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t rep$rep cfg4 value=%d scan_us=%.2f sync=%d' % (d['value'], d['roofline']['avg_launch_us'], d['synchronous_mode']['value']))") >> $O
    (cd $t && timeout 300 python bench.py --workload cfg5 --no-cpu --no-extras --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t rep$rep cfg5 value=%d scan_us=%.2f' % (d['value'], d['roofline']['avg_launch_us']))") >> $O
  done
done
echo "== CU mask sweep (same tree: no arithmetic changes), seeds 1 2" >> $O
AB_SEEDS="1 2" bash tools/ab_detail.sh - LQRRT_CU_XCDS=1 LQRRT_CU_XCDS=2 LQRRT_CU_XCDS=4 - LQRRT_CU_XCDS=1 LQRRT_CU_XCDS=2 LQRRT_CU_XCDS=4 > /dev/null 2>&1
cat gpurun_out/ab_detail.txt >> $O
echo "== torque default A/B, 8 seeds" >> $O
AB_SEEDS="1 2 3 4 5 6 7 8" bash tools/ab_detail.sh LQRRT_TORQUE_VMIN=0.01 LQRRT_TORQUE_VMIN=inf > /dev/null 2>&1
cat gpurun_out/ab_detail.txt >> $O
echo "== pytest -m gpu" >> $O
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_call1_pytest.log 2>&1
tail -5 gpurun_out/r05_call1_pytest.log >> $O
cat $O
