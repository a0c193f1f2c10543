#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_hip_vs_coracle.py tests/test_teacher_gpu.py tests/test_switches_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r04_tests3.txt
AB_SEEDS="1 2 3" bash tools/ab_detail.sh - > /dev/null
cp gpurun_out/ab_detail.txt gpurun_out/r04_ab_chain4.txt
timeout 600 python tools/steer_phases_bench.py > gpurun_out/r04_steer_phases.txt 2>&1
tail -5 gpurun_out/r04_tests3.txt; cat gpurun_out/r04_ab_chain4.txt; cat gpurun_out/r04_steer_phases.txt
