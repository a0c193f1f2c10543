#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r04_tests4.txt
AB_SEEDS="1 2 3 4" bash tools/ab_detail.sh - > /dev/null
cp gpurun_out/ab_detail.txt gpurun_out/r04_ab_final_torque.txt
timeout 600 python tools/steer_phases_bench.py > gpurun_out/r04_steer_phases.txt 2>&1
tail -5 gpurun_out/r04_tests4.txt; cat gpurun_out/r04_ab_final_torque.txt; cat gpurun_out/r04_steer_phases.txt
