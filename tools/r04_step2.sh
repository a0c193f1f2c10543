#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/r04_tests2.txt
AB_SEEDS="1 2 3" bash tools/ab_detail.sh - LQRRT_STEER_WAVEFRONTS=4 LQRRT_TORQUE_VMIN=inf > /dev/null
cp gpurun_out/ab_detail.txt gpurun_out/r04_ab_chain.txt
timeout 600 python tools/steer_phases_bench.py > gpurun_out/r04_steer_phases.txt 2>&1
tail -5 gpurun_out/r04_tests2.txt; cat gpurun_out/r04_ab_chain.txt; cat gpurun_out/r04_steer_phases.txt
