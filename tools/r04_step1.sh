#!/bin/bash
# round 4, step 1 on the GPU box: GPU suite with the one-atan2 heading torque, A/B against round 3's library and across
# the speed threshold, per-step phases of the rebalanced chain-owner rollout
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r04_tests1.txt
AB_ARGS="--steps 10 --warmup 2 --repeats 1" bash tools/ab_bench.sh - LQRRT_TORQUE_VMIN=0 LQRRT_TORQUE_VMIN=inf
cp gpurun_out/ab.txt gpurun_out/r04_ab_torque.txt
timeout 600 python tools/steer_phases_bench.py > gpurun_out/r04_steer_phases.txt 2>&1
tail -5 gpurun_out/r04_tests1.txt; cat gpurun_out/r04_ab_torque.txt; cat gpurun_out/r04_steer_phases.txt
