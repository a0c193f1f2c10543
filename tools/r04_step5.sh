#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "^Hostname\|^Librccl\|^HIP version\|^ROCm version\|amdgpu.ids" | tail -12 > gpurun_out/r04_tests5.txt
AB_SEEDS="1 2 3" AB_ARGS="--steps 10 --warmup 2 --repeats 1 --no-extras" bash tools/ab_detail.sh - LQRRT_SCAN_OVERLAP=0 > /dev/null
cp gpurun_out/ab_detail.txt gpurun_out/r04_ab_overlap.txt
tail -6 gpurun_out/r04_tests5.txt; cut -c1-190 gpurun_out/r04_ab_overlap.txt
