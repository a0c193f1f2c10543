#!/bin/bash
# experiment: the in-wave matrix row computed by the checking wavefront beside the chain wavefront's epilogue (variants/*_mrow.so against *_base.so)
cd /root/repo
mkdir -p gpurun_out/c23
timeout 1200 python -m pytest tests/test_hip_vs_coracle.py tests/test_switches_gpu.py tests/test_fuzz_gpu.py tests/test_native_sharded_gpu.py tests/test_multi_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/c23/tests.txt
AB_ARGS="--steps 12 --warmup 3" bash tools/ab_bench.sh -
cp gpurun_out/ab.txt gpurun_out/c23/ab.txt
cat gpurun_out/c23/tests.txt gpurun_out/c23/ab.txt
