#!/bin/bash
# re-entry check of the round-5 build on a fresh box: GPU tests, smoke, the bench line, the two launch-boundary micro-benchmarks
cd /root/repo
mkdir -p gpurun_out/c13
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/c13/gpu_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/c13/smoke.txt
tools/micro/launch_floor.bin > gpurun_out/c13/launch_floor.txt 2>&1
tools/micro/launch_chain.bin > gpurun_out/c13/launch_chain.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c13/bench.json 2> gpurun_out/c13/bench.err
cat gpurun_out/c13/gpu_tests.txt gpurun_out/c13/smoke.txt gpurun_out/c13/launch_floor.txt gpurun_out/c13/launch_chain.txt
tail -c 1500 gpurun_out/c13/bench.json
