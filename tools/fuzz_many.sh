#!/bin/bash
# Long fuzz session on the GPU box: tools/fuzz_parity.py over several seeds -> gpurun_out/fuzz_many.log
cd /root/repo
: > gpurun_out/fuzz_many.log
for seed in "$@"; do
  timeout 1500 python tools/fuzz_parity.py 400 $seed 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/fuzz_many.log
done
