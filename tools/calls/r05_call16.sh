#!/bin/bash
# launches per second from 1..8 host threads (the multi loop's host-side ceiling), then a fuzz record of the final build
cd /root/repo
mkdir -p gpurun_out/c16
tools/micro/launch_threads.bin > gpurun_out/c16/launch_threads.txt 2>&1
out=gpurun_out/c16/fuzz.txt
: > $out
for seed in 601 602; do
  echo "# default switches, seed $seed, 250 cases" >> $out
  timeout 1200 python tools/fuzz_parity.py 250 $seed 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
done
echo "# FUZZ_RICCATI=1, seed 611, 80 cases" >> $out
FUZZ_RICCATI=1 timeout 1200 python tools/fuzz_parity.py 80 611 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
cat gpurun_out/c16/launch_threads.txt $out
