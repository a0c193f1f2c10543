#!/bin/bash
# fuzz record of the round-5 Riccati build: default fuzz, the Riccati-only fuzz with one and four wavefronts per rollout
cd /root/repo
out=gpurun_out/fuzz_r05.log
: > $out
echo "# default switches, seeds 501 502, 300 cases" >> $out
for seed in 501 502; do timeout 1200 python tools/fuzz_parity.py 300 $seed 2>&1 | grep -v amdgpu.ids | tail -1 >> $out; done
for sw in A=0 LQRRT_DARE_WAVEFRONTS=1 LQRRT_DARE_WAVEFRONTS=4 LQRRT_POISON=1; do
  echo "# FUZZ_RICCATI=1 $sw, seed 511, 120 cases" >> $out
  env FUZZ_RICCATI=1 $sw timeout 1500 python tools/fuzz_parity.py 120 511 2>&1 | grep -v amdgpu.ids | tail -3 >> $out
done
cat $out
