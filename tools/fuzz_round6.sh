#!/bin/bash
# Fuzz record of the final round-6 build (the round-5 matrix with new seeds; the switches removed in round 6 are gone from it) on the GPU box -> gpurun_out/fuzz_r06_final.log: the default build over several seeds, poisoned allocations,
# every rollout / scan / gather form behind a switch, the legacy round sequence, and the out-of-tree example against its oracle.
cd /root/repo
out=gpurun_out/fuzz_r06_final.log
: > $out
run() { timeout 1500 python tools/fuzz_parity.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $out; }
echo "# default switches, seeds 901-904" >> $out
for seed in 901 902 903 904; do run 400 $seed; done
echo "# LQRRT_POISON=1, seeds 911 912" >> $out
for seed in 911 912; do LQRRT_POISON=1 run 400 $seed; done
for sw in LQRRT_STEER_WAVEFRONTS=2 LQRRT_STEER_WAVEFRONTS=3 LQRRT_DARE_WAVEFRONTS=1 LQRRT_DARE_WAVEFRONTS=4 LQRRT_NN_WG4=1 \
           LQRRT_FUSED_ROUNDS=0 LQRRT_MATRIX_MAX_W=0; do
  echo "# $sw, seed 921, 300 cases" >> $out
  env $sw timeout 1500 python tools/fuzz_parity.py 300 921 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
done
echo "# out-of-tree unicycle (FUZZ_USER), seeds 931 932, 200 cases" >> $out
for seed in 931 932; do
  FUZZ_USER=examples/user_system/liblqrrt_unicycle_oracle.so LQRRT_LIB=examples/user_system/liblqrrt_unicycle.so run 200 $seed
done
cat $out
