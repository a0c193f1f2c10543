#!/bin/bash
# Riccati systems: a steer launch is ~240 us whatever it holds, so the wave-size controller (tuned on 18 us launches) may want other limits
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call11.txt
: > $O
run() { echo "-- $*" >> $O; env "$@" RC_ONLY="Riccati boat" RC_DETAIL=1 timeout 300 python tools/run_configs.py 2>/dev/null | grep -v lqr_dare_batch >> $O; }
run A=0
run LQRRT_CTL_MIN=64
run LQRRT_CTL_MIN=32
run LQRRT_CTL_MIN=32 LQRRT_CTL_HI=4
run LQRRT_CTL_MIN=32 LQRRT_CTL_HI=2 LQRRT_CTL_LO=0
run LQRRT_CTL_MIN=16 LQRRT_CTL_HI=3 LQRRT_CTL_LO=1
run LQRRT_CTL_MIN=256
run LQRRT_CTL_MIN=256 LQRRT_CTL_HI=20 LQRRT_CTL_LO=6
run LQRRT_CTL_HI=20 LQRRT_CTL_LO=6
cat $O
