#!/bin/bash
# The two Riccati lines of tools/run_configs.py with every library under variants/ on one box -> gpurun_out/riccati_ab.txt
cd /root/repo
{
for so in variants/*.so; do
  echo "== $so"
  RC_ONLY=Riccati LQRRT_LIB=$PWD/$so python tools/run_configs.py 2>/dev/null
done
} | tee gpurun_out/riccati_ab.txt
