#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r05_call5.txt
: > $O
echo "== hostprof multi" >> $O
LQRRT_HOSTPROF=1 timeout 600 python tools/multi_bench.py --trees 4,16,32 --steps 3 2>&1 | grep -E "hostprof multi|trees" | tail -40 >> $O
echo "== rocprofv3 kernel stats, 16 trees" >> $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_multi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_multi -o m -- python /root/repo/tools/multi_bench.py --trees 16 --steps 3 > /tmp/prof_multi.log 2>&1
f=$(find /tmp/prof_multi -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-220 >> /root/repo/$O
cd /root/repo
tail -60 $O
