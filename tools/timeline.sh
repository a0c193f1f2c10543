#!/bin/bash
# Kernel timeline of the steady-state bench loop: rocprofv3 kernel trace -> gpurun_out/kt_tail.csv (last 4000 dispatches)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python /root/repo/bench.py --steps 3 --warmup 1 --units 16 --no-cpu --no-extras > /dev/null 2>&1 < /dev/null
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then head -1 "$f" > /root/repo/gpurun_out/kt_tail.csv; tail -4000 "$f" >> /root/repo/gpurun_out/kt_tail.csv; wc -l "$f"; fi
