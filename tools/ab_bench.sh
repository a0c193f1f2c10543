#!/bin/bash
# A/B on ONE box: bench.py with each library under variants/ (LQRRT_LIB) and each environment setting given as
# arguments ("VAR=value[,VAR2=value2]" or "-"), interleaved, 2 repetitions -> gpurun_out/ab.txt
cd /root/repo
: > gpurun_out/ab.txt
[ $# -eq 0 ] && set -- "-"
for r in 1 2; do
  for so in variants/*.so; do
    for kv in "$@"; do
      if [ "$kv" = "-" ]; then envs=""; else envs="${kv//,/ }"; fi
      v=$(env $envs LQRRT_LIB=$PWD/$so python bench.py --no-cpu --no-extras ${AB_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
      echo "$so $kv $v" >> gpurun_out/ab.txt
    done
  done
done
