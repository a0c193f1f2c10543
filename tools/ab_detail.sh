#!/bin/bash
# A/B on ONE box with the loop's own counters next to the value: every library under variants/ x every environment setting
# given as argument ("VAR=value[,VAR2=value2]" or "-") x the seeds in AB_SEEDS (default "1"): different torque forms grow
# different trees (the problem is chaotic), and tree-to-tree spread is several per cent -> gpurun_out/ab_detail.txt
cd /root/repo
: > gpurun_out/ab_detail.txt
[ $# -eq 0 ] && set -- "-"
for seed in ${AB_SEEDS:-1}; do
  for so in variants/*.so; do
    for kv in "$@"; do
      if [ "$kv" = "-" ]; then envs=""; else envs="${kv//,/ }"; fi
      env $envs LQRRT_LIB=$PWD/$so python bench.py --no-cpu --seed $seed ${AB_ARGS:---steps 10 --warmup 2 --repeats 1} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sk=d['steer_kernel']
print('$so $kv seed=$seed value=%d acc/s=%d yield=%.3f waves/1024=%.2f rounds/1024=%.2f resteers/1024=%.1f hits/1024=%.2f mean_wave=%.1f steer_us=%.2f steer_launches/wave=%.2f growth_s=%.3f' % (
  d['value'], d['accepted_nodes_per_s'], d['accepted_nodes_per_s']/d['value'], d['waves_per_1024'], d['repair_rounds_per_1024'], d['resteers_per_1024'], d['goal_hits_per_1024'], d['mean_wave'], sk['avg_launch_us'], 0.0, d['tree_growth_s']))" >> gpurun_out/ab_detail.txt
    done
  done
done
cat gpurun_out/ab_detail.txt
