#!/bin/bash
# sweep of the wave-size controller's knobs (LQRRT_CTL_*) on the headline workload; same trees in every run
cd /root/repo
run() { env "$@" python bench.py --steps 10 --warmup 2 --repeats 1 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g rounds/1024 %.1f resteers/1024 %.1f waves/1024 %.1f mean wave %.1f' % (d['value'], d['repair_rounds_per_1024'], d['resteers_per_1024'], d['waves_per_1024'], d['mean_wave']))"; }
for cfg in "X=1" "LQRRT_CTL_MIN=96" "LQRRT_CTL_MIN=192" "LQRRT_CTL_MIN=256" "LQRRT_CTL_LO=3" "LQRRT_CTL_LO=4" "LQRRT_CTL_CUT=1.5" "LQRRT_CTL_CUT=2.0" "LQRRT_CTL_HI=6" "LQRRT_CTL_MIN=192 LQRRT_CTL_LO=3" "X=1"; do
  echo "== $cfg"; run $cfg
done
