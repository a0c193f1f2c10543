#!/bin/bash
# quick A/B of the wave-size controller and chunking knobs on the bench loop (same box, back to back)
cd /root/repo
run() { env "$@" timeout 100 python bench.py --no-cpu --steps 60 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['waves_per_step'],2), round(d['repair_rounds_per_step'],1))"; }
run LQRRT_CTL_CUT=2 LQRRT_TRI_CHUNK=32
run LQRRT_CTL_CUT=1.5 LQRRT_TRI_CHUNK=32
run LQRRT_CTL_CUT=1.2 LQRRT_TRI_CHUNK=32
run LQRRT_CTL_CUT=2 LQRRT_TRI_CHUNK=32 LQRRT_CTL_MIN=64
run LQRRT_CTL_CUT=1.5 LQRRT_TRI_CHUNK=32 LQRRT_CTL_MIN=64
run LQRRT_CTL_CUT=2 LQRRT_TRI_CHUNK=32 LQRRT_CTL_MIN=96
run LQRRT_CTL_CUT=4
