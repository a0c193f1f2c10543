#!/bin/bash
# exact-mode value, live scan launch time and synchronous-mode extra of bench.py for every library under variants/ and every
# environment setting given as argument ("VAR=value" or "-"); same box, interleaved
cd /root/repo
[ $# -eq 0 ] && set -- "-"
for r in 1 2; do for so in variants/*.so; do for kv in "$@"; do
 if [ "$kv" = "-" ]; then envs=""; else envs="$kv"; fi
 env $envs LQRRT_LIB=$PWD/$so python bench.py --no-cpu --units 16 --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so $kv', round(d['value']), 'scan us', round(d['roofline']['avg_launch_us'],2), 'sync', round(d['synchronous_mode']['value']))"
done; done; done
