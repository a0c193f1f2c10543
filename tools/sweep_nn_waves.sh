cd /root/repo
for w in 256 512 1024 2048 4096; do LQRRT_NN_WAVES=$w timeout 100 python bench.py --no-cpu --steps 60 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('NN_WAVES=$w', round(d['value']), 'nn us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],2), 'alg MB', round(r['algorithmic_bytes_per_launch']/1e6,1))"; done
