#!/bin/bash
# A/B of the two-level NN-scan reduction (LQRRT_NN_WG4=0|1) on one box: exact-mode headline, its synchronous-mode extra,
# BASELINE config 5; then WRITE_SIZE / FETCH_SIZE of the scan per launch under both settings (rocprofv3 PMC passes)
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/ab_wg4.txt
for r in 1 2; do
for v in 0 1; do
  LQRRT_NN_WG4=$v python bench.py --no-cpu --steps 10 --warmup 2 --repeats 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WG4=$v cfg4 exact %d | sync %d | scan avg launch %.2f us' % (d['value'], d['synchronous_mode']['value'], d['roofline']['avg_launch_us']))" >> gpurun_out/ab_wg4.txt
  LQRRT_NN_WG4=$v python bench.py --no-cpu --no-extras --workload cfg5 --units 32 --steps 6 --warmup 1 --repeats 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WG4=$v cfg5 exact %d | scan avg launch %.2f us' % (d['value'], d['roofline']['avg_launch_us']))" >> gpurun_out/ab_wg4.txt
done
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
 for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_wg4
  LQRRT_NN_WG4=$v timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_wg4 -o p -- python /root/repo/bench.py --workload cfg5 --steps 2 --warmup 1 --units 8 --no-cpu --no-extras --repeats 1 > /dev/null 2>&1
  f=$(find /tmp/pmc_wg4 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $v $c >> /root/repo/gpurun_out/ab_wg4.txt <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_nn_scan' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[3]]
per={}
for r in rows: per[r['Dispatch_Id']]=per.get(r['Dispatch_Id'],0.0)+float(r['Counter_Value'])
v=sorted(per.items(), key=lambda kv:int(kv[0])); tail=[x for _,x in v[len(v)//2:]]
print('WG4=%s cfg5 scan %s per launch (steady half, %d launches): %.1f (counter units)' % (sys.argv[2], sys.argv[3], len(tail), sum(tail)/max(1,len(tail))))
PY
 done
done
cat /root/repo/gpurun_out/ab_wg4.txt
