#!/bin/bash
# Dynamic instruction mix of the steer kernel: rocprofv3 PMC pass over a short bench run.
# usage (on the GPU box): bash tools/pmc_steer.sh  -> gpurun_out/pmc_steer.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1 < /dev/null
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
echo "file: $f" > /root/repo/gpurun_out/pmc_steer.txt
[ -n "$f" ] && python3 - "$f" >> /root/repo/gpurun_out/pmc_steer.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); n[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0))[:8]:
    w = max(c.get('SQ_WAVES', 1), 1)
    print("%-60s launches %6d waves/launch %7.1f | per wave: VALU %8.0f SALU %8.0f LDS %7.0f cycles %9.0f" % (
        k, n[k], w / n[k], c['SQ_INSTS_VALU'] / w, c['SQ_INSTS_SALU'] / w, c['SQ_INSTS_LDS'] / w, c.get('SQ_WAVE_CYCLES', 0) / w))
PY
