#!/bin/bash
# Round 3, item 1: the effective shader clock under the latency-bound loop (gpurun -- 'bash tools/clock_round.sh')
R=/root/repo
O=$R/gpurun_out/clock
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
echo "### rocm-smi before"; rocm-smi --showclocks --showperflevel 2>&1 | grep -v "^$" | head -40
echo "### clock probe (perf level auto)"
timeout 300 $R/tools/micro/clock.bin
echo "### rocm-smi --setperflevel high"
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showclocks --showperflevel 2>&1 | grep -i -E "sclk|perf" | head -4
timeout 200 $R/tools/micro/clock.bin quick
echo "### rocm-smi --setperflevel auto"
rocm-smi --setperflevel auto 2>&1 | tail -3
} > $O/clock_probe.txt 2>&1
timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/bench_base.json 2> $O/bench_base.err < /dev/null
# GRBM_GUI_ACTIVE / GRBM_COUNT over the bench loop (own run, kernel-trace only)
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d /tmp/pmc_grbm -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > /dev/null 2> $O/grbm.log < /dev/null
f=$(find /tmp/pmc_grbm -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/grbm.txt <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
disp = {}
for r in rows:
    d = r["Dispatch_Id"]
    e = disp.setdefault(d, dict(k=r["Kernel_Name"][:60], s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), c={}))
    e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
per = collections.defaultdict(list)
for d in disp.values(): per[re.sub(r"\(.*", "", d["k"])].append(d)
for k, ds in sorted(per.items(), key=lambda kv: -len(kv[1]))[:8]:
    ds.sort(key=lambda d: d["s"]); tail = ds[len(ds)//2:]
    ns = sum(d["e"] - d["s"] for d in tail) / len(tail)
    ga = sum(d["c"].get("GRBM_GUI_ACTIVE", 0) for d in tail) / len(tail)
    gc = sum(d["c"].get("GRBM_COUNT", 0) for d in tail) / len(tail)
    print("%-60s launches %6d  avg %8.0f ns  GRBM_GUI_ACTIVE %10.0f  GRBM_COUNT %10.0f  -> GUI_ACTIVE/ns = %.3f GHz (x1 XCD?)  COUNT/ns = %.3f" % (k, len(ds), ns, ga, gc, ga / ns, gc / ns))
PY
rm -rf /tmp/pmc_grbm
ls -la $O; cat $O/clock_probe.txt | head -120
