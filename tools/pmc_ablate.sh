#!/bin/bash
# Dynamic instruction counts of the steer micro-benchmark (tools/ablate_steer.py): instructions per wavefront for the
# 0-step and the 10-step problems, hence per rollout step.  -> gpurun_out/pmc_ablate.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /tmp/pa -o p -- python /root/repo/tools/ablate_steer.py "$1" > /tmp/pa.log 2>&1 < /dev/null
f=$(find /tmp/pa -name "*counter_collection.csv" | head -1)
python3 - "$f" > /root/repo/gpurun_out/pmc_ablate.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
    if "k_steer<" not in r["Kernel_Name"]: continue
    d = disp.setdefault(r["Dispatch_Id"], dict(grid=int(r["Grid_Size"]), t=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), c={}))
    d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ds = list(disp.values())
# four groups of 30 launches: (64 problems, 0 steps), (64, 10 steps), (64, 10 steps), (1024, 10 steps)
for g in range(0, len(ds), 30):
    grp = ds[g:g + 30]
    w = sum(d["c"].get("SQ_WAVES", 0) for d in grp) / len(grp)
    print("group", g // 30, "grid", grp[0]["grid"], "waves %.0f" % w, "avg us %.2f" % (sum(d["t"] for d in grp) / len(grp) / 1e3),
          " per wavefront:", " ".join("%s %.0f" % (k[8:], sum(d["c"].get(k, 0) for d in grp) / len(grp) / max(w, 1)) for k in sorted(grp[0]["c"]) if k != "SQ_WAVES"))
PY
