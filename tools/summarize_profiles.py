#!/usr/bin/env python
"""
Turns the rocprofv3 CSVs of one round (kernel-trace --stats pass, FETCH_SIZE pass, WRITE_SIZE pass)
into the committed summaries under profiles/:
  profiles/rNN_kernel_stats.csv      -- rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/rNN_nn_traffic.json       -- per-launch HBM traffic of the NN scan from the PMC passes

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so fetch bytes = 2 * FETCH_SIZE * 1024.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, name_part):
    rows = [r for r in csv.DictReader(open(path)) if name_part in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    return len(rows), sum(vals), sum(durs)


def steady_state_ns(path, name_part, last):
    """Average duration of the last `last` launches of a kernel (the bench's windowed loop at ~10k nodes; the
    launches before them belong to the tree-growth phase, where the node table is smaller)."""
    rows = [r for r in csv.DictReader(open(path)) if name_part in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-last:]
    return sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / float(len(rows)), len(rows)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(src, "prof_r1", "bench_kernel_stats.csv"), os.path.join(out, "%s_kernel_stats.csv" % rnd))
    kern = "k_nn_scan<lq::BoatAdvanced, 0, false>"
    n_f, fetch_kib, dur_f = per_kernel(os.path.join(src, "prof_r1_fetch", "bench_counter_collection.csv"), kern)
    n_w, write_kib, dur_w = per_kernel(os.path.join(src, "prof_r1_write", "bench_counter_collection.csv"), kern)
    summary = {
        "kernel": kern, "command": "python bench.py --steps 20 --warmup 3 --no-cpu --no-extras (whole process: tree growth + warm-up + timed steps)",
        "launches": n_f,
        "FETCH_SIZE_KiB_per_launch": fetch_kib / n_f, "WRITE_SIZE_KiB_per_launch": write_kib / n_w,
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": (2.0 * fetch_kib / n_f + write_kib / n_w) * 1024.0,
        "avg_launch_ns_fetch_pass": dur_f / n_f, "avg_launch_ns_write_pass": dur_w / n_w,
    }
    ss, cnt = steady_state_ns(os.path.join(src, "prof_r1_fetch", "bench_counter_collection.csv"), kern, 180)
    summary["steady_state_avg_launch_ns"] = ss
    summary["steady_state_launches"] = cnt
    summary["steady_state_note"] = ("last %d launches of the process = bench.py's windowed loop at 9.5k-10.5k nodes (the same launches "
                                    "bench.py times with dispatch-attached HIP events)" % cnt)
    with open(os.path.join(out, "%s_nn_traffic.json" % rnd), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
