#!/usr/bin/env python
"""
Condenses what tools/profile_round.sh left under gpurun_out/prof/ (rocprofv3 passes of bench.py on the GPU box) into the
committed summaries under profiles/:

  profiles/rNN_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the bench command (verbatim)
  profiles/rNN_bench.json            the plain bench line of the same build (driver's command)
  profiles/rNN_nn_traffic.json       per-launch HBM traffic of the NN scan (FETCH_SIZE and WRITE_SIZE passes)
  profiles/rNN_nn_pmc.json           per-launch instruction and wait counters of the NN scan, with derived issue utilisation
  profiles/rNN_steer_pmc.json        the same for k_steer (the kernel that holds ~70 % of GPU time)
  profiles/rNN_decide_pmc.json       the same for k_decide
  profiles/rNN_launch_floor.txt      duration an EMPTY kernel shows under the same dispatch-attached events
  profiles/rNN_nn_bench.txt          NN-scan micro-benchmark (tools/nn_bench.py)
  profiles/rNN_issue.txt             what one instruction costs a wavefront that has a SIMD to itself (tools/micro/issue.hip)
  profiles/rNN_barrier.txt           workgroup barrier / LDS hand-off between wavefronts (tools/micro/barrier.hip)
  profiles/rNN_teacher_*.json        teacher-forced parity results written by tests/test_teacher_gpu.py on the GPU box

Units.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so fetch bytes = 2 * FETCH_SIZE * 1024.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles (4 shader cycles) summed over wavefronts.  A wave64 fp64 VALU instruction occupies its SIMD for
4 cycles (78.6 TFLOP/s fp64 vector peak = 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz).
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_GHZ, SIMDS = 2.4, 1024


def kernel(d, part):
    """Entry of the kernel whose name contains `part`; several instantiations (k_steer's 2 / 3 / 4-wavefront variants)
    are merged, weighted by their launch counts."""
    ks = [k for k in d if part in k]
    if not ks:
        return None
    if len(ks) == 1:
        return d[ks[0]]
    n = sum(d[k]["launches_steady"] for k in ks)
    out = dict(launches_total=sum(d[k]["launches_total"] for k in ks), launches_steady=n, merged=ks,
               avg_ns=sum(d[k]["avg_ns"] * d[k]["launches_steady"] for k in ks) / n,
               avg_grid_threads=sum(d[k].get("avg_grid_threads", 0) * d[k]["launches_steady"] for k in ks) / n)
    names = sorted({c for k in ks for c in d[k]["per_launch"]})
    out["per_launch"] = {c: sum(d[k]["per_launch"].get(c, 0.0) * d[k]["launches_steady"] for k in ks) / n for c in names}
    return out


def pmc(a, b, part, note):
    ka, kb = kernel(a, part), kernel(b, part)
    if ka is None:
        return None
    p, q = ka["per_launch"], (kb or {}).get("per_launch", {})
    waves = max(p.get("SQ_WAVES", 1.0), 1.0)
    out = dict(kernel=part, note=note, launches_steady=ka["launches_steady"], avg_launch_us=ka["avg_ns"] / 1e3,
               waves_per_launch=waves, per_launch=p, per_launch_wait_pass=q,
               per_wave={k: v / waves for k, v in p.items() if k != "SQ_WAVES"})
    valu = p.get("SQ_INSTS_VALU", 0.0)
    out["valu_issue_utilisation"] = valu * 4.0 / (ka["avg_ns"] * CLOCK_GHZ * SIMDS)       # share of all SIMD issue cycles of the launch
    out["valu_issue_note"] = "SQ_INSTS_VALU x 4 cycles / (launch duration x 2.4 GHz x 1024 SIMDs); fp64 wave64 = 4 cycles per instruction"
    if q.get("SQ_WAIT_ANY") is not None:
        tot = q.get("SQ_WAIT_ANY", 0) + q.get("SQ_WAIT_INST_ANY", 0) + q.get("SQ_ACTIVE_INST_ANY", 0)
        if tot > 0:
            out["wave_time_split"] = {"waiting (s_waitcnt / dependency)": q.get("SQ_WAIT_ANY", 0) / tot,
                                      "issue stall": q.get("SQ_WAIT_INST_ANY", 0) / tot,
                                      "issuing": q.get("SQ_ACTIVE_INST_ANY", 0) / tot}
    return out


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "prof")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(out, "%s_kernel_stats.csv" % rnd))
    for a, b in (("bench_plain.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                 ("launch_floor.txt", "launch_floor.txt"), ("launch_chain.txt", "launch_chain.txt"), ("nn_bench.txt", "nn_bench.txt"),
                 ("issue.txt", "issue.txt"), ("barrier.txt", "barrier.txt"), ("bench_forced_sharded.json", "bench_forced_sharded.json"),
                 ("configs.txt", "configs.txt"), ("timeline.txt", "timeline.txt")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(out, "%s_%s" % (rnd, b)))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "teacher_*.json")):
        shutil.copy(f, os.path.join(out, "%s_%s" % (rnd, os.path.basename(f))))
    load = lambda n: json.load(open(os.path.join(src, n)))
    fetch, write, sq_a, sq_b = load("fetch.csv.json"), load("write.csv.json"), load("sq_a.csv.json"), load("sq_b.csv.json")
    kern = "k_nn_scan<lq::BoatAdvanced, 0, false"      # both instantiations of the tree scan (with / without ignore patch), merged by launch count
    kf, kw = kernel(fetch, kern), kernel(write, kern)
    traffic = {
        "kernel": kern, "command": "python bench.py --steps 3 --warmup 1 --units 16 --no-cpu --no-extras (3 x 16,384 attempts in the 10k-node window)",
        "launches": kf["launches_steady"], "launches_note": "second half of the process's launches = the windowed loop at 9.5k-10.5k nodes",
        "FETCH_SIZE_KiB_per_launch": kf["per_launch"]["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": kw["per_launch"]["WRITE_SIZE"],
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": (2.0 * kf["per_launch"]["FETCH_SIZE"] + kw["per_launch"]["WRITE_SIZE"]) * 1024.0,
        "steady_state_avg_launch_ns": 0.5 * (kf["avg_ns"] + kw["avg_ns"]),
        "note": "writes dominate: every (sample, node chunk) pair stores ONE 16-byte partial minimum (round 6; two stores of 8 + 4 bytes before: WRITE_SIZE 3.37 -> 2.30 MB per launch), sample-major so that the steer prologue reads a sample's partials contiguously; every lane's store is its own memory transaction"
                "prologue reads a sample's partials contiguously; the scattered 8-byte stores count as 64-byte memory transactions",
    }
    json.dump(traffic, open(os.path.join(out, "%s_nn_traffic.json" % rnd), "w"), indent=1)
    for name, part, note in (
            ("nn_pmc", kern, "tree scan, average wave of the bench loop (W ~ 230 samples x 10k nodes)"),
            ("steer_pmc", "k_steer<lq::BoatAdvanced, 0,", "all steer launches of the loop (2-, 3- and 4-wavefront instantiations merged by launch count): "
                                                           "speculative launches and fused repair rounds (W workgroups each; most wavefronts of a round only "
                                                           "decide and leave), incl. the launch that performs the append"),
            ("decide_pmc", "k_decide", "one workgroup; thread t scans column t of the in-wave cost matrix")):
        r = pmc(sq_a, sq_b, part, note)
        if r is not None:
            json.dump(r, open(os.path.join(out, "%s_%s.json" % (rnd, name)), "w"), indent=1)
            print(name, "avg %.1f us, VALU/wave %.0f, SALU/wave %.0f, issue utilisation %.3f, split %s" % (
                r["avg_launch_us"], r["per_wave"].get("SQ_INSTS_VALU", 0), r["per_wave"].get("SQ_INSTS_SALU", 0),
                r["valu_issue_utilisation"], {k: round(v, 2) for k, v in r.get("wave_time_split", {}).items()}))
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
