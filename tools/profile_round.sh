#!/bin/bash
# The round's evidence, collected on the GPU box (gpurun -- 'bash tools/profile_round.sh'):
#   1. plain bench line (the driver's command)               -> gpurun_out/prof/bench_plain.json
#   2. rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/prof/stats/bench_kernel_stats.csv (+ bench line under rocprof)
#   3. PMC pass FETCH_SIZE (own run, kernel-trace only)      -> gpurun_out/prof/fetch.csv.json   (per-kernel sums)
#   4. PMC pass WRITE_SIZE (own run)                         -> gpurun_out/prof/write.csv.json
#   5. PMC pass SQ instruction counters (own run)            -> gpurun_out/prof/sq_a.csv.json
#   6. PMC pass SQ wait / active counters (own run)          -> gpurun_out/prof/sq_b.csv.json
#   7. launch floor micro-benchmark                          -> gpurun_out/prof/launch_floor.txt
#   7b. dependent-kernel boundary without events (stream / hipGraph) -> gpurun_out/prof/launch_chain.txt
#   8. NN-scan micro-benchmark                               -> gpurun_out/prof/nn_bench.txt
# then, back in the container: python tools/summarize_profiles.py rNN
R=/root/repo
O=$R/gpurun_out/prof
# the micro-benchmarks are built artefacts (git-ignored): a fresh checkout has none -- build what is missing here (hipcc is on the box)
for b in launch_floor launch_chain issue barrier; do
  [ -x $R/tools/micro/$b.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/micro/$b.bin $R/tools/micro/$b.hip > /dev/null 2>&1
done
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_plain.json 2> $O/bench_plain.err < /dev/null
CMD="python $R/bench.py --steps 3 --warmup 1 --units 16 --no-cpu --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.log < /dev/null
find $O/stats -name "bench_kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats
agg() {   # $1 = tag, rest = counters: one PMC pass, condensed per kernel on the spot (the raw CSV is large)
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- $CMD > /dev/null 2> $O/$tag.log < /dev/null
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$O/$tag.csv.json" <<'PY'
import csv, sys, collections, json, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace("void ", "")
    m = re.match(r"((?:lq::)?\w+(?:<[^(]*>)?)", n)
    return m.group(1) if m else n[:60]
# the LAST launches of each kernel are the steady state (bench's windowed loop); the first ones grow the tree
per = collections.defaultdict(list)
disp = {}
for r in rows:
    k = short(r["Kernel_Name"])
    d = r["Dispatch_Id"]
    if d not in disp:
        disp[d] = dict(kernel=k, start=int(r["Start_Timestamp"]), end=int(r["End_Timestamp"]), grid=int(r.get("Grid_Size", 0) or 0), c={})
    disp[d]["c"][r["Counter_Name"]] = disp[d]["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for d in disp.values():
    per[d["kernel"]].append(d)
out = {}
for k, ds in per.items():
    ds.sort(key=lambda d: d["start"])
    tail = ds[len(ds) // 2:]                       # second half of the launches = steady state
    names = sorted({n for d in tail for n in d["c"]})
    out[k] = dict(launches_total=len(ds), launches_steady=len(tail),
                  avg_ns=sum(d["end"] - d["start"] for d in tail) / len(tail),
                  avg_grid_threads=sum(d["grid"] for d in tail) / len(tail),
                  per_launch={n: sum(d["c"].get(n, 0.0) for d in tail) / len(tail) for n in names})
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(sys.argv[2], len(rows), "rows", len(out), "kernels")
PY
  rm -rf /tmp/pmc_$tag
}
agg fetch FETCH_SIZE
agg write WRITE_SIZE
agg sq_a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
agg sq_b SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
$R/tools/micro/launch_floor.bin > $O/launch_floor.txt 2>&1
$R/tools/micro/launch_chain.bin > $O/launch_chain.txt 2>&1
timeout 200 python $R/tools/nn_bench.py --reps 200 > $O/nn_bench.txt 2>&1
ls -la $O
tail -c 400 $O/bench_plain.json
# 9. what an instruction costs when a wavefront has a SIMD to itself; cost of a workgroup barrier hand-off
$R/tools/micro/issue.bin > $O/issue.txt 2>&1
$R/tools/micro/barrier.bin > $O/barrier.txt 2>&1
# 10. the N > 1 code path on this one GPU: native sharded loop, real RCCL communicator, world of one (+ its two extra curves)
LQRRT_FORCE_SHARDED=1 timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_forced_sharded.json 2> $O/bench_forced_sharded.err < /dev/null
# 11. every BASELINE configuration + the two Riccati systems; busy / gap breakdown of the loop's kernel timeline
timeout 600 python $R/tools/run_configs.py 2>&1 | grep -v amdgpu.ids > $O/configs.txt
bash $R/tools/timeline.sh > /dev/null 2>&1; python $R/tools/timeline_report.py $R/gpurun_out/kt_tail.csv > $O/timeline.txt 2>&1
