"""bench.py keeps its contract: one JSON line (the last line of stdout) with the driver's fields, the roofline and,
when asked for, the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    return json.loads(lines[-1])                     # the JSON line is the LAST line of stdout


def test_bench_line_fields():
    d = _run(["--cpu-seconds", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e4                                          # north_star's floor, by a wide margin in practice
    assert abs(d["ms_per_step"] * d["value"] / 1e3 - d["config"]["attempts_per_step"]) < 1e-6 * d["config"]["attempts_per_step"] + 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches"] > 0 and r["avg_launch_us"] > 0
    assert 0 < r["compute"]["frac"] < 1 and 0 < r["compute"]["frac_of_non_fma_peak"] < 1      # the binding resource: fp64 issue
    assert d["timed_region_s"] > 0.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert d["synchronous_mode"]["value"] > d["value"]             # the optional mode is an extra, never the metric
    # measured by the run itself: shader clock and per-FMA issue cost of a lone wavefront; mean committed wave
    assert 1500.0 < d["effective_mhz"] < 2600.0
    ck = d["steer_kernel"]["clock"]
    assert 3.5 < ck["cycles_per_dependent_fp64_fma"] < 9.0 and 1.0 < ck["ns_per_independent_fp64_fma"] < 5.0
    # the cap in force (exact-mode controller: the fused rounds' 256), not the command line's upper bound
    assert 8 <= d["config"]["mean_wave"] <= d["config"]["wave_cap"] == 256 and d["config"]["wave_cap_cli"] == 1024
    sp = d["seed_spread"]                                            # tree-to-tree spread on the line itself
    assert set(sp["by_sample_seed"]) == {"1", "2", "3", "4"} and sp["min"] <= sp["mean"] <= sp["max"] and sp["min"] > 1e4
    rp = d["repeats"]
    assert rp["regions"] == 3 and rp["min"] <= rp["median"] <= rp["max"] and rp["min"] <= d["value"] <= rp["max"]
    assert rp["max"] < 1.5 * rp["min"]                              # back-to-back regions on one box agree
    # whatever is copied from a committed profile says so
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic_static_from"].startswith("profiles/")
    # round 6, all live: the kernel that holds the GPU time, the dependency bound the run measures for it, the scan against its own best
    dk = d["dominant_kernel"]
    assert "k_steer" in dk["name"] and 0.7 < dk["share_of_event_timed_gpu_time"] < 1.0 and 2.0 < dk["launches_per_wave"] < 12.0
    cb = d["chain_bound"]
    assert 20.0 < cb["chain_slots_per_1024"] <= cb["launches_run_per_1024"]          # a lower bound on what the loop runs
    assert 8.0 < cb["rollout_probe"]["full_horizon_rollout_launch_us"] < 60.0 and cb["rollout_probe"]["full_horizon_launches"] > 0
    assert 0.1 < cb["frac_of_chain_bound"] < 1.0
    ob = d["roofline"]["vs_own_best"]
    assert ob["pairs_per_s_best"] > ob["pairs_per_s_in_loop"] > 0 and abs(ob["frac"] - ob["pairs_per_s_in_loop"] / ob["pairs_per_s_best"]) < 1e-12
    assert not any(k.endswith("static_from") for k in list(dk) + list(cb) + list(ob))
    cm = c["callback_mode"]                                             # plain Python plugins, nearest-neighbour stage on the GPU
    assert cm["value"] > 3 * c["value"] and cm["vs_numpy_oracle_same_plugins"] == cm["value"] / c["value"]


def test_bench_sharded_code_path_world_of_one():
    d = _run(["--no-cpu", "--no-extras"], env={"LQRRT_FORCE_SHARDED": "1"})
    assert d["value"] > 1e4 and "cpu_baseline" not in d
    d = _run(["--no-cpu", "--no-extras", "--shard", "tree"], env={"LQRRT_FORCE_SHARDED": "1"})
    assert d["value"] > 1e4


def test_bench_sharded_run_reports_four_curves():
    """A sharded run (here: a world of one through the native loop and a real RCCL communicator) prints, next to the exact-mode
    value, the synchronous mode sample-sharded, BASELINE config 5 tree-sharded and -- the one that scales -- a fleet of independent
    planners per device."""
    d = _run(["--no-cpu", "--units", "16"], env={"LQRRT_FORCE_SHARDED": "1"})
    assert "native loop" in d["config"]["parallelism"]
    assert d["synchronous_mode"]["value"] > d["value"] and "sample-sharded" in d["synchronous_mode"]["parallelism"]
    c5 = d["config5_tree_sharded"]
    assert c5["value"] > 1e4 and c5["workload"] == "double_integrator_100k_boxes_50k"
    fl = d["fleet_per_device"]
    assert fl["trees"] == 64 and fl["trees_per_device"] == 64 and fl["scaling"] == "weak" and fl["value"] > 3 * d["value"]


def test_bench_config5_workload():
    """BASELINE config 5 (100k boxes, 50k-node window) through the same harness, single GPU and the tree-sharded
    N>1 code path in a world of one."""
    d = _run(["--workload", "cfg5", "--units", "4", "--cpu-seconds", "3"])
    assert d["config"]["workload"] == "double_integrator_100k_boxes_50k" and d["config"]["nodes_window"] == [47500, 52500]
    assert d["value"] > 1e4 and d["cpu_baseline"]["c_oracle_value"] > 0
    d = _run(["--workload", "cfg5", "--units", "4", "--no-cpu", "--no-extras"], env={"LQRRT_FORCE_SHARDED": "1"})
    assert d["value"] > 1e4


def test_bench_gpus_2_spawns_its_own_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` without a launcher (the form of the driver's N = 1 command): two ranks over RCCL when the box has
    two devices, otherwise a non-zero exit that says why -- never an N = 1 line under an N = 2 command."""
    import torch
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--units", "16"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if torch.cuda.device_count() >= 2:
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
        assert d["n_gpus"] == 2 and "x2" in d["config"]["parallelism"] and d["value"] > 1e4
    else:
        assert out.returncode != 0
        assert "--gpus 2" in out.stderr and "only 1 HIP device" in out.stderr
        assert not any(l.lstrip().startswith("{") for l in out.stdout.splitlines())


def test_bench_refuses_a_world_that_differs_from_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
