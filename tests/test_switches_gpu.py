"""Speed levers are never results: the exact-mode loop grown to 7,000 nodes of demo_boat_advanced (about 25,000 attempts, 250
waves, 1,200 repair rounds -- long enough for several sample-pool blocks to be prepared ahead and for dozens of goal hits) gives the
same tree, bit for bit, whatever the environment switches say.  Every setting runs in a process of its own: the switches are
read once."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import lqrrt_amd
from lqrrt_amd.engine import Engine
s = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
eng = Engine(s, capacity=7000 + 1024 + 64, max_wave=1024)
kw = s.plan_kwargs
eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
space = np.array(s.sample_space, dtype=np.float64)
eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
st = np.random.RandomState(1).get_state()
eng.set_mt19937(st[1], st[2])
eng.tree_reset(s.x0)
stats = eng.extend(1024, node_limit=7000)
h = hashlib.sha256()
for a in (eng.states(), eng.gains(), eng.parents()):
    h.update(np.ascontiguousarray(a).tobytes())
print("TREE", eng.size, stats.attempts, stats.goal_hits, h.hexdigest())
""" % ROOT


def _run(env_extra):
    env = dict(os.environ)
    for k in ("LQRRT_FUSED_ROUNDS", "LQRRT_STEER_WAVEFRONTS", "LQRRT_NN_WG4", "LQRRT_EXACT_WAVE_MAX", "LQRRT_MATRIX_MAX_W"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("TREE")][-1]
    return line


def test_switches_do_not_change_the_tree():
    base = _run({})
    size, attempts, hits = (int(x) for x in base.split()[1:4])
    assert size > 7000 and attempts > 20000 and hits > 10            # the run is long enough to go through every mechanism
    # (round 6 removed the switches whose A/B is settled -- LQRRT_SECOND_CHOICE, _IGNORE_PATCH, _REFILL_AHEAD, _SHARD_FOLD -- with the
    #  code paths they selected; what is left are forms that are all still in use somewhere: large waves, sharded waves, large tables)
    for env in ({"LQRRT_FUSED_ROUNDS": "0"}, {"LQRRT_STEER_WAVEFRONTS": "2"}, {"LQRRT_NN_WG4": "1"},
                {"LQRRT_EXACT_WAVE_MAX": "1024", "LQRRT_MATRIX_MAX_W": "128"},
                {"LQRRT_FUSED_ROUNDS": "0", "LQRRT_STEER_WAVEFRONTS": "2", "LQRRT_NN_WG4": "1"}):
        assert _run(env) == base, env


RICCATI_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import lqrrt_amd
from lqrrt_amd.engine import Engine
name, nodes = sys.argv[1], int(sys.argv[2])
s = lqrrt_amd.systems.SYSTEMS[name](0)
eng = Engine(s, capacity=nodes + 256 + 64, max_wave=256)
kw = s.plan_kwargs
eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
space = np.array(s.sample_space, dtype=np.float64)
eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
st = np.random.RandomState(2).get_state()
eng.set_mt19937(st[1], st[2])
eng.tree_reset(s.x0)
stats = eng.extend(256, node_limit=nodes)
h = hashlib.sha256()
for a in (eng.states(), eng.gains(), eng.parents(), eng.edge_lengths()):
    h.update(np.ascontiguousarray(a).tobytes())
print("TREE", eng.size, stats.attempts, stats.fix_rounds, h.hexdigest())
""" % ROOT


@pytest.mark.parametrize("name,nodes", [("boat_novice_lqr", 400), ("pendulum_lqr", 250)])
def test_riccati_rollouts_one_or_four_wavefronts_same_tree(name, nodes):
    """The Riccati gain inside a rollout is computed by one wavefront (dare_lqr<S, 64>) or by four that share it (dare_lqr<S, 256>:
    the elimination in the first, G / H updates and the convergence test spread over two); states, gains, parents and edge lengths of
    the grown tree are the same bits either way.  (The default is four for every Riccati system since round 5.)"""
    def run(dw):
        env = dict(os.environ)
        env.pop("LQRRT_DARE_WAVEFRONTS", None)
        if dw:
            env["LQRRT_DARE_WAVEFRONTS"] = str(dw)
        out = subprocess.run([sys.executable, "-c", RICCATI_SCRIPT, name, str(nodes)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        return [l for l in out.stdout.splitlines() if l.startswith("TREE")][-1]
    base = run(None)
    assert int(base.split()[1]) > nodes
    assert run(1) == base
    assert run(4) == base


def _grow(system_name, nodes, seed=1, xcds=None, wave=1024):
    import numpy as np
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.SYSTEMS[system_name](0)
    eng = Engine(s, capacity=nodes + 1024 + 64, max_wave=1024)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(seed).get_state()
    eng.set_mt19937(st[1], st[2])
    if xcds is not None:
        eng.set_cu_mask(xcds=xcds)
    eng.tree_reset(s.x0)
    stats = eng.extend(wave, node_limit=nodes)
    h = hashlib.sha256()
    for a in (eng.states(), eng.gains(), eng.parents()):
        h.update(np.ascontiguousarray(a).tobytes())
    out = (eng.size, stats.attempts, stats.waves, stats.fix_rounds, stats.resteers, stats.goal_hits, h.hexdigest())
    eng.close()
    return out


def test_round_counts_do_not_depend_on_timing():
    """with the second-choice rule a sample may steer from a record whose owner re-steers in the same launch.
    Round 4 read that record in place -- a torn read whose rollout was always redone (same tree) but which made the COUNT of repair
    rounds and re-steers depend on timing, and the wave-size controller (hence the all-gather sizes of a sharded world) with it.
    Round 5 double-buffers the record heads (RoundArgs::head2).  Here: the same 4,000-node growth alone and three times with two
    other planners hammering the same GPU from other threads -- waves, rounds, re-steers and the tree must be identical."""
    import threading
    alone = _grow("boat_advanced", 4000)
    assert alone[3] > 300 and alone[4] > 2000                      # enough rounds and re-steers for a race to show
    stop = threading.Event()

    def noise(seed):
        while not stop.is_set():
            _grow("boat_advanced", 1500, seed=seed)
    threads = [threading.Thread(target=noise, args=(k,)) for k in (7, 8)]
    for th in threads:
        th.start()
    try:
        for _ in range(3):
            assert _grow("boat_advanced", 4000) == alone
    finally:
        stop.set()
        for th in threads:
            th.join()


def test_cu_mask_changes_nothing_but_placement():
    """lqrrt_engine_set_cu_mask: the native loop on a stream restricted to one / two XCDs grows the same tree with the same counts."""
    base = _grow("boat_advanced", 2500)
    assert _grow("boat_advanced", 2500, xcds=[0]) == base
    assert _grow("boat_advanced", 2500, xcds=[2, 5]) == base
    assert _grow("car", 1200, wave=256, xcds=[7]) == _grow("car", 1200, wave=256)
