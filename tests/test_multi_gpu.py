"""GPU: lqrrt_engine_extend_multi -- n independent engines advanced in lock step by one native loop, two launches per step whose
grids span the engines (include/lqrrt_hip.h; csrc/engine_multi.hpp; kernels.hpp k_nn_scan_multi / k_steer_multi).

The claim to check is simple: an engine's tree in a multi call is EXACTLY the tree it grows alone with lqrrt_engine_extend --
parents, states, gains, every edge row, ignore set, best plan -- and so are its counts (waves, repair rounds, re-steers, attempts,
candidate rows): the per-engine protocol is the one of the fused repair rounds, only the launches are shared.  Checked for four
sample seeds of the headline problem, for the car, for engines that stop at different times, and against the sequential C oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _make(name, cap, wave, seed):
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.SYSTEMS[name](0)
    eng = Engine(s, capacity=cap, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(seed).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    return s, eng


def _same(a, b):
    assert a.size == b.size
    np.testing.assert_array_equal(a.parents(), b.parents())
    np.testing.assert_array_equal(a.states(), b.states())
    np.testing.assert_array_equal(a.gains(), b.gains())
    np.testing.assert_array_equal(a.edge_lengths(), b.edge_lengths())
    np.testing.assert_array_equal(a.ignored(), b.ignored())
    xa, ua, la = a.edges()
    xb, ub, lb = b.edges()
    for i in range(a.size):
        np.testing.assert_array_equal(xa[i, :la[i]], xb[i, :lb[i]])
        np.testing.assert_array_equal(ua[i, :la[i]], ub[i, :lb[i]])
    assert a.plan_best() == b.plan_best()


def _counts(st):
    return (st.attempts, st.accepted, st.waves, st.fix_rounds, st.resteers, st.goal_hits, st.candidates, st.tree_size, st.stop_reason)


@pytest.mark.parametrize("name,nodes,wave,seeds", [("boat_advanced", 3000, 256, (1, 2, 3, 4)), ("car", 1500, 128, (5, 6, 7)),
                                                   ("boat_intermediate", 700, 64, (1, 9)), ("pendulum", 150, 64, (1, 2, 3, 4, 5))])
def test_every_tree_of_a_multi_call_is_the_tree_its_engine_grows_alone(name, nodes, wave, seeds):
    from lqrrt_amd.engine import Engine
    alone, stats_alone = [], []
    for sd in seeds:
        _, e = _make(name, nodes + wave + 8, wave, sd)
        stats_alone.append(_counts(e.extend(wave, node_limit=nodes)))
        alone.append(e)
    multi = [_make(name, nodes + wave + 8, wave, sd)[1] for sd in seeds]
    stats = Engine.extend_multi(multi, wave, node_limit=nodes)
    for k in range(len(seeds)):
        assert _counts(stats[k]) == stats_alone[k], (k, _counts(stats[k]), stats_alone[k])
        _same(multi[k], alone[k])
    # a second call continues every tree (warm engines, uploaded prototypes, sample pools in mid-stream)
    more = nodes + 200
    for e in alone:
        e.close()
    alone2 = []
    for sd in seeds:
        _, e = _make(name, more + wave + 8, wave, sd)
        e.extend(wave, node_limit=more)
        alone2.append(e)
    # (capacity was sized for `nodes`: fresh engines for the longer run, grown in two multi calls)
    multi2 = [_make(name, more + wave + 8, wave, sd)[1] for sd in seeds]
    Engine.extend_multi(multi2, wave, node_limit=nodes)
    Engine.extend_multi(multi2, wave, node_limit=more)
    for k in range(len(seeds)):
        _same(multi2[k], alone2[k])


def test_engines_that_stop_at_different_times_and_the_c_oracle():
    """max_attempts ends every engine after the same number of attempts but after different numbers of ticks; until_size and
    stop_on_goal end them at different times.  And the multi loop against the sequential C oracle directly."""
    import coracle
    from lqrrt_amd.engine import Engine
    seeds = (1, 2, 3, 4, 5, 6)
    engs = [_make("boat_advanced", 2600, 256, sd) for sd in seeds]
    stats = Engine.extend_multi([e for _, e in engs], 256, max_attempts=6000)
    for (s, e), st, sd in zip(engs, stats, seeds):
        assert st.attempts == 6000 and st.stop_reason == 1
        o = coracle.make(s, 2600, seed=sd)
        o.extend(max_iters=6000)
        assert e.size == o.size
        np.testing.assert_array_equal(e.parents(), o.parents())
        np.testing.assert_array_equal(e.states(), o.states())
    goal = [_make("car", 1400, 128, sd) for sd in (11, 12, 13)]
    st_goal = Engine.extend_multi([e for _, e in goal], 128, node_limit=1200, stop_on_goal=True)
    for (s, e), st, sd in zip(goal, st_goal, (11, 12, 13)):
        _, ref = _make("car", 1400, 128, sd)
        rs = ref.extend(128, node_limit=1200, stop_on_goal=True)
        assert _counts(st) == _counts(rs)
        _same(e, ref)


def test_multi_call_argument_checks():
    from lqrrt_amd.engine import Engine
    _, a = _make("car", 600, 64, 1)
    _, b = _make("boat_advanced", 600, 64, 1)
    with pytest.raises(ValueError):
        Engine.extend_multi([a, b], 64, node_limit=100)            # two models
    with pytest.raises(ValueError):
        Engine.extend_multi([a, a], 64, node_limit=100)            # the same engine twice
    with pytest.raises(ValueError):
        Engine.extend_multi([], 64)
    _, c = _make("boat_novice_lqr", 300, 64, 1)
    with pytest.raises(ValueError):
        Engine.extend_multi([c], 64, node_limit=50)                # Riccati gain: not served by this loop


def test_groups_on_host_threads_change_nothing():
    """Calls with four or more engines are cut into groups, each advanced by its own host thread on its own stream (engine_multi.hpp):
    nine engines = four groups of 3 / 2 / 2 / 2, launches overlapping on the GPU -- every tree still the one its engine grows alone;
    and the same call forced onto one thread and onto three (LQRRT_MULTI_THREADS, read once per process: child processes)."""
    import subprocess
    from lqrrt_amd.engine import Engine
    seeds = tuple(range(21, 30))
    alone, stats_alone = [], []
    for sd in seeds:
        _, e = _make("boat_advanced", 1900, 256, sd)
        stats_alone.append(_counts(e.extend(256, node_limit=1600)))
        alone.append(e)
    multi = [_make("boat_advanced", 1900, 256, sd)[1] for sd in seeds]
    stats = Engine.extend_multi(multi, 256, node_limit=1600)
    for k in range(len(seeds)):
        assert _counts(stats[k]) == stats_alone[k]
        _same(multi[k], alone[k])
    code = r"""
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import test_multi_gpu as t
from lqrrt_amd.engine import Engine
engs = [t._make("car", 1100, 128, sd)[1] for sd in (1, 2, 3, 4, 5, 6)]
st = Engine.extend_multi(engs, 128, node_limit=900)
h = hashlib.sha256()
for e, s in zip(engs, st):
    for a in (e.states(), e.gains(), e.parents()):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(repr(t._counts(s)).encode())
print("HASH", h.hexdigest())
""" % (ROOT, os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for nthreads in ("1", "3", "6"):
        env = dict(os.environ, LQRRT_MULTI_THREADS=nthreads)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        hashes.append([l for l in out.stdout.splitlines() if l.startswith("HASH")][-1])
    assert hashes[0] == hashes[1] == hashes[2], hashes


def test_many_engines_take_the_two_wavefront_rollout_and_grow_the_same_trees():
    """From 24 engines on, a multi call rolls the heading-torque boats out with two wavefronts per rollout instead of three (fewer
    wavefront slots per rollout: engine_multi.hpp, profiles/r05_multi.txt section 4).  The form of the rollout is not allowed to
    change a bit: 26 small trees of the headline problem, each against the tree its engine grows alone (three wavefronts)."""
    from lqrrt_amd.engine import Engine
    seeds = tuple(range(41, 67))
    alone, stats_alone = [], []
    for sd in seeds:
        _, e = _make("boat_advanced", 700, 128, sd)
        stats_alone.append(_counts(e.extend(128, node_limit=420)))
        alone.append(e)
    multi = [_make("boat_advanced", 700, 128, sd)[1] for sd in seeds]
    stats = Engine.extend_multi(multi, 128, node_limit=420)
    for k in range(len(seeds)):
        assert _counts(stats[k]) == stats_alone[k]
        _same(multi[k], alone[k])


def test_update_plans_gives_every_planner_its_own_update_plan():
    """lqrrt_amd.update_plans: several Planner objects through shared native calls.  Each planner's tree, plan and statistics are
    those of its own update_plan from the same sample stream (fake clock: the plans end by the node limit / at the first goal
    hit after min_time exactly like the solo ones)."""
    import lqrrt_amd

    def mk(max_nodes, **over):
        s = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
        cons = lqrrt_amd.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
        kw = dict(s.plan_kwargs)
        kw.update(error_tol=s.error_tol, erf=s.erf, min_time=2, max_time=3, max_nodes=max_nodes, goal0=s.goal,
                  sys_time=lambda: 0.0, printing=False, wave_size=256)
        kw.update(over)
        return s, lqrrt_amd.Planner(s.dynamics, s.lqr, cons, **kw)

    seeds = (3, 4, 5, 6, 7)
    solo = []
    for sd in seeds:
        s, p = mk(1200)
        np.random.seed(sd)
        assert p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10) is False       # ended by the node limit
        solo.append(p)
    fleet = [mk(1200) for _ in seeds]
    before = np.random.get_state()[1].copy()
    res = lqrrt_amd.update_plans([dict(planner=p, x0=s.x0, sample_space=s.sample_space, goal_bias=s.goal_bias, xrand_gen=10, seed=sd)
                                  for (s, p), sd in zip(fleet, seeds)])
    assert res == [False] * len(seeds)
    assert np.array_equal(np.random.get_state()[1], before)                           # per-planner streams: np.random untouched
    for (s, p), q in zip(fleet, solo):
        assert p.tree.size == q.tree.size and list(p.tree.pID) == list(q.tree.pID)
        np.testing.assert_array_equal(p.tree.state, q.tree.state)
        assert p.plan_reached_goal == q.plan_reached_goal and list(p.node_seq) == list(q.node_seq) and p.T == q.T
        np.testing.assert_array_equal(np.array(p.x_seq), np.array(q.x_seq))
        np.testing.assert_array_equal(np.array(p.u_seq), np.array(q.u_seq))
        for key in ("attempts", "accepted", "candidates", "goal_hits", "tree_size"):       # (waves / rounds depend on where the calls end)
            assert p.stats[key] == q.stats[key], key
        np.testing.assert_array_equal(p.get_state(0.5 * p.T), q.get_state(0.5 * q.T))
    # a clock that lets the plans finish: every planner returns True with a plan that ends in its goal region, finish_on_goal included
    t = [0.0]

    def clock():
        t[0] += 0.05
        return t[0]
    fleet2 = [mk(60000, sys_time=clock, min_time=0.5, max_time=400.0) for _ in range(3)]
    res2 = lqrrt_amd.update_plans([dict(planner=p, x0=s.x0, sample_space=s.sample_space, goal_bias=s.goal_bias, seed=10 + k,
                                        finish_on_goal=(k == 0)) for k, (s, p) in enumerate(fleet2)])
    assert res2 == [True, True, True]
    for k, (s, p) in enumerate(fleet2):
        assert p.plan_reached_goal and p.node_seq == p.tree.climb(p.node_seq[-1])
        assert p._in_goal(p.x_seq[-1]) or k == 0
    np.testing.assert_array_equal(fleet2[0][1].tree.state[-1], np.array(fleet2[0][0].goal, dtype=np.float64))


def test_fleet_example_runs():
    """examples/fleet_gpu.py: real clock, a quarter of a second per plan -- four planners one by one, then together through update_plans;
    together they must get more attempts per second of wall clock than one by one (on a shared box: any gain at all)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "fleet_gpu.py"), "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("attempts per second")][-1]
    solo, joint = [float(tok) for tok in line.replace(",", " ").split() if tok[0].isdigit() and "e" in tok]
    assert joint > solo, line


def test_update_plans_mixed_jobs_on_the_car():
    """Jobs of one update_plans call need not look alike: the default sampler, one try per sample (xrand_gen = 1), a user sampling
    function, a different start state and finish_on_goal -- each planner still ends with what its own update_plan gives."""
    import lqrrt_amd

    def mk():
        s = lqrrt_amd.systems.SYSTEMS["car"](0)
        cons = lqrrt_amd.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
        kw = dict(s.plan_kwargs)
        kw.update(error_tol=s.error_tol, erf=s.erf, min_time=0, max_time=3, max_nodes=900, goal0=s.goal, sys_time=lambda: 0.0,
                  printing=False, wave_size=128)
        p = lqrrt_amd.Planner(s.dynamics, s.lqr, cons, **kw)
        p.force_arrive_max_steps = 5000
        return s, p

    def sampler_for(seed, s):
        rng = np.random.RandomState(seed)
        lo, hi = np.array(s.sample_space, dtype=np.float64).T
        goal = np.array(s.goal, dtype=np.float64)
        return lambda planner: goal.copy() if rng.uniform() < 0.3 else lo + (hi - lo) * rng.uniform(size=len(lo))

    s0 = mk()[0]
    x1 = np.array(s0.x0, dtype=np.float64) + np.array([1.0, -0.5, 0.1, 0.0, 0.0])[:s0.nstates]
    specs = [dict(goal_bias=s0.goal_bias, xrand_gen=10, seed=31),
             dict(goal_bias=s0.goal_bias, xrand_gen=1, seed=32, finish_on_goal=True),
             dict(xrand_gen="fn33"),
             dict(goal_bias=s0.goal_bias, seed=34, x0=x1)]
    outcomes = []
    for joint in (False, True):
        fleet = [mk() for _ in specs]
        jobs = []
        for (s, p), sp in zip(fleet, specs):
            j = dict(planner=p, x0=sp.get("x0", s.x0), sample_space=s.sample_space)
            j.update({k: v for k, v in sp.items() if k != "x0"})
            if j.get("xrand_gen") == "fn33":
                j["xrand_gen"] = sampler_for(33, s)
            jobs.append(j)
        if joint:
            res = lqrrt_amd.update_plans(jobs)
        else:
            res = []
            for j in jobs:
                kw = {k: v for k, v in j.items() if k not in ("planner", "x0", "sample_space", "seed")}
                if "seed" in j:
                    np.random.seed(j["seed"])
                res.append(j["planner"].update_plan(j["x0"], j["sample_space"], **kw))
        outcomes.append((res, [p for _, p in fleet]))
    (res_a, pa), (res_b, pb) = outcomes
    assert res_a == res_b
    for a, b in zip(pa, pb):
        assert a.tree.size == b.tree.size and list(a.tree.pID) == list(b.tree.pID)
        np.testing.assert_array_equal(a.tree.state, b.tree.state)
        assert a.plan_reached_goal == b.plan_reached_goal and list(a.node_seq) == list(b.node_seq)
        np.testing.assert_array_equal(np.array(a.x_seq), np.array(b.x_seq))
        np.testing.assert_array_equal(np.array(a.u_seq), np.array(b.u_seq))


def test_update_plans_groups_are_independent_fleets():
    """Planners of different devices form groups that run their own shared loops on host threads, with nothing exchanged between them
    (the multi-GPU form of this path that scales: one fleet per device).  On a one-GPU box the `group` key makes two groups of device
    0 stand in for two devices: every planner still grows the tree it grows alone, and groups may differ in what a group must share
    (here: the node limit)."""
    import lqrrt_amd

    def mk(max_nodes, device=0):
        s = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
        cons = lqrrt_amd.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
        kw = dict(s.plan_kwargs)
        kw.update(error_tol=s.error_tol, erf=s.erf, min_time=2, max_time=3, max_nodes=max_nodes, goal0=s.goal,
                  sys_time=lambda: 0.0, printing=False, wave_size=256, device=device)
        return s, lqrrt_amd.Planner(s.dynamics, s.lqr, cons, **kw)

    import torch
    two = torch.cuda.device_count() >= 2
    plan = [(900, 0, "a", 21), (900, 0, "a", 22), (900, 0, "a", 23), (1300, 1 if two else 0, "b", 24), (1300, 1 if two else 0, "b", 25)]
    solo = []
    for nodes, dev, _, sd in plan:
        s, p = mk(nodes, dev)
        np.random.seed(sd)
        assert p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10) is False
        solo.append(p)
    fleet = [mk(nodes, dev) for nodes, dev, _, _ in plan]
    res = lqrrt_amd.update_plans([dict(planner=p, x0=s.x0, sample_space=s.sample_space, goal_bias=s.goal_bias, xrand_gen=10, seed=sd, group=grp)
                                  for (s, p), (_, _, grp, sd) in zip(fleet, plan)])
    assert res == [False] * len(plan)
    for (s, p), q in zip(fleet, solo):
        assert p.tree.size == q.tree.size and list(p.tree.pID) == list(q.tree.pID)
        np.testing.assert_array_equal(p.tree.state, q.tree.state)
        assert list(p.node_seq) == list(q.node_seq) and p.T == q.T
    print("groups ran on %s" % ("two devices" if two else "one device (two groups of device 0)"))
    # without the group key the two node limits cannot share native calls: refused before any planner is touched
    untouched = [mk(nodes, 0) for nodes, _, _, _ in plan]
    trees_before = [p.tree for _, p in untouched]
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([dict(planner=p, x0=s.x0, sample_space=s.sample_space, seed=1) for s, p in untouched])
    assert [p.tree for _, p in untouched] == trees_before


def test_update_plans_holds_every_planner_to_its_time_budget():
    """with many jobs the wrap-up of finished planners must not be charged to the others' clocks.  16 planners, real clock,
    a 0.25 s budget each: every planner's own elapsed time at its exit decision is within one shared native call of the budget, and
    the whole call returns within the budget plus the (deferred) wrap-ups."""
    import time
    import lqrrt_amd
    s0 = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
    fleet = []
    for k in range(16):
        s = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
        cons = lqrrt_amd.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
        kw = dict(s.plan_kwargs)
        kw.update(error_tol=s.error_tol, erf=s.erf, min_time=0.25, max_time=0.25, max_nodes=60000, goal0=s.goal, sys_time=time.time,
                  printing=False, wave_size=256)
        fleet.append(lqrrt_amd.Planner(s.dynamics, s.lqr, cons, **kw))
    jobs = [dict(planner=p, x0=s0.x0, sample_space=s0.sample_space, goal_bias=s0.goal_bias, seed=40 + k) for k, p in enumerate(fleet)]
    lqrrt_amd.update_plans(jobs)                                    # warm: engines, kernels, sampler pools
    exits = []
    orig = lqrrt_amd.Planner._plan_after_call

    def spy(self, run, st, dt_call, wrap_up=True):
        over = orig(self, run, st, dt_call, wrap_up)
        if over:
            exits.append(run.time_elapsed)
        return over
    lqrrt_amd.Planner._plan_after_call = spy
    try:
        t0 = time.time()
        res = lqrrt_amd.update_plans(jobs)
        wall = time.time() - t0
    finally:
        lqrrt_amd.Planner._plan_after_call = orig
    assert len(res) == 16 and len(exits) == 16
    assert all(0.25 <= e < 0.25 + 0.03 for e in exits), exits       # the exit decision: at most one shared call past the budget
    # outside every planner's clock, inside this wall time: set-up (16 trees of the previous plans detached: bulk copies out of HBM)
    # and the 16 deferred wrap-ups (plan extraction, interpolators)
    assert wall < 0.25 + 0.5, wall
    for p in fleet:
        assert len(p.x_seq) == len(p.t_seq) and p.tree.size > 100
