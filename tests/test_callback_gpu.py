"""
Callback mode on the GPU: lqrrt.Planner handed PLAIN PYTHON FUNCTIONS -- the reference's plugin API (planner.py:35-59) -- must grow
the tree the reference's own Planner grows from the same functions and the same np.random stream.

The plugins are the NumPy functions of oracle/systems_np.py (restatements of the demo scripts' dynamics / lqr / erf / is_feasible;
here they only play "the caller's functions" -- the product never sees the oracle, it is handed four callables).  Expected values
are the committed fixtures generated from the reference itself (tools/gen_golden.py).

Exact: parent arrays and their hash, per-iteration nearest node, steer lengths, edge lengths, iteration count, the position of
np.random after the plan, node_seq, T.  Floating point (states, gains, plan) at 1e-9 absolute: the functions are NumPy and bit-equal
on the generating machine, another CPU's NumPy may differ in the last bit of arctan2 / tanh.
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-9

TRAJ = [("car", "500"), ("car", "2000"), ("boat_novice", "300"), ("boat_intermediate", "300"), ("pendulum", "150"),
        ("boat_advanced", "200"), ("car", "firstgoal"), ("boat_novice", "firstgoal"), ("car", "adaptive"),
        ("boat_intermediate", "adaptive"), ("car", "nopruning"), ("boat_novice", "nopruning"), ("car", "tries1"),
        ("boat_intermediate", "tries1"), ("car", "guide"), ("boat_intermediate", "guide")]


def _load(golden_dir, fname):
    path = os.path.join(golden_dir, fname)
    if not os.path.exists(path):
        pytest.fail("fixture %s missing" % fname)
    return np.load(path)


def _plain_functions(s):
    """The caller's plugins: plain functions (closures over the NumPy system object), nothing lqrrt_amd could recognise."""
    def dynamics(x, u, dt):
        return s.dynamics(x, u, dt)

    def lqr(x, u):
        return s.lqr(x, u)

    def erf(xgoal, x):
        return s.erf(xgoal, x)

    def is_feasible(x, u):
        return s.is_feasible(x, u)
    return dynamics, lqr, erf, is_feasible


def make_callback_planner(s, max_nodes, **over):
    import lqrrt
    dynamics, lqr, erf, is_feasible = _plain_functions(s)
    cons = lqrrt.Constraints(s.nstates, s.ncontrols, s.goal_buffer, is_feasible)
    kw = dict(s.plan_kwargs)
    kw.update(error_tol=s.error_tol, erf=erf, min_time=0, max_time=1, max_nodes=max_nodes, goal0=s.goal, printing=False,
              sys_time=lambda: 0.0)
    kw.update(over)
    return lqrrt.Planner(dynamics, lqr, cons, **kw)


@pytest.mark.parametrize("name,tag", TRAJ)
def test_reference_fixture_from_plain_python_plugins(golden_dir, name, tag):
    from systems_np import SYSTEMS
    g = _load(golden_dir, "traj_%s_%s.npz" % (name, tag))
    s = SYSTEMS[name](0)
    extra = dict(horizon=(0.1, 3)) if tag == "adaptive" else {}
    p = make_callback_planner(s, int(g["max_nodes"]), min_time=float(g["min_time"]), max_time=max(float(g["min_time"]), 1.0), **extra)
    assert p.callback_mode and p.system is None

    # trace the decisions through the planner's own hooks: a sampling function that wraps the default sampler is not possible without
    # changing the stream, so the nearest ids and steer lengths are read off the run object
    run_log = {"nearest": [], "steer_len": []}
    from lqrrt_amd import callback
    orig_nearest, orig_steer = callback.CallbackRun.nearest, callback.CallbackRun.steer

    def nearest(self, x, pruning):
        i = orig_nearest(self, x, pruning)
        run_log["nearest"].append(i)
        return i

    def steer(self, ID, xtar, force_arrive=False):
        xs, us = orig_steer(self, ID, xtar, force_arrive)
        if not force_arrive:
            run_log["steer_len"].append(len(xs))
        return xs, us
    callback.CallbackRun.nearest, callback.CallbackRun.steer = nearest, steer
    try:
        np.random.seed(1)
        pruning = bool(g["pruning"]) if "pruning" in g.files else True
        tries = int(g["tries"]) if "tries" in g.files else 10
        guide = g["guide"] if "guide" in g.files and len(g["guide"]) else None
        ret = p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=tries, pruning=pruning, guide=guide)
    finally:
        callback.CallbackRun.nearest, callback.CallbackRun.steer = orig_nearest, orig_steer

    assert ret == bool(g["returned"])
    assert p.stats["attempts"] == int(g["iterations"])
    # np.random stands where the reference's sampler left it: the next draw is the one after the candidates the fixture counted
    n = s.nstates
    want_next = np.random.RandomState(1).random_sample(int(g["n_candidates"]) * (n + 1) + 1)[-1]
    assert np.random.sample() == want_next
    np.testing.assert_array_equal(np.array(run_log["nearest"], dtype=np.int32), g["nearest"])
    np.testing.assert_array_equal(np.array(run_log["steer_len"], dtype=np.int16), g["steer_len"])
    np.testing.assert_array_equal(np.array(p.tree.pID, dtype=np.int32), g["pID"])
    assert hashlib.sha1(np.array(p.tree.pID, np.int64).tobytes()).hexdigest()[:16] == str(g["pid_hash"])
    np.testing.assert_allclose(p.tree.state, g["state"], rtol=0, atol=ATOL)
    np.testing.assert_array_equal(np.array([len(e) for e in p.tree.x_seq], dtype=np.int32), g["edge_len"])
    K = np.array([lk[1] for lk in p.tree.lqr], dtype=np.float64)
    np.testing.assert_allclose(K, g["K"], rtol=0, atol=ATOL)
    for t in "abc":
        ID = int(g["edge_%s_id" % t])
        np.testing.assert_allclose(np.array(p.tree.x_seq[ID]), g["edge_%s_x" % t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(np.array(p.tree.u_seq[ID]), g["edge_%s_u" % t], rtol=0, atol=1e-7)
    assert bool(p.plan_reached_goal) == bool(g["reached_goal"])
    np.testing.assert_array_equal(np.array(p.node_seq, dtype=np.int32), g["node_seq"])
    np.testing.assert_allclose(np.array(p.x_seq), g["plan_x"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(np.array(p.u_seq), g["plan_u"], rtol=0, atol=1e-7)
    assert p.T == float(g["plan_T"])
    if tag == "adaptive":
        assert p.horizon_iters == int(g["horizon_iters_final"])
    # the device's node table is the tree: states bit for bit what the host appended, parents, and the ignore set = the union of the
    # goal paths (planner.py:270)
    table = p._callback_run.table
    assert table.size == p.tree.size
    np.testing.assert_array_equal(table.states(), p.tree.state)
    np.testing.assert_array_equal(table.parents(), np.array(p.tree.pID, dtype=np.int32))
    # interpolators (planner.py:451-464)
    tq = 0.37 * p.T
    assert np.all(np.isfinite(p.get_state(tq))) and np.all(np.isfinite(p.get_effort(tq)))
    np.testing.assert_allclose(p.get_state(10 * p.T + 1), np.array(p.x_seq[-1]), rtol=0, atol=0)


def test_node_table_against_numpy():
    """The generic nearest-neighbour stage against NumPy's own expression of planner.py:340-350 (np.sum(np.tensordot(diffs, S, axes=1)
    * diffs, axis=1), argmin with lowest-id ties), for identity and dense S, with and without angular states and an ignore set."""
    from lqrrt_amd.engine import NodeTable
    rs = np.random.RandomState(5)
    for n, angles in ((1, ()), (4, (0, 1)), (5, (2,)), (6, (2,)), (7, ()), (9, (0, 4, 8)), (12, ()), (12, (3, 11)),
                      (13, ()), (14, (1, 13)), (20, (0, 5, 19)), (33, (2,)), (64, ())):        # beyond 12 states: run-time width
        N = 3000
        t = NodeTable(n, 2, angles, capacity=N + 64)
        nodes = rs.uniform(-6, 6, (N, n))
        nodes[7] = nodes[3]                                        # an exact tie: the lower id wins
        t.reset(nodes[0])
        for i in range(1, 40):
            t.append(int(rs.randint(0, i)), nodes[i])
        np.testing.assert_array_equal(t.states(), nodes[:40])
        pid = np.concatenate(([-1], [rs.randint(0, i) for i in range(1, N)])).astype(np.int32)
        t.load(nodes, pid)
        assert t.size == N
        np.testing.assert_array_equal(t.parents(), pid)
        ign = rs.rand(N) < 0.3
        ign[3] = False
        t.ignore(np.nonzero(ign)[0])
        np.testing.assert_array_equal(t.ignored(), ign)

        def errors(x):
            e = x - nodes
            for d in angles:
                e[:, d] = np.arctan2(np.sin(x[d]) * np.cos(nodes[:, d]) - np.cos(x[d]) * np.sin(nodes[:, d]),
                                     np.cos(x[d]) * np.cos(nodes[:, d]) + np.sin(x[d]) * np.sin(nodes[:, d]))
            return e
        A = rs.uniform(-1, 1, (n, n))
        Sd = A.dot(A.T) + n * np.eye(n)
        queries = list(rs.uniform(-6, 6, (24, n))) + [nodes[3].copy()]
        for S in (None, Sd):
            for x in queries:
                e = errors(x)
                Sm = np.eye(n) if S is None else S
                costs = np.sum(np.tensordot(e, Sm, axes=1) * e, axis=1)
                for use_ignore in (True, False):
                    masked = np.where(ign, np.inf, costs) if use_ignore else costs
                    want = int(np.argmin(masked))
                    got, c = t.nearest(x, S, use_ignore=use_ignore)
                    if got != want:                                   # only a last-bit difference of the atan2 may reorder two nodes
                        assert abs(costs[got] - costs[want]) <= 1e-9 * max(1.0, abs(costs[want])), (n, angles, got, want)
                    assert abs(c - costs[got]) <= 1e-9 * max(1.0, abs(costs[got]))
                    # the caller-evaluated form: same selection on the same rows, now exactly (no device atan2 involved)
                    got2, c2 = t.nearest_from_errors(e, S, use_ignore=use_ignore)
                    assert got2 == want or costs[got2] == costs[want]
            if n > 12:
                with pytest.raises(Exception):
                    t.nn_argmin(np.array(queries), S)                        # the batched device form serves up to 12 states
                continue
            ids, cs = t.nn_argmin(np.array(queries), S, use_ignore=True)
            for x, i, c in zip(queries, ids, cs):
                assert (int(i), c) == t.nearest(x, S, use_ignore=True)       # batched device form == host form, bit for bit
            full = t.costs_to_go(queries[0], S)
            e = errors(queries[0])
            np.testing.assert_allclose(full, np.sum(np.tensordot(e, np.eye(n) if S is None else S, axes=1) * e, axis=1), rtol=1e-12, atol=1e-12)
        # every node ignored: the overall nearest (planner.py:241,245)
        t.ignore(range(N))
        x = queries[1]
        e = errors(x)
        assert t.nearest(x, None, use_ignore=True)[0] == int(np.argmin(np.sum(e * e, axis=1)))
        t.close()


def test_exact_tie_takes_the_older_node():
    from lqrrt_amd.engine import NodeTable
    t = NodeTable(3, 1, (), capacity=1024)
    t.reset([0.0, 0.0, 0.0])
    for i in range(600):
        t.append(0, [1.0, 2.0, 3.0] if i % 2 else [-1.0, -2.0, -3.0])
    assert t.nearest([1.0, 2.0, 3.0])[0] == 2 and t.nearest([-1.0, -2.0, -3.0])[0] == 1
    t.ignore([1, 2])
    assert t.nearest([1.0, 2.0, 3.0])[0] == 4 and t.nearest([-1.0, -2.0, -3.0])[0] == 3
    t.truncate(3)
    assert t.size == 3
    t.close()


def test_unknown_erf_is_evaluated_on_the_host_and_selected_on_the_device(golden_dir):
    """An erf the probe cannot classify (here: the car's, with its angle error scaled -- not subtract-and-wrap) takes planner.py:588's
    path: one Python call per node, the rows are uploaded, the device contracts and selects.  Same tree as the NumPy oracle grows with
    the same functions."""
    from systems_np import SYSTEMS, make_oracle_planner
    from lqrrt_amd import callback
    s = SYSTEMS["car"](0)
    base = s.erf

    def odd_erf(xgoal, x):
        e = np.array(base(xgoal, x), dtype=np.float64)
        e[2] = 0.5 * e[2]
        return e
    assert callback.classify_erf(odd_erf, 5) is None
    s.erf = odd_erf
    s.batch_erf = None
    p = make_callback_planner(s, 200)
    np.random.seed(3)
    p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    o = make_oracle_planner(s, 200, vectorised_nn=False)
    np.random.seed(3)
    o.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    assert list(p.tree.pID) == list(o.tree.pID)
    np.testing.assert_array_equal(p.tree.state, o.tree.state)
    with pytest.raises(ValueError):
        make_callback_planner(s, 50, angle_dims=(2,)).update_plan(s.x0, s.sample_space)      # declared, and the probe disagrees


def test_callback_mode_control_surface():
    """Kill from another thread, real clock budget, finish_on_goal, a user sampling function, plugin swap between plans."""
    import threading
    import time
    from systems_np import SYSTEMS
    s = SYSTEMS["car"](0)
    p = make_callback_planner(s, 100000, min_time=5.0, max_time=5.0, sys_time=time.time)
    timer = threading.Timer(0.3, p.kill_update)
    t0 = time.time()
    timer.start()
    np.random.seed(1)
    assert p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias) is False         # planner.py:289,330-334
    assert 0.25 < time.time() - t0 < 1.5 and p.killed is False and p.tree.size > 5
    # budgeted plan: ends within one iteration of max_time
    p.set_runtime(min_time=0.2, max_time=0.4)
    t0 = time.time()
    np.random.seed(1)
    assert p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias) is True
    assert 0.19 < time.time() - t0 < 1.0
    assert len(p.x_seq) == len(p.u_seq) == len(p.t_seq) and p.node_seq[0] == 0
    # a sampling function of the planner sees the tree of THIS iteration (planner.py:236)
    seen = []
    own = np.random.RandomState(4)
    space = np.array(s.sample_space, dtype=np.float64)

    def sampler(planner):
        seen.append(planner.tree.size)
        return space[:, 0] + (space[:, 1] - space[:, 0]) * own.random_sample(5)
    p2 = make_callback_planner(s, 30)
    assert p2.update_plan(s.x0, s.sample_space, xrand_gen=sampler) is False
    assert seen[0] == 1 and seen == sorted(seen) and seen[-1] >= 2
    with pytest.raises(ValueError):
        p2.update_plan(s.x0, s.sample_space, xrand_gen="nope")
    with pytest.raises(ValueError):
        p2.update_plan(s.x0, [(0, 1)] * 4)
    # swapping to the native plugins (and back) between plans changes the mode, not the API
    import lqrrt
    car = lqrrt.systems.Car()
    p2.set_system(car.dynamics, car.lqr, lqrrt.Constraints(5, 2, car.goal_buffer, car.is_feasible), car.erf)
    assert not p2.callback_mode
    np.random.seed(1)
    p2.update_plan(car.x0, car.sample_space, goal_bias=car.goal_bias, xrand_gen=10)
    native_parents = list(p2.tree.pID)
    d, l, e, f = _plain_functions(s)
    p2.set_system(d, l, lqrrt.Constraints(5, 2, s.goal_buffer, f), e)
    assert p2.callback_mode
    np.random.seed(1)
    p2.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    assert list(p2.tree.pID) == native_parents                                        # one problem, two routes, one tree


def test_a_fourteen_state_problem_through_callback_mode():
    """More states than any compiled-in system has (LQRRT_MAX_STATES = 12): a 7-DoF double integrator with one angular joint, plain
    Python plugins.  lqrrt.Planner (node table with a run-time state dimension) against the NumPy oracle's planner driven by the same
    functions: the same tree, node for node."""
    from lqrrt_oracle import RefConstraints, RefPlanner
    dof = 7
    n, m = 2 * dof, dof
    Kgain = np.hstack((4.0 * np.eye(dof), 3.0 * np.eye(dof)))
    balls = np.array([[3.0, 3.0, 1.0], [6.0, 2.0, 1.2], [2.0, 7.0, 0.8]])

    def dynamics(x, u, dt):
        u = np.clip(u, -5, 5)
        return x + np.concatenate((x[dof:], u)) * dt

    def lqr(x, u):
        return np.eye(n), Kgain

    def erf(g, x):
        e = np.subtract(g, x)
        e[6] = np.arctan2(np.sin(e[6]), np.cos(e[6]))
        return e

    def is_feasible(x, u):
        return bool(np.all(np.hypot(balls[:, 0] - x[0], balls[:, 1] - x[1]) > balls[:, 2]))
    goal = np.concatenate(([8, 8, 1, -1, 0.5, 0, 2.5], np.zeros(dof)))
    buf = np.concatenate(([1.0, 1.0], np.full(n - 2, np.inf)))
    space = [(0, 10)] * 2 + [(-2, 2)] * 4 + [(-np.pi, np.pi)] + [(-1, 1)] * dof
    bias = [0.3, 0.3] + [0.0] * (n - 2)
    kw = dict(horizon=1.0, dt=0.1, FPR=0.5, error_tol=np.concatenate(([0.3, 0.3], np.full(n - 2, np.inf))), erf=erf, min_time=2, max_time=3,
              max_nodes=400, goal0=goal, printing=False, sys_time=lambda: 0.0)        # fake clock: ends by the node limit
    import lqrrt
    p = lqrrt.Planner(dynamics, lqr, lqrrt.Constraints(n, m, buf, is_feasible), **kw)
    o = RefPlanner(dynamics, lqr, RefConstraints(n, m, buf, is_feasible), **kw)
    np.random.seed(8)
    rp = p.update_plan(np.zeros(n), space, goal_bias=bias, xrand_gen=10)
    assert p._erf_angles == (6,) and p._callback_run.table.n == 14
    np.random.seed(8)
    ro = o.update_plan(np.zeros(n), space, goal_bias=bias, xrand_gen=10)
    assert rp is False and ro is False and p.tree.size == o.tree.size == 401
    assert list(p.tree.pID) == list(o.tree.pID)
    np.testing.assert_array_equal(p.tree.state, np.array(o.tree.state))
    assert p.plan_reached_goal == o.plan_reached_goal and list(p.node_seq) == list(o.node_seq)


@pytest.mark.parametrize("name", ["boat", "car", "escape"])
def test_ros_behaviour_fixture_from_plain_python_plugins(golden_dir, name):
    """The ROS package's three behaviours (demos/lqrrt_ros/behaviors/*.py) as plain Python plugins: adaptive horizon (planner.py:418-425),
    the node's occupancy-grid feasibility (lqrrt_node.py:719-745) and, for 'car', a cost-to-go matrix that is NOT the identity
    (S = diag(1,1,1,0,0,0), car.py:65) -- the dense form of the device's nearest-neighbour kernel against the reference's own decisions."""
    from systems_np import RosBoat
    g = _load(golden_dir, "ros_%s.npz" % name)
    rs = RosBoat(name)
    rs.set_occupancy_grid(g["grid"], g["origin"], float(g["cpm"]), float(g["threshold"]))
    rs.goal = [float(v) for v in g["goal"]]
    rs.sample_space = [tuple(r) for r in g["sample_space"]]
    p = make_callback_planner(rs, 300, min_time=2, max_time=3)
    assert p.callback_mode
    log = []
    from lqrrt_amd import callback
    orig = callback.CallbackRun.nearest

    def nearest(self, x, pruning):
        i = orig(self, x, pruning)
        log.append(i)
        return i
    callback.CallbackRun.nearest = nearest
    try:
        np.random.seed(1)
        ret = p.update_plan(rs.x0, rs.sample_space, goal_bias=rs.goal_bias, xrand_gen=10)
    finally:
        callback.CallbackRun.nearest = orig
    assert ret == bool(g["returned"]) and p.stats["attempts"] == int(g["iterations"])
    np.testing.assert_array_equal(np.array(log, dtype=np.int32), g["nearest"])
    np.testing.assert_array_equal(np.array(p.tree.pID, dtype=np.int32), g["pID"])
    assert hashlib.sha1(np.array(p.tree.pID, np.int64).tobytes()).hexdigest()[:16] == str(g["pid_hash"])
    np.testing.assert_array_equal(np.array([len(e) for e in p.tree.x_seq], dtype=np.int32), g["edge_len"])
    err = np.abs(p.tree.state - g["state"]).max(axis=1)
    assert np.median(err) < 1e-12 and (err.max() < ATOL or name == "car")       # (car: the reference's own sensitivity, tests/test_ros_behaviors.py)
    assert p.horizon_iters == int(g["horizon_iters_final"])
    assert bool(p.plan_reached_goal) == bool(g["reached_goal"])
    np.testing.assert_array_equal(np.array(p.node_seq, dtype=np.int32), g["node_seq"])
    if name == "car":
        S = np.asarray(rs.lqr(rs.x0, np.zeros(3))[0], dtype=np.float64)
        assert not np.array_equal(S, np.eye(6))                                    # the dense-S kernel was the one that ran
