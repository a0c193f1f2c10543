"""
BASELINE.json config 5 (synthetic 12-DoF double integrator, box obstacles) pinned to the REFERENCE.

The system is not in the reference; SURVEY 8(d) names its oracle: the reference's own Planner driven by the build's
NumPy callbacks (oracle/systems_np.DoubleIntegrator, S and K from scipy.linalg.solve_discrete_are).
tools/gen_golden.py --job di600 / di2500 ran exactly that and wrote tests/golden/traj_double_integrator_*.npz
(3000 boxes from RandomState(0), plan seed 1, teacher data included).

CPU: the NumPy oracle, the C oracle (free-running and teacher-forced) against those fixtures.
GPU: the HIP path free-running through the drop-in Planner and teacher-forced through the C ABI; then the full-size
configuration (100 000 boxes, 50 000 nodes) through size-independent properties.

Tolerances: parents, nearest ids, edge lengths, iteration / sampler-row counts exact; states 1e-9, efforts 1e-6.
(The product computes S, K with its own doubling DARE; the fixture's are SciPy's: they agree to 1e-12.)
"""
import os

import numpy as np
import pytest

import teacher

N_BOXES = 3000
TAGS = ["600", "2500"]


def _fx(golden_dir, tag):
    path = os.path.join(golden_dir, "traj_double_integrator_%s.npz" % tag)
    if not os.path.exists(path):
        pytest.fail("fixture missing: tests/golden is committed, a lost fixture must not turn into a pass")
    return np.load(path)


def _native_system(g):
    import lqrrt_amd
    s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=int(g["n_boxes"]), seed=int(g["box_seed"]))
    assert float(s.obs[:, :3].sum()) == float(g["box_lo_sum"]) and float(s.obs[:, 3:].sum()) == float(g["box_hi_sum"])
    return s


def test_riccati_solution_matches_scipy_fixture(golden_dir):
    g = _fx(golden_dir, "600")
    s = _native_system(g)
    np.testing.assert_allclose(s.S, g["S"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(s.K, g["Kconst"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag", TAGS)
def test_numpy_oracle_vs_reference_run(golden_dir, tag):
    from systems_np import DoubleIntegrator, make_oracle_planner
    g = _fx(golden_dir, tag)
    s = DoubleIntegrator(n_boxes=int(g["n_boxes"]), seed=int(g["box_seed"]))
    p = make_oracle_planner(s, int(g["max_nodes"]), min_time=2, max_time=3)
    np.random.seed(1)
    ret = p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10, trace=True)
    assert ret == bool(g["returned"]) and p.iterations == int(g["iterations"])
    np.testing.assert_array_equal(np.array(p.tree.pID, dtype=np.int32), g["pID"])
    np.testing.assert_array_equal(np.array(p.trace["nearest"], dtype=np.int32), g["nearest"])
    np.testing.assert_array_equal(np.array(p.trace["steer_len"], dtype=np.int16), g["steer_len"])
    np.testing.assert_allclose(p.tree.state, g["state"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(np.array(p.node_seq, dtype=np.int32), g["node_seq"])
    np.testing.assert_allclose(np.array(p.x_seq), g["plan_x"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag", TAGS)
def test_c_oracle_vs_reference_run(golden_dir, tag):
    import coracle
    g = _fx(golden_dir, tag)
    s = _native_system(g)
    o = coracle.make(s, int(g["max_nodes"]), seed=1)
    o.enable_trace(int(g["iterations"]) + 16)
    assert o.extend(max_nodes=int(g["max_nodes"])) == 2
    assert o.iterations == int(g["iterations"]) and o.candidates == int(g["n_candidates"])
    np.testing.assert_array_equal(o.parents(), g["pID"])
    near, ln = o.trace()
    np.testing.assert_array_equal(near, g["nearest"])
    np.testing.assert_array_equal(ln, g["steer_len"].astype(np.int32))
    np.testing.assert_array_equal(o.edge_lengths(), g["edge_len"])
    np.testing.assert_allclose(o.states(), g["state"], rtol=0, atol=1e-9)
    for t in "abc":
        x, u = o.edge(int(g["edge_%s_id" % t]))
        np.testing.assert_allclose(x, g["edge_%s_x" % t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(u, g["edge_%s_u" % t], rtol=0, atol=1e-6)
    # teacher-forced: every decision of the reference's run from the reference's tree
    sch = teacher.Schedule(g, s.goal, s.goal_buffer)
    o.load_tree(sch.state, sch.K, sch.pID)
    cur = None
    for size, a, b in sch.groups():
        ign = sch.ignored_at(size)
        if ign is not cur:
            o.set_ignored(ign)
            cur = ign
        for t in range(a, b):
            assert o.nearest_prefix(sch.xrand[t], size) == sch.nearest[t]
            k, xs, _, _ = o.steer_from(sch.nearest[t], sch.xrand[t])
            assert k == sch.steer_len[t]
            if k:
                assert np.abs(xs[-1] - sch.state[sch.new_node[t]]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("tag,wave", [("600", 128), ("2500", 1024)])
def test_hip_vs_reference_run(golden_dir, tag, wave):
    import lqrrt_amd as lqrrt
    from test_teacher_gpu import replay_hip
    g = _fx(golden_dir, tag)
    s = _native_system(g)
    cons = lqrrt.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
    p = lqrrt.Planner(s.dynamics, s.lqr, cons, error_tol=s.error_tol, erf=s.erf, min_time=2, max_time=3,
                      max_nodes=int(g["max_nodes"]), goal0=s.goal, sys_time=lambda: 0.0, printing=False, wave_size=wave,
                      **s.plan_kwargs)
    np.random.seed(1)
    ret = p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    assert ret == bool(g["returned"])
    assert p.stats["attempts"] == int(g["iterations"]) and p.stats["candidates"] == int(g["n_candidates"])
    np.testing.assert_array_equal(np.array(p.tree.pID, dtype=np.int32), g["pID"])
    np.testing.assert_array_equal(p._engine.edge_lengths(), g["edge_len"])
    np.testing.assert_allclose(p.tree.state, g["state"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(p._engine.gains(), g["K"], rtol=0, atol=1e-9)
    assert bool(p.plan_reached_goal) == bool(g["reached_goal"])
    np.testing.assert_array_equal(np.array(p.node_seq, dtype=np.int32), g["node_seq"])
    np.testing.assert_allclose(np.array(p.x_seq), g["plan_x"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(p.u_seq), g["plan_u"], rtol=0, atol=1e-6)
    assert abs(p.T - float(g["plan_T"])) < 1e-12
    for t in "abc":
        ID = int(g["edge_%s_id" % t])
        np.testing.assert_allclose(np.array(p.tree.x_seq[ID]), g["edge_%s_x" % t], rtol=0, atol=1e-9)
    # teacher-forced, through lqrrt_tree_load / lqrrt_nn_argmin / lqrrt_steer_batch
    sch = teacher.Schedule(g, s.goal, s.goal_buffer)
    kw = s.plan_kwargs
    r = replay_hip(s, sch, kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), wave=256)
    print(r)
    assert r["nearest_miss"] == 0 and r["steer_len_mismatch"] == 0 and r["end_state_max_err"] < 1e-9
    assert r["end_state_compared"] == len(sch.state) - 1


@pytest.mark.gpu
def test_full_size_config5_invariants():
    """100 000 boxes, tree grown to 50 000 nodes: the reference's data-structure invariants at BASELINE's full size."""
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    nodes, wave = 50000, 1024
    s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=100000, seed=0)
    eng = Engine(s, capacity=nodes + wave + 8, max_wave=wave)
    kw = s.plan_kwargs
    H = int(kw["horizon"] / kw["dt"])
    eng.set_resolution(kw["dt"], kw["FPR"], H, np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st0 = np.random.RandomState(1).get_state()
    eng.set_mt19937(st0[1], st0[2])
    eng.tree_reset(s.x0)
    stats = eng.extend(wave, node_limit=nodes - 1)
    N = eng.size
    assert N == nodes and stats.accepted == N - 1 and stats.attempts >= N - 1
    pid, elen, st = eng.parents(), eng.edge_lengths(), eng.states()
    assert pid[0] == -1 and np.all(pid[1:] >= 0) and np.all(pid[1:] < np.arange(1, N))
    assert elen[0] == 1 and np.all(elen[1:] >= 1) and np.all(elen[1:] <= H)
    # every node is the end of its edge and starts at its parent: re-simulate sampled edges with the dynamics operator
    ids = np.r_[1:40, N // 2:N // 2 + 40, N - 40:N]
    xe, ue, ln = eng.edges()
    for ID in ids:
        k = int(ln[ID])
        np.testing.assert_array_equal(xe[ID, k - 1], st[ID])
        prev = np.vstack((st[pid[ID]][None, :], xe[ID, :k - 1]))
        np.testing.assert_array_equal(eng.dynamics_batch(prev, ue[ID, :k]), xe[ID, :k])     # x_{i+1} = f(x_i, u_i)
    rec = np.vstack([xe[ID, :ln[ID]] for ID in ids])
    assert eng.feasible_batch(rec).all()                             # no recorded state lies in a box
    lo, hi = s.obs[:, :3], s.obs[:, 3:]
    inside = np.any(np.all((rec[:, None, :3] >= lo[None]) & (rec[:, None, :3] <= hi[None]), axis=2), axis=1)
    assert not inside.any()                                          # ... by brute force over all 100k boxes, too
    np.testing.assert_array_equal(eng.gains()[ids], np.broadcast_to(s.K, (len(ids),) + s.K.shape))
    # nearest-neighbour kernels == arg-min of the full cost vector (dense S, banded fast path), with the ignore set
    rng = np.random.RandomState(4)
    q = space[:, 0] + (space[:, 1] - space[:, 0]) * rng.random_sample((64, s.nstates))
    ids_all, cost_all = eng.nn_argmin(q, use_ignore=False)
    ids_ign, _ = eng.nn_argmin(q, use_ignore=True)
    ign = eng.ignored()
    for k in range(0, 64, 8):
        c = eng.costs_to_go(q[k])
        d = q[k][None, :] - st
        want = np.sum(d.dot(s.S) * d, axis=1)
        np.testing.assert_allclose(c, want, rtol=1e-11)
        assert int(ids_all[k]) == int(np.argmin(c)) and cost_all[k] == c.min()
        assert int(ids_ign[k]) == int(np.argmin(np.where(ign, np.inf, c)))
    # ignore set = union of the root paths of the goal hits (planner.py:270)
    g_lo = np.array(s.goal) - np.array(s.goal_buffer)
    g_hi = np.array(s.goal) + np.array(s.goal_buffer)
    in_goal = np.all((st > g_lo) & (st < g_hi), axis=1)
    in_goal[0] = False
    want = np.zeros(N, dtype=bool)
    for ID in np.flatnonzero(in_goal):
        v = int(ID)
        while v != -1 and not want[v]:
            want[v] = True
            v = int(pid[v])
    np.testing.assert_array_equal(ign, want)
    end, steps, hits = eng.plan_best()
    assert hits == int(in_goal.sum()) > 0 and in_goal[end]


def _inside_any_box(P, lo, hi, order=None):
    """Closed-box membership of every point of P (B, 3) in 100k boxes, in NumPy: the boxes' half-extents are below 0.5, so a box that
    holds p has lo_x in [p_x - 1.001, p_x] -- the candidates of a point are a contiguous run of the boxes sorted by lo_x, and those are
    tested with the plain comparisons.  `order=None` tests every box for every point (the brute force proper; subsets only)."""
    if order is None:
        out = np.zeros(len(P), dtype=bool)
        for a in range(0, len(P), 256):
            p = P[a:a + 256, None, :]
            out[a:a + 256] = np.any(np.all((p >= lo[None]) & (p <= hi[None]), axis=2), axis=1)
        return out
    slo, shi = lo[order], hi[order]
    first = np.searchsorted(slo[:, 0], P[:, 0] - 1.001, side="left")
    last = np.searchsorted(slo[:, 0], P[:, 0], side="right")
    out = np.zeros(len(P), dtype=bool)
    width = int((last - first).max())
    for a in range(0, len(P), 2048):
        f = first[a:a + 2048]
        idx = np.minimum(f[:, None] + np.arange(width)[None, :], len(slo) - 1)
        live = (f[:, None] + np.arange(width)[None, :]) < last[a:a + 2048, None]
        p = P[a:a + 2048, None, :]
        out[a:a + 2048] = np.any(live & np.all((p >= slo[idx]) & (p <= shi[idx]), axis=2), axis=1)
    return out


@pytest.mark.gpu
def test_feasible_batch_full_obstacle_count_vs_numpy():
    """is_feasible of config 5 at its full obstacle count: 100 000 states, most of them ON or one ulp beside a face of one of the 100 000
    boxes, against NumPy (exact booleans).  The device looks boxes up through its CSR grid; NumPy tests the boxes themselves."""
    import lqrrt_amd
    s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=100000, seed=0)
    lo, hi = np.ascontiguousarray(s.obs[:, :3]), np.ascontiguousarray(s.obs[:, 3:])
    assert float((hi - lo).max()) < 1.0
    rng = np.random.RandomState(21)
    B = 100000
    X = np.zeros((B, 12))
    X[:, 3:] = rng.uniform(-2, 2, (B, 9))
    X[:, :3] = rng.uniform(-1.0, 101.0, (B, 3))                     # a slab of free-flying points, some outside the boxes' volume
    pick = rng.randint(0, len(lo), B)
    dim = rng.randint(0, 3, B)
    side = rng.randint(0, 2, B)
    kind = rng.randint(0, 4, B)                                     # 0: leave uniform | 1: on the face | 2: one ulp outside | 3: one ulp inside
    inside_pt = lo[pick] + (hi[pick] - lo[pick]) * rng.random_sample((B, 3))
    face = np.where(side == 0, lo[pick, dim], hi[pick, dim])
    outward = np.where(side == 0, -np.inf, np.inf)
    coord = np.where(kind == 1, face, np.where(kind == 2, np.nextafter(face, outward), np.nextafter(face, -outward)))
    rows = np.nonzero(kind > 0)[0]
    X[rows, :3] = inside_pt[rows]
    X[rows, dim[rows]] = coord[rows]
    order = np.argsort(lo[:, 0], kind="stable")
    want_inside = _inside_any_box(X[:, :3], lo, hi, order)
    sub = rng.choice(B, 3000, replace=False)
    np.testing.assert_array_equal(want_inside[sub], _inside_any_box(X[sub, :3], lo, hi))      # the slab search == every box, on a subset
    assert 0.3 < want_inside.mean() < 0.7                           # both answers are well represented
    eng = s._engine()
    got = np.concatenate([eng.feasible_batch(X[a:a + 25000]) for a in range(0, B, 25000)])
    np.testing.assert_array_equal(got, ~want_inside)
    # on-face points are inside (closed boxes), their outward neighbours are outside unless another box holds them
    on_face = rows[kind[rows] == 1]
    assert not got[on_face].any()
