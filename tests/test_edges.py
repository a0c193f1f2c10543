"""
Edge interiors and the plan's interpolators against the reference (missing #5): until round 3 a trajectory fixture
held three complete edges per run; every other x_seq / u_seq row was pinned to the reference through its edge's length and
end state only.  tests/golden/edges_<run>.npz (tools/gen_golden.py --job edges_car2000 / edges_adv3000: the SAME runs as the
traj_* fixtures, asserted at generation) hold every row of tree.x_seq / tree.u_seq (tree.py:121-132) and, where the run has a
plan, get_state / get_effort (planner.py:451-464) at 64 times including both out-of-range sides.

  car, 2,001 nodes       free-running (the whole tree is reproduced): all 97,867 rows of all edges + the interpolators
  boat_advanced, 3,001   teacher-forced (chaotic free run): every accepted decision's edge, row by row, steered
                         from the reference's own node
Tolerances: states 1e-9, efforts 1e-6 (they are O(1e3)).
"""
import os

import numpy as np
import pytest

import teacher

X_ATOL, U_ATOL = 1e-9, 1e-6


def _load(golden_dir, fname):
    path = os.path.join(golden_dir, fname)
    if not os.path.exists(path):
        pytest.fail("fixture %s missing: tests/golden is committed, a lost fixture must not turn into a pass" % fname)
    return np.load(path)


def _edges(golden_dir, name, tag):
    g = _load(golden_dir, "traj_%s_%s.npz" % (name, tag))
    e = _load(golden_dir, "edges_%s_%s.npz" % (name, tag))
    assert str(e["pid_hash"]) == str(g["pid_hash"]) and np.array_equal(e["edge_len"], g["edge_len"])
    off = np.concatenate(([0], np.cumsum(e["edge_len"])))
    return g, e, off


def _system(name):
    import lqrrt_amd
    return lqrrt_amd.systems.SYSTEMS[name](0)


def _check_free_run(edge_of, size, e, off):
    worst_x = worst_u = 0.0
    for ID in range(size):
        x, u = edge_of(ID)
        rx, ru = e["x_cat"][off[ID]:off[ID + 1]], e["u_cat"][off[ID]:off[ID + 1]]
        assert len(x) == len(rx), ID
        worst_x = max(worst_x, float(np.abs(np.asarray(x) - rx).max()))
        worst_u = max(worst_u, float(np.abs(np.asarray(u) - ru).max()))
    return worst_x, worst_u


def test_c_oracle_every_edge_row_car_2000(golden_dir):
    import coracle
    g, e, off = _edges(golden_dir, "car", "2000")
    o = coracle.make(_system("car"), int(g["max_nodes"]), seed=1)
    o.extend(max_nodes=int(g["max_nodes"]))
    np.testing.assert_array_equal(o.parents(), g["pID"])
    wx, wu = _check_free_run(o.edge, o.size, e, off)
    print("car 2000, C oracle: %d rows, worst |dx| %.2e, worst |du| %.2e" % (len(e["x_cat"]), wx, wu))
    assert wx < X_ATOL and wu < U_ATOL


def test_c_oracle_edge_rows_boat_advanced_3000_teacher_forced(golden_dir):
    import coracle
    g, e, off = _edges(golden_dir, "boat_advanced", "3000")
    s = _system("boat_advanced")
    sch = teacher.Schedule(teacher.load_fixture(os.path.join(golden_dir, "traj_boat_advanced_3000.npz")), s.goal, s.goal_buffer)
    o = coracle.make(s, len(sch.state) + 8, seed=1)
    o.load_tree(sch.state, sch.K, sch.pID)
    wx = wu = 0.0
    rows = 0
    acc = np.flatnonzero(sch.steer_len > 0)[::4]            # (every 4th edge here to keep the CPU suite short; the GPU test takes them all)
    for t in acc:
        ID = int(sch.new_node[t])
        k, xs, us, _ = o.steer_from(sch.nearest[t], sch.xrand[t])
        assert k == off[ID + 1] - off[ID]
        wx = max(wx, float(np.abs(xs[:k] - e["x_cat"][off[ID]:off[ID + 1]]).max()))
        wu = max(wu, float(np.abs(us[:k] - e["u_cat"][off[ID]:off[ID + 1]]).max()))
        rows += k
    print("boat_advanced 3000, C oracle, teacher-forced: %d rows, worst |dx| %.2e, worst |du| %.2e" % (rows, wx, wu))
    assert rows == int(sum(e["edge_len"][sch.new_node[t]] for t in acc))
    assert wx < X_ATOL and wu < U_ATOL


@pytest.mark.gpu
def test_hip_every_edge_row_and_the_interpolators_car_2000(golden_dir):
    import lqrrt_amd as lqrrt
    g, e, off = _edges(golden_dir, "car", "2000")
    s = _system("car")
    cons = lqrrt.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
    p = lqrrt.Planner(s.dynamics, s.lqr, cons, error_tol=s.error_tol, erf=s.erf, min_time=float(g["min_time"]),
                      max_time=float(g["min_time"]) + 1, max_nodes=int(g["max_nodes"]), goal0=s.goal, sys_time=lambda: 0.0,
                      printing=False, wave_size=1024, **s.plan_kwargs)
    np.random.seed(1)
    p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    np.testing.assert_array_equal(np.array(p.tree.pID, dtype=np.int32), g["pID"])
    xe, ue, le = p._engine.edges()
    np.testing.assert_array_equal(le[1:], e["edge_len"][1:])
    wx = wu = 0.0
    for ID in range(1, p.tree.size):
        wx = max(wx, float(np.abs(xe[ID, :le[ID]] - e["x_cat"][off[ID]:off[ID + 1]]).max()))
        wu = max(wu, float(np.abs(ue[ID, :le[ID]] - e["u_cat"][off[ID]:off[ID + 1]]).max()))
    assert wx < X_ATOL and wu < U_ATOL
    # the drop-in Tree's own accessors (tree.x_seq[i] / u_seq[i], tree.py:121-132) on a few nodes, the root included
    for ID in (0, 1, 777, p.tree.size - 1):
        np.testing.assert_allclose(np.asarray(p.tree.x_seq[ID]), e["x_cat"][off[ID]:off[ID + 1]], rtol=0, atol=X_ATOL)
        np.testing.assert_allclose(np.asarray(p.tree.u_seq[ID]), e["u_cat"][off[ID]:off[ID + 1]], rtol=0, atol=U_ATOL)
    # interpolators (planner.py:451-464): inside the plan, before its start, beyond its end
    assert abs(p.T - float(g["plan_T"])) < 1e-12
    for t, rx, ru in zip(e["interp_t"], e["interp_x"], e["interp_u"]):
        np.testing.assert_allclose(p.get_state(float(t)), rx, rtol=0, atol=X_ATOL)
        np.testing.assert_allclose(p.get_effort(float(t)), ru, rtol=0, atol=U_ATOL)


@pytest.mark.gpu
def test_hip_every_edge_row_boat_advanced_3000_teacher_forced(golden_dir):
    from lqrrt_amd.engine import Engine
    g, e, off = _edges(golden_dir, "boat_advanced", "3000")
    s = _system("boat_advanced")
    sch = teacher.Schedule(teacher.load_fixture(os.path.join(golden_dir, "traj_boat_advanced_3000.npz")), s.goal, s.goal_buffer)
    kw = s.plan_kwargs
    eng = Engine(s, capacity=len(sch.state) + 8, max_wave=1024)
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    eng.tree_load(sch.state, sch.K, sch.pID)
    acc = np.flatnonzero(sch.steer_len > 0)
    wx = wu = 0.0
    rows = 0
    for a in range(0, len(acc), 1024):
        ts = acc[a:a + 1024]
        ln, xs, us, _, _ = eng.steer_batch(sch.nearest[ts], sch.xrand[ts])
        for j, t in enumerate(ts):
            ID = int(sch.new_node[t])
            k = int(ln[j])
            assert k == off[ID + 1] - off[ID]
            wx = max(wx, float(np.abs(xs[j, :k] - e["x_cat"][off[ID]:off[ID + 1]]).max()))
            wu = max(wu, float(np.abs(us[j, :k] - e["u_cat"][off[ID]:off[ID + 1]]).max()))
            rows += k
    eng.close()
    print("boat_advanced 3000, HIP, teacher-forced: %d rows, worst |dx| %.2e, worst |du| %.2e" % (rows, wx, wu))
    assert rows == len(e["x_cat"]) - int(e["edge_len"][0])
    assert wx < X_ATOL and wu < U_ATOL
