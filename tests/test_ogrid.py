"""
Occupancy-grid feasibility (demos/lqrrt_ros/nodes/lqrrt_node.py:719-745): the fixture
tests/golden/ops_ogrid.npz was produced by the reference's own method text (tools/gen_golden.py:gen_ogrid),
768 poses around and beyond a 300 x 400 grid with the ROS package's 1512-point hull lattice.
"""
import os

import numpy as np
import pytest


def _fixture(golden_dir):
    path = os.path.join(golden_dir, "ops_ogrid.npz")
    if not os.path.exists(path):
        pytest.fail("fixture missing: tests/golden is committed, a lost fixture must not turn into a pass")
    return np.load(path)


def _native_boat(g):
    import lqrrt_amd
    s = lqrrt_amd.systems.BoatAdvanced(0)
    # park the planning speed box far away so that the grid alone decides (the ROS node has no speed box)
    s.velmax_pos_plan = np.full(3, 1e9)
    s.velmax_neg_plan = np.full(3, -1e9)
    s.set_occupancy_grid(g["grid"], g["origin"], cpm=float(g["cpm"]), threshold=float(g["threshold"]), vps=g["vps"])
    return s


def test_oracles_match_reference_ogrid(golden_dir):
    import coracle
    from systems_np import SYSTEMS
    g = _fixture(golden_dir)
    rs = SYSTEMS["boat_advanced"](0)
    rs.velmax_pos_plan, rs.velmax_neg_plan = np.full(3, 1e9), np.full(3, -1e9)
    rs.set_occupancy_grid(g["grid"], g["origin"], float(g["cpm"]), float(g["threshold"]), vps=g["vps"])
    ok = np.array([rs.is_feasible(np.copy(x), np.zeros(3)) for x in g["x"]])
    np.testing.assert_array_equal(ok, g["ok"])
    o = coracle.make(_native_boat(g), 16)
    okc = np.array([o.feasible(x, np.zeros(3)) for x in g["x"]])
    np.testing.assert_array_equal(okc, g["ok"])
    assert 0.2 < g["ok"].mean() < 0.8


def test_ogrid_argument_checks():
    import lqrrt_amd
    b = lqrrt_amd.systems.BoatNovice(0)
    with pytest.raises(ValueError):
        b.set_occupancy_grid(np.zeros((4, 4)), (0, 0), resolution=0.5)          # centre-point model has no hull
    a = lqrrt_amd.systems.BoatAdvanced(0)
    with pytest.raises(ValueError):
        a.set_occupancy_grid(np.zeros((4, 4)), (0, 0))                          # resolution xor cpm
    with pytest.raises(ValueError):
        a.set_occupancy_grid(np.zeros(16), (0, 0), resolution=0.5)


@pytest.mark.gpu
def test_hip_ogrid_feasibility_golden(golden_dir):
    g = _fixture(golden_dir)
    s = _native_boat(g)
    ok = s._engine(0.1).feasible_batch(g["x"])
    np.testing.assert_array_equal(ok, g["ok"])


@pytest.mark.gpu
def test_hip_ogrid_tree_bit_exact_vs_coracle(golden_dir):
    """Plan through the occupancy grid: engine and sequential C oracle grow the same tree."""
    import coracle
    from lqrrt_amd.engine import Engine
    g = _fixture(golden_dir)
    import lqrrt_amd
    s = lqrrt_amd.systems.BoatIntermediate(0)
    grid = np.array(g["grid"])
    cpm, origin = float(g["cpm"]), g["origin"]
    for px, py in ((s.x0[0], s.x0[1]), (s.goal[0], s.goal[1])):          # keep start and goal areas free
        c, r = int(cpm * (px - origin[0])), int(cpm * (py - origin[1]))
        grid[max(r - 40, 0):r + 40, max(c - 40, 0):c + 40] = 0
    s.set_occupancy_grid(grid, origin, cpm=cpm, threshold=float(g["threshold"]))
    nodes, wave, budget = 1500, 256, 12000
    eng = Engine(s, capacity=nodes + wave + 8, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(3).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    stats = eng.extend(wave, max_attempts=budget, node_limit=nodes)       # bounded: a cluttered map may saturate
    o = coracle.make(s, nodes + wave + 8, seed=3)
    o.extend(max_iters=budget, max_nodes=nodes)
    assert eng.size == o.size and eng.size > 300
    assert stats.attempts == o.iterations and stats.candidates == o.candidates
    np.testing.assert_array_equal(eng.parents(), o.parents())
    np.testing.assert_array_equal(eng.states(), o.states())
    np.testing.assert_array_equal(eng.edge_lengths(), o.edge_lengths())
    assert (eng.edge_lengths() < 20).mean() > 0.05          # the grid actually cut edges


@pytest.mark.gpu
def test_map_swap_between_plans_matches_fresh_oracle(golden_dir):
    """A new map arrives between two plans (lqrrt_node.py:65, 719-745): the existing engine must plan through
    the NEW map exactly like the sequential oracle built on it, and the plugin handle must see it too."""
    import coracle
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    g = _fixture(golden_dir)
    s = lqrrt_amd.systems.BoatIntermediate(0)
    cpm, origin = float(g["cpm"]), g["origin"]

    def cleared(grid):
        grid = np.array(grid)
        for px, py in ((s.x0[0], s.x0[1]), (s.goal[0], s.goal[1])):
            c, r = int(cpm * (px - origin[0])), int(cpm * (py - origin[1]))
            grid[max(r - 40, 0):r + 40, max(c - 40, 0):c + 40] = 0
        return grid

    map_a = cleared(g["grid"])
    map_b = cleared(np.roll(np.array(g["grid"]), 37, axis=1)[::-1])       # a different world
    s.set_occupancy_grid(map_a, origin, cpm=cpm, threshold=float(g["threshold"]))
    nodes, wave, budget = 600, 128, 6000
    eng = Engine(s, capacity=nodes + wave + 8, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(5).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    eng.extend(wave, max_attempts=budget // 2, node_limit=nodes)            # first plan, map A (leaves queued samples behind)
    probe = g["x"][:256]
    ok_a = s._engine(0.1).feasible_batch(probe)

    s.set_occupancy_grid(map_b, origin, cpm=cpm, threshold=float(g["threshold"]))
    assert eng.sync_geometry() and not eng.sync_geometry()
    ok_b = s._engine(0.1).feasible_batch(probe)
    assert (ok_a != ok_b).any()
    key, pos = eng.get_mt19937()                                             # the stream continues where plan 1 stopped
    eng.tree_reset(s.x0)
    stats = eng.extend(wave, max_attempts=budget, node_limit=nodes)

    o = coracle.make(s, nodes + wave + 8, seed=5)                            # fresh oracle on map B ...
    o.set_mt19937(key, pos)                                                  # ... fed the same continued stream
    o.extend(max_iters=budget, max_nodes=nodes)
    assert eng.size == o.size and eng.size > 100
    np.testing.assert_array_equal(eng.parents(), o.parents())
    np.testing.assert_array_equal(eng.states(), o.states())
    assert stats.attempts == o.iterations
    okc = np.array([o.feasible(x, np.zeros(3)) for x in probe])
    np.testing.assert_array_equal(ok_b, okc)


@pytest.mark.gpu
def test_first_infeasible_plan_reevaluation(golden_dir):
    """Constraints.first_infeasible = the node's walk along the current plan (lqrrt_node.py:806-824)."""
    import lqrrt_amd
    g = _fixture(golden_dir)
    s = _native_boat(g)
    c = lqrrt_amd.Constraints(6, 3, s.goal_buffer, s.is_feasible)
    X = np.array(g["x"][:400])
    ok = np.array(g["ok"][:400], dtype=bool)
    want = int(np.nonzero(~ok)[0][0]) if (~ok).any() else -1
    assert c.first_infeasible(X) == want
    np.testing.assert_array_equal(c.feasible_batch(X), ok)
    free = X[ok]
    assert c.first_infeasible(free) == -1
    assert c.first_infeasible(free[:0].reshape(0, 6)) == -1


@pytest.mark.gpu
def test_tree_chain_example_runs():
    """examples/tree_chain_gpu.py: chained update_plan(specific_time=...) with maps changing in between
    (lqrrt_node.py:389-500).  Wall-clock budgeted, so only structural properties are asserted."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tree_chain_gpu", os.path.join(root, "examples", "tree_chain_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = mod.run(moves=3, basic_duration=0.15, verbose=False)
    assert len(log) >= 1
    for i, e in enumerate(log):
        assert e["nodes"] > 50 and e["attempts"] >= e["nodes"] - 1        # the budget buys a real tree (loose: shared boxes)
        assert 0.1 <= e["seconds"] < 30.0
        if i > 0 and log[i - 1]["collision_ahead_s"] is None:
            np.testing.assert_allclose(e["start"], log[i - 1]["seed"])       # chained: starts where the last plan said
