"""
Control surface of a running plan on the GPU: kill_update from ANOTHER thread (the ROS node's use, lqrrt_node.py:678,803,823 --
planner.py:289 polls the flag once per iteration, this build between native calls of at most four waves), what the planner's
attributes hold afterwards (planner.py:330-336), and when the HBM pools are allocated.
"""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _boat_planner(**over):
    import lqrrt
    boat = lqrrt.systems.BoatAdvanced(0)
    cons = lqrrt.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
    kw = dict(error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal, printing=False, wave_size=256, **boat.plan_kwargs)
    kw.update(over)
    return boat, lqrrt.Planner(boat.dynamics, boat.lqr, cons, **kw)


@pytest.mark.parametrize("delay", [0.05, 0.4])
def test_kill_update_from_another_thread(delay):
    boat, p = _boat_planner(min_time=5.0, max_time=5.0, max_nodes=200000, sys_time=time.time)
    assert p._engine is not None                       # explicit max_nodes: the pools exist before the clock starts
    # one short plan first: kernels loaded, sampler warm (what a node that replans continuously looks like)
    p.set_runtime(min_time=0.05, max_time=0.05)
    np.random.seed(1)
    assert p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias) is True
    p.set_runtime(min_time=5.0, max_time=5.0)
    stamp = {}

    def kill():
        stamp["kill"] = time.perf_counter()
        p.kill_update()
    timer = threading.Timer(delay, kill)
    np.random.seed(1)
    t0 = time.perf_counter()
    timer.start()
    ret = p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
    t1 = time.perf_counter()
    timer.join()
    assert ret is False                                # planner.py:330-334
    assert p.killed is False                           # the flag is consumed (planner.py:333)
    assert t1 - t0 < 1.0, "the 5 s budget was not cut short"
    assert t1 - stamp["kill"] < 0.010, "update_plan returned %.1f ms after kill_update" % (1e3 * (t1 - stamp["kill"]))
    # the tree is the one that was growing, and whatever plan it had found is the planner's plan (planner.py:276-281)
    assert p.tree.size > 100 and p.tree.pID[0] == -1
    if p.plan_reached_goal:
        assert p.node_seq[0] == 0 and p.node_seq[-1] < p.tree.size
        xs, us = p.tree.trajectory(p.node_seq)
        np.testing.assert_array_equal(np.array(p.x_seq), np.array(xs))
        assert p.T == len(p.x_seq) * p.dt and len(p.t_seq) == len(p.x_seq)
        assert p._in_goal(p.x_seq[-1])
    # and the planner plans again at once
    p.set_runtime(min_time=0.05, max_time=0.1)
    np.random.seed(1)
    assert p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias) is True


def test_unkill_withdraws_a_kill_that_was_not_seen():
    boat, p = _boat_planner(min_time=0.05, max_time=0.05, max_nodes=50000, sys_time=time.time)
    p.kill_update()
    p.unkill()                                         # planner.py:605-610
    np.random.seed(1)
    assert p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias) is True


def test_pools_are_sized_when_they_are_needed():
    """A planner left at the reference's default max_nodes = 1e5 allocates nothing until it plans (or warm_up() is called); its
    footprint is what Engine.footprint() says, and the documented per-node figure holds."""
    boat, p = _boat_planner()                          # max_nodes left at 1e5
    assert p._engine is None and p.warm_up_error is None
    p.set_runtime(min_time=0.02, max_time=0.02, sys_time=time.time)
    np.random.seed(1)
    p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
    assert p._engine is not None
    fp = p._engine.footprint()
    per_node = 8 * (6 + 3 + 2 + 3 * 6 + 20 * (6 + 3)) + 8          # state, trig, werr, K, edges (H = 20), parent + edge length
    assert fp["device_bytes"] >= p._engine.capacity * per_node
    assert fp["device_bytes"] < p._engine.capacity * per_node + 64 * 2 ** 20       # wave buffers, partial minima, sample pools: < 64 MB
    assert 0 < fp["pinned_bytes"] < 16 * 2 ** 20
    boat2, q = _boat_planner()
    q.warm_up()                                        # the explicit form
    assert q._engine is not None


def test_previous_tree_is_copied_out_only_if_somebody_holds_it():
    """The engine is reused by the next plan.  A Tree the caller kept (the ROS node does: lqrrt_node.py:477) is snapshotted to the host
    first and stays what it was; a tree nobody refers to any more is simply dropped -- no copy out of HBM (tens of milliseconds for a
    100k-node tree, between two plans of a replanning loop)."""
    import lqrrt_amd.tree as tree_mod
    boat, p = _boat_planner(min_time=0.0, max_time=1.0, max_nodes=3000, sys_time=lambda: 0.0)
    calls = []
    orig = tree_mod.Tree._detach

    def counting(self):
        if self._e is not None:
            calls.append(self.size)
        return orig(self)
    tree_mod.Tree._detach = counting
    try:
        np.random.seed(1)
        p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
        np.random.seed(2)
        p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)          # nobody kept the first tree
        assert calls == []
        kept = p.tree
        state, parents, edge7 = np.array(kept.state), list(kept.pID), [np.array(v) for v in kept.x_seq[7]]
        rows = p.tree.u_seq                                                           # holding a feature sequence counts as holding the tree
        np.random.seed(3)
        p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
        assert calls == [kept.size] and kept is not p.tree and not kept.on_device
        np.testing.assert_array_equal(kept.state, state)
        assert list(kept.pID) == parents and len(rows) == kept.size
        for a, b in zip(kept.x_seq[7], edge7):
            np.testing.assert_array_equal(a, b)
        assert not np.array_equal(p.tree.state[:50], state[:50])                       # the new plan grew another tree
    finally:
        tree_mod.Tree._detach = orig


def test_path_extraction_reads_the_path_not_the_tree():
    """Tree.climb on the engine's host mirror (lqrrt_tree_climb) and Tree.trajectory through one device-side gather
    (lqrrt_tree_get_edges_of) give what the node-by-node reads give."""
    boat, p = _boat_planner(min_time=0.0, max_time=1.0, max_nodes=6000, sys_time=lambda: 0.0)
    np.random.seed(4)
    p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
    eng, tree = p._engine, p.tree
    parents = eng.parents()
    for end in (0, 1, tree.size // 2, tree.size - 1, int(p.node_seq[-1])):
        walk = [end]
        while parents[walk[-1]] != -1:
            walk.append(int(parents[walk[-1]]))
        walk.reverse()
        assert eng.climb(end) == walk == tree.climb(end)
        xs, us = tree.trajectory(walk)
        want_x = np.vstack([eng.edge(i)[0] for i in walk])
        want_u = np.vstack([eng.edge(i)[1] for i in walk])
        np.testing.assert_array_equal(np.array(xs), want_x)
        np.testing.assert_array_equal(np.array(us), want_u)
    with pytest.raises(ValueError):
        eng.climb(tree.size)
    np.testing.assert_array_equal(np.array(p.x_seq), np.vstack([eng.edge(i)[0] for i in p.node_seq]))
