"""The tree chain, end to end against the reference.

tests/golden/chain_car.npz is the REFERENCE's own Planner (tools/gen_golden.py --job chain_car) driven the way the ROS node
drives it (lqrrt_node.py:444-484), made deterministic: three chained update_plan calls on one Planner, fake clock, each ended by
the node limit (planner.py:311, return value False by design :330-334), plan k+1 seeded at get_state(0.75 T_k) of plan k
(planner.py:451-464; the node seeds at get_ref(next_runtime)), the obstacle table swapped after the first plan (the node rewrites
the plugin's module globals between plans, :260-263).

  * CPU: the NumPy oracle (oracle/lqrrt_oracle.py RefPlanner + oracle/systems_np.Car) walks the same chain -- every tree's
    parents, every seed, every plan, iteration and sampler-row counts EXACT, floats bit-equal on the generating machine (1e-12 here).
  * GPU: lqrrt_amd.Planner walks it through the C ABI -- parents, edge lengths, node_seq, counts exact; seeds, states and plans
    within 1e-9, efforts 1e-6.  Each side chains from ITS OWN plan (nothing is teacher-forced): an error in a plan, an
    interpolator or the re-seeding would compound from plan to plan."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

ATOL = 1e-9


def _fixture(golden_dir):
    path = os.path.join(golden_dir, "chain_car.npz")
    if not os.path.exists(path):
        pytest.fail("fixture chain_car.npz missing: tests/golden is committed, a lost fixture must not turn into a pass")
    return np.load(path)


def _walk(g, planner, system, set_obstacles, atol_x, atol_u, exact_floats):
    plans, frac = int(g["plans"]), float(g["frac"])
    x0 = np.array(g["p0_x0"])
    for k in range(plans):
        pre = "p%d_" % k
        if k == 1:
            set_obstacles(np.array(g["obs_b"]))
        # the seed each side computed from ITS OWN previous plan
        np.testing.assert_allclose(x0, g[pre + "x0"], rtol=0, atol=0 if exact_floats else atol_x)
        np.random.seed(int(g["plan_seeds"][k]))
        ret = planner.update_plan(x0, system.sample_space, goal_bias=system.goal_bias, xrand_gen=10)
        after = np.random.sample()
        rs = np.random.RandomState(int(g["plan_seeds"][k]))
        rs.random_sample(int(g[pre + "n_candidates"]) * (system.nstates + 1))
        assert after == rs.random_sample(), "plan %d left np.random somewhere else than the reference did" % k
        assert bool(ret) == bool(g[pre + "returned"]) and bool(planner.plan_reached_goal) == bool(g[pre + "reached_goal"])
        tree = planner.tree
        assert list(tree.pID) == g[pre + "pID"].tolist(), "plan %d: parents differ" % k
        np.testing.assert_allclose(np.array(tree.state), g[pre + "state"], rtol=0, atol=atol_x)
        assert [len(e) for e in tree.x_seq] == g[pre + "edge_len"].tolist()
        K = np.array([lk[1] for lk in tree.lqr]) if not hasattr(planner, "_engine") else planner._engine.gains()
        np.testing.assert_allclose(K, g[pre + "K"], rtol=0, atol=1e-8)
        assert list(planner.node_seq) == g[pre + "node_seq"].tolist()
        assert planner.T == float(g[pre + "plan_T"])
        np.testing.assert_allclose(np.array(planner.x_seq), g[pre + "plan_x"], rtol=0, atol=atol_x)
        np.testing.assert_allclose(np.array(planner.u_seq), g[pre + "plan_u"], rtol=0, atol=atol_u)
        for t, xw, uw in zip(g[pre + "interp_t"], g[pre + "interp_x"], g[pre + "interp_u"]):
            np.testing.assert_allclose(planner.get_state(t), xw, rtol=0, atol=atol_x)
            np.testing.assert_allclose(planner.get_effort(t), uw, rtol=0, atol=atol_u)
        x0 = np.array(planner.get_state(frac * planner.T), dtype=np.float64)
    set_obstacles(np.array(g["obs_a"]))


def test_numpy_oracle_walks_the_reference_chain(golden_dir):
    from systems_np import SYSTEMS, make_oracle_planner
    g = _fixture(golden_dir)
    rs = SYSTEMS["car"](0)
    np.testing.assert_array_equal(rs.obs, g["obs_a"])
    ref = make_oracle_planner(rs, int(g["max_nodes"]), min_time=2, max_time=3, vectorised_nn=False)

    def swap(obs):
        rs.obs = obs
    _walk(g, ref, rs, swap, 1e-12, 1e-9, exact_floats=False)
    assert int(g["p1_iterations"]) == 431 and int(g["p2_iterations"]) == 450            # the fixture is the committed run


@pytest.mark.gpu
@pytest.mark.parametrize("wave", [64, 256])
def test_hip_planner_walks_the_reference_chain(golden_dir, wave):
    import lqrrt_amd as lqrrt
    g = _fixture(golden_dir)
    s = lqrrt.systems.SYSTEMS["car"](0)
    np.testing.assert_array_equal(np.asarray(s.obs).reshape(-1, 3), g["obs_a"])
    cons = lqrrt.Constraints(s.nstates, s.ncontrols, s.goal_buffer, s.is_feasible)
    kw = dict(s.plan_kwargs)
    kw.update(error_tol=s.error_tol, erf=s.erf, min_time=2, max_time=3, max_nodes=int(g["max_nodes"]), goal0=s.goal,
              sys_time=lambda: 0.0, printing=False, wave_size=wave)
    p = lqrrt.Planner(s.dynamics, s.lqr, cons, **kw)
    trees = []

    def swap(obs):
        s.set_obstacles(obs)                                    # the engine re-uploads on its next use (system.revision)
    _walk(g, p, s, swap, ATOL, 1e-6, exact_floats=False)
    for k in range(int(g["plans"])):
        assert p.stats["attempts"] >= 0
    trees.append(p.tree)
    assert p.tree.size == int(g["max_nodes"]) + 1
