"""
Pins the C oracle (oracle/lqrrt_oracle.c) against the fixtures generated from the reference (tie order
stabilised for car / pendulum; the untouched reference: tests/test_teacher_cpu.py).  CPU only.

Exact: iterations, sampler rows consumed, parent arrays, edge lengths, per-iteration nearest ids
and edge lengths.  Floating point: 1e-9 absolute -- except on demo_boat_advanced, whose dynamics
are chaotic near standstill (DESIGN.md "Conditioning"): there the free run is a conditioning smoke test
(parents exact for the first 150 nodes, states of that prefix to machine precision in the median); what
pins the C oracle to the reference on that problem is teacher forcing, tests/test_teacher_cpu.py (all
36,936 decisions of the 10k-node run).  Round 4: the heading torque of a moving boat is one atan2
(rudder_term in lqrrt_oracle.c), which moved the first ulp-triggered divergence of the 200-node free run
from decision 1381 to 1264; with torque_vmin = inf (the reference's sequence everywhere) the parent array
of the 200-node fixture is reproduced exactly, asserted below as well.
"""
import os

import numpy as np
import pytest

import coracle
import lqrrt_amd

CASES = [("boat_advanced", "200"), ("boat_intermediate", "300"), ("boat_novice", "300"), ("car", "500"),
         ("pendulum", "150"), ("car", "2000"), ("car", "firstgoal"), ("boat_novice", "firstgoal"),
         ("boat_intermediate", "adaptive"), ("car", "adaptive"),
         ("car", "nopruning"), ("boat_novice", "nopruning"), ("car", "tries1"), ("boat_intermediate", "tries1"),
         ("car", "guide"), ("boat_intermediate", "guide")]


@pytest.mark.parametrize("name,tag", CASES)
def test_coracle_trajectory(golden_dir, name, tag):
    path = os.path.join(golden_dir, "traj_%s_%s.npz" % (name, tag))
    if not os.path.exists(path):
        pytest.fail("fixture missing: tests/golden is committed, a lost fixture must not turn into a pass")
    g = np.load(path)
    s = lqrrt_amd.systems.SYSTEMS[name](0)
    pruning = bool(g["pruning"]) if "pruning" in g.files else True
    tries = int(g["tries"]) if "tries" in g.files else 10
    o = coracle.make(s, int(g["max_nodes"]), seed=1, tries=tries, horizon=(0.1, 3) if tag == "adaptive" else None)
    o.enable_trace(int(g["iterations"]) + 16)
    first_goal = float(g["min_time"]) == 0.0
    reason = o.extend(max_nodes=int(g["max_nodes"]), pruning=pruning, stop_on_goal=first_goal)
    assert reason == (4 if first_goal else 2)
    if name == "boat_advanced":
        _boat_advanced_free_run(o, g, tries)
        return
    assert o.iterations == int(g["iterations"])
    assert o.candidates == int(g["n_candidates"])
    np.testing.assert_array_equal(o.parents(), g["pID"])
    near, ln = o.trace()
    np.testing.assert_array_equal(near, g["nearest"])
    err = np.abs(o.states() - g["state"]).max(axis=1)
    np.testing.assert_array_equal(o.edge_lengths(), g["edge_len"])
    np.testing.assert_array_equal(ln, g["steer_len"].astype(np.int32))
    assert err.max() < 1e-9
    np.testing.assert_allclose(o.gains(), g["K"], rtol=0, atol=1e-8)
    for t in "abc":
        ID = int(g["edge_%s_id" % t])
        x, u = o.edge(ID)
        np.testing.assert_allclose(x, g["edge_%s_x" % t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(u, g["edge_%s_u" % t], rtol=0, atol=1e-6)
    assert (o.hits > 0) == bool(g["reached_goal"])
    if tag == "adaptive":
        assert o.horizon_iters == int(g["horizon_iters_final"])


def _boat_advanced_free_run(o, g, tries):
    """Conditioning smoke test (module docstring): common prefix with the reference's run >= 150 nodes; and, with the
    reference's torque sequence everywhere (torque_vmin = inf), the whole parent array of the 200-node fixture."""
    def prefix(oo):
        p, q = oo.parents(), g["pID"]
        n = min(len(p), len(q))
        d = np.flatnonzero(p[:n] != q[:n])
        return int(d[0]) if len(d) else n
    first = prefix(o)
    # (the measured first divergence of the default torque form, one atan2 above 1 cm/s: node 152 on every fixture of this problem;
    #  with the reference's sequence 201 = never / 211 / 211 -- tests/test_hip_parity.py FREE_RUN_FIRST_DIVERGENCE has the table)
    assert first == 152, "parents leave the reference's at node %d" % first
    err = np.abs(o.states()[:first] - g["state"][:first]).max(axis=1)
    assert np.median(err) < 1e-12 and np.mean(err < 1e-9) > 0.8
    s = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
    s.torque_vmin = np.inf
    r = coracle.make(s, int(g["max_nodes"]), seed=1, tries=tries)
    r.enable_trace(int(g["iterations"]) + 16)
    assert r.extend(max_nodes=int(g["max_nodes"])) == 2
    assert r.iterations == int(g["iterations"]) and r.candidates == int(g["n_candidates"])
    np.testing.assert_array_equal(r.parents(), g["pID"])
    near, ln = r.trace()
    # chaotic edges (boat at standstill with saturated thrusters) may be cut one collision later/earlier
    assert np.mean(r.edge_lengths() == g["edge_len"]) > 0.98
    assert np.mean(ln == g["steer_len"].astype(np.int32)) > 0.99
    err = np.abs(r.states() - g["state"]).max(axis=1)
    assert np.median(err) < 1e-12 and np.mean(err < 1e-9) > 0.8


@pytest.mark.parametrize("name", ["boat_advanced", "boat_intermediate", "boat_novice", "car", "pendulum"])
def test_coracle_operators(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "ops_%s.npz" % name))
    s = lqrrt_amd.systems.SYSTEMS[name](0)
    o = coracle.make(s, 16)
    e = np.array([o.erf(a, b) for a, b in zip(g["erf_xg"], g["erf_x"])])
    np.testing.assert_allclose(e, g["erf_e"], rtol=0, atol=1e-12)
    K = np.array([o.gain(a, np.zeros(s.ncontrols)) for a in g["lqr_x"]])
    np.testing.assert_allclose(K, g["lqr_K"], rtol=0, atol=1e-11)
    xn = np.array([o.dynamics(a, b) for a, b in zip(g["dyn_x"], g["dyn_u"])])
    np.testing.assert_allclose(xn, g["dyn_xnext"], rtol=0, atol=1e-12)
    ok = np.array([o.feasible(a, b) for a, b in zip(g["feas_x"], g["feas_u"])])
    np.testing.assert_array_equal(ok, g["feas_ok"])


def test_coracle_matches_numpy_oracle_beyond_fixtures():
    """Two independent restatements (NumPy callbacks vs plain C) agree on a seed no fixture covers."""
    from systems_np import SYSTEMS, make_oracle_planner
    for name, nodes in (("car", 400), ("boat_intermediate", 250)):
        s = lqrrt_amd.systems.SYSTEMS[name](0)
        o = coracle.make(s, nodes, seed=9)
        o.extend(max_nodes=nodes)
        rs = SYSTEMS[name](0)
        ref = make_oracle_planner(rs, nodes, min_time=2, max_time=3)
        np.random.seed(9)
        ref.update_plan(rs.x0, rs.sample_space, goal_bias=rs.goal_bias, xrand_gen=10)
        np.testing.assert_array_equal(o.parents(), np.array(ref.tree.pID, dtype=np.int32))
        assert o.iterations == ref.iterations
        np.testing.assert_allclose(o.states(), ref.tree.state, rtol=0, atol=1e-9)
        np.testing.assert_array_equal(o.ignored(), ref._ignored)


def test_synchronous_wave_oracle_properties():
    """orc_extend_sync: wave size 1 IS the sequential algorithm; larger waves are deterministic,
    consume the same sample stream, and no sample of a wave takes a node born in that wave as its parent."""
    s = lqrrt_amd.systems.SYSTEMS["boat_intermediate"](0)
    a = coracle.make(s, 400, seed=4)
    a.extend(max_iters=3000, max_nodes=300)
    b = coracle.make(s, 400, seed=4)
    b.extend_sync(1, max_iters=3000, max_nodes=300)
    np.testing.assert_array_equal(a.parents(), b.parents())
    np.testing.assert_array_equal(a.states(), b.states())
    assert a.iterations == b.iterations and a.candidates == b.candidates
    wave = 64
    c = coracle.make(s, 3000, seed=4)
    c.enable_trace(4000)
    c.extend_sync(wave, max_iters=2048, max_nodes=10 ** 6)
    d = coracle.make(s, 3000, seed=4)
    d.extend_sync(wave, max_iters=2048, max_nodes=10 ** 6)
    np.testing.assert_array_equal(c.parents(), d.parents())
    np.testing.assert_array_equal(c.states(), d.states())
    assert c.iterations == 2048
    near, length = c.trace()
    size_at_wave_start, size = [], 1
    for k in range(c.iterations):
        if k % wave == 0:
            start = size
        assert near[k] < start                      # parents come from the wave-start snapshot only
        size += 1 if length[k] > 0 else 0
    assert size == c.size
    assert not np.array_equal(c.parents()[:min(c.size, a.size)], a.parents()[:min(c.size, a.size)])   # a different tree
