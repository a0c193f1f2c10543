"""
Pins the oracle (oracle/lqrrt_oracle.py + oracle/systems_np.py) against the fixtures generated
from the reference by tools/gen_golden.py (tie order stabilised for car / pendulum).  CPU only.

Tolerances: parent arrays / nearest ids / edge lengths / iteration and RNG-consumption counts
are compared EXACTLY.  Floating-point values are compared at 1e-9 absolute (they are bit-equal
on the generating machine; NumPy dispatches arctan2/tanh to CPU-specific SIMD code elsewhere).
"""
import hashlib
import os

import numpy as np
import pytest

import systems_np
from systems_np import SYSTEMS, make_oracle_planner

ATOL = 1e-9
DEMO_SYSTEMS = ["boat_advanced", "boat_intermediate", "boat_novice", "car", "pendulum"]


def _load(golden_dir, fname):
    path = os.path.join(golden_dir, fname)
    if not os.path.exists(path):
        pytest.fail("fixture %s missing: tests/golden is committed, a lost fixture must not turn into a pass" % fname)
    return np.load(path)


@pytest.fixture(scope="module", params=DEMO_SYSTEMS)
def sys_and_ops(request, golden_dir):
    name = request.param
    return name, SYSTEMS[name](0), _load(golden_dir, "ops_%s.npz" % name)


def test_tables(sys_and_ops):
    name, s, g = sys_and_ops
    if "obs" in g.files and name != "pendulum":
        np.testing.assert_array_equal(np.asarray(s.obs, dtype=np.float64).reshape(-1, 3), g["obs"])
    if "vps" in g.files:
        np.testing.assert_array_equal(s.vps, g["vps"])
    for key in ("B", "invB", "D_pos", "D_neg", "D", "invM", "u_max", "kp", "kd"):
        if "tbl_" + key in g.files:
            np.testing.assert_array_equal(np.asarray(getattr(s, key), dtype=np.float64), g["tbl_" + key])
    np.testing.assert_array_equal(np.asarray(s.goal, dtype=np.float64), g["tbl_goal"])
    np.testing.assert_array_equal(np.asarray(s.goal_buffer, dtype=np.float64), g["tbl_goal_buffer"])
    np.testing.assert_array_equal(np.asarray(s.error_tol, dtype=np.float64), g["tbl_error_tol"])
    np.testing.assert_array_equal(np.asarray(s.sample_space, dtype=np.float64), g["tbl_sample_space"])
    np.testing.assert_array_equal(np.asarray(s.goal_bias, dtype=np.float64), g["tbl_goal_bias"])
    np.testing.assert_array_equal(np.asarray(s.x0, dtype=np.float64), g["x0"])


def test_erf(sys_and_ops):
    name, s, g = sys_and_ops
    e = np.array([s.erf(np.copy(a), np.copy(b)) for a, b in zip(g["erf_xg"], g["erf_x"])])
    np.testing.assert_allclose(e, g["erf_e"], rtol=0, atol=ATOL)
    # vectorised form == scalar form, row by row
    eb = s.batch_erf(np.copy(g["erf_xg"][0]), np.copy(g["erf_x"]))
    es = np.array([s.erf(np.copy(g["erf_xg"][0]), np.copy(b)) for b in g["erf_x"]])
    np.testing.assert_array_equal(eb, es)


def test_lqr(sys_and_ops):
    name, s, g = sys_and_ops
    S, _ = s.lqr(g["lqr_x"][0], np.zeros(s.ncontrols))
    np.testing.assert_array_equal(np.asarray(S, dtype=np.float64), g["lqr_S"])
    K = np.array([s.lqr(np.copy(a), np.zeros(s.ncontrols))[1] for a in g["lqr_x"]], dtype=np.float64)
    np.testing.assert_allclose(K, g["lqr_K"], rtol=0, atol=ATOL)


def test_dynamics(sys_and_ops):
    name, s, g = sys_and_ops
    xn = np.array([s.dynamics(np.copy(a), np.copy(b), float(g["dt"])) for a, b in zip(g["dyn_x"], g["dyn_u"])])
    np.testing.assert_allclose(xn, g["dyn_xnext"], rtol=0, atol=ATOL)


def test_feasibility(sys_and_ops):
    name, s, g = sys_and_ops
    ok = np.array([bool(s.is_feasible(np.copy(a), np.copy(b))) for a, b in zip(g["feas_x"], g["feas_u"])])
    np.testing.assert_array_equal(ok, g["feas_ok"])


def test_costs_to_go(sys_and_ops):
    name, s, g = sys_and_ops
    from lqrrt_oracle import RefTree
    for vectorised in (False, True):
        p = make_oracle_planner(s, 10, vectorised_nn=vectorised)
        t = RefTree(g["ctg_nodes"][0], s.lqr(g["ctg_nodes"][0], np.zeros(s.ncontrols)))
        t.state = np.array(g["ctg_nodes"])
        t.size = len(t.state)
        p.tree = t
        c = np.array([p._costs_to_go(np.copy(q)) for q in g["ctg_x"]])
        np.testing.assert_allclose(c, g["ctg_costs"], rtol=1e-12, atol=ATOL)
        np.testing.assert_array_equal(np.argmin(c, axis=1), np.argmin(g["ctg_costs"], axis=1))


TRAJ = [("boat_advanced", "200"), ("boat_intermediate", "300"), ("boat_novice", "300"), ("car", "500"),
        ("pendulum", "150"), ("car", "2000"), ("car", "firstgoal"), ("boat_novice", "firstgoal"),
        ("boat_intermediate", "adaptive"), ("car", "adaptive"),
        ("car", "nopruning"), ("boat_novice", "nopruning"), ("car", "tries1"), ("boat_intermediate", "tries1"),
        ("car", "guide"), ("boat_intermediate", "guide")]


@pytest.mark.parametrize("name,tag", TRAJ)
def test_trajectory(golden_dir, name, tag):
    g = _load(golden_dir, "traj_%s_%s.npz" % (name, tag))
    s = SYSTEMS[name](0)
    extra = dict(horizon=(0.1, 3)) if tag == "adaptive" else {}       # adaptive-horizon heuristic, planner.py:418-425
    p = make_oracle_planner(s, int(g["max_nodes"]), min_time=float(g["min_time"]),
                            max_time=max(float(g["min_time"]), 1.0), **extra)
    np.random.seed(1)
    pruning = bool(g["pruning"]) if "pruning" in g.files else True            # the "modes" fixtures carry their switches
    tries = int(g["tries"]) if "tries" in g.files else 10
    guide = g["guide"] if "guide" in g.files and len(g["guide"]) else None    # fallback-plan fixtures (planner.py:311-328)
    ret = p.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=tries, pruning=pruning, guide=guide, trace=True)
    assert ret == bool(g["returned"])
    assert p.iterations == int(g["iterations"])
    assert p.sampler.candidates == int(g["n_candidates"])
    np.testing.assert_array_equal(np.array(p.trace["nearest"], dtype=np.int32), g["nearest"])
    np.testing.assert_array_equal(np.array(p.trace["steer_len"], dtype=np.int16), g["steer_len"])
    np.testing.assert_array_equal(np.array(p.tree.pID, dtype=np.int32), g["pID"])
    assert hashlib.sha1(np.array(p.tree.pID, np.int64).tobytes()).hexdigest()[:16] == str(g["pid_hash"])
    np.testing.assert_allclose(p.tree.state, g["state"], rtol=0, atol=ATOL)
    np.testing.assert_array_equal(np.array([len(e) for e in p.tree.x_seq], dtype=np.int32), g["edge_len"])
    nh = len(g["xrand_head"])
    np.testing.assert_allclose(np.array(p.trace["xrand"][:nh]), g["xrand_head"], rtol=0, atol=0)
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
    # interpolators (planner.py:451-464)
    tq = 0.37 * p.T
    assert np.all(np.isfinite(p.get_state(tq))) and np.all(np.isfinite(p.get_effort(tq)))
