#!/usr/bin/env python
"""
Generate the golden fixtures under tests/golden/ by importing and running the reference planner
(jnez71/lqRRT, /root/reference; no file of it is modified) in the build container.

The reference itself ships no tests or golden vectors, so these
fixtures are the parity pin for the oracle (oracle/) and, through it, for the HIP path.
Only DATA is written (inputs + expected outputs); no reference source text is stored.

  python tools/gen_golden.py            # operator-level + small/medium trajectory fixtures
  python tools/gen_golden.py --long     # additionally the 10k-node boat_advanced run (~20 min)

Tie order: the reference's np.argsort(costs) leaves the order of exactly-equal costs
unspecified; the trajectory fixtures pin it to "lowest node id first" by rebinding the name `np` inside
the reference's planner module to a proxy with a stable argsort (ref_loader._StableSortNumpy).  This
changes the car / pendulum trees (car-500: 0c64b54cdd315792 instead of the untouched reference's
219124599a587d8c) and nothing else; the `*_unpatched` fixtures (--job car500u, car2000u, pend150u) come
from the reference with nothing rebound, and tests/test_teacher_cpu.py proves the two differ only between
nodes of bit-equal cost.

Determinism recipe: obstacle seed 0 before the demo's definition section,
np.random.seed(1) right before update_plan, fake clock, xrand_gen=10, exit on max_nodes.
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_loader as rl  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
PLAN_SEED = 1
OBS_SEED = 0


def pid_hash(pid):
    return hashlib.sha1(np.array(pid, np.int64).tobytes()).hexdigest()[:16]


# --------------------------------------------------------------------------- operator level

def random_states(name, ns, rng, count):
    """States spread over (and beyond) the region the planner visits, incl. wrapped headings."""
    n = ns["nstates"]
    if name == "pendulum":
        x = rng.uniform(-4.0, 4.0, (count, n))
        x[:, 2:] *= 2.0
        return x
    x = np.zeros((count, n))
    x[:, 0] = rng.uniform(-10, 70, count)
    x[:, 1] = rng.uniform(-10, 70, count)
    x[:, 2] = rng.uniform(-7.0, 7.0, count)           # beyond +-pi on purpose
    x[:, 3:] = rng.uniform(-1.5, 1.5, (count, n - 3))
    # exact zeros / sign edges exercised by the drag-sign and "no reverse" branches
    x[::7, 3] = 0.0
    x[3::11, 3] = -x[3::11, 3]
    return x


def boundary_states(name, ns, rng, count):
    """States whose hull lies close to an obstacle rim, so is_feasible straddles both outcomes."""
    n = ns["nstates"]
    obs = np.array([o for o in ns.get("obs", [])], dtype=np.float64).reshape(-1, 3)
    real = obs[obs[:, 2] > 0]
    x = random_states(name, ns, rng, count)
    if len(real) == 0:
        return x
    for i in range(count):
        ob = real[rng.randint(len(real))]
        ang = rng.uniform(-np.pi, np.pi)
        reach = rng.uniform(0.0, 6.0)
        x[i, 0] = ob[0] + (ob[2] + reach) * np.cos(ang)
        x[i, 1] = ob[1] + (ob[2] + reach) * np.sin(ang)
        if n >= 6 and name == "boat_advanced":
            # keep most of them inside the planning speed box so collisions decide
            x[i, 3] = rng.uniform(0.0, 1.0)
            x[i, 4] = rng.uniform(-0.35, 0.35)
            x[i, 5] = rng.uniform(-0.18, 0.18)
    return x


def gen_ops(name):
    lq = rl.import_reference()
    ns = rl.load_demo(name, OBS_SEED)
    cfg = rl.DEMOS[name]
    rng = np.random.RandomState(12345)
    n, m = ns["nstates"], ns["ncontrols"]
    B = 256
    out = {}

    # tables the systems are parameterised by
    if "obs" in ns:
        out["obs"] = np.array([o for o in ns["obs"]], dtype=np.float64).reshape(-1, 3)
    if "vps" in ns:
        out["vps"] = np.array(ns["vps"], dtype=np.float64)
    for key in ("B", "invB", "D_pos", "D_neg", "D", "invM", "u_max", "thrust_max", "kp", "kd",
                "velmax_pos", "velmax_neg", "velmax", "velmax_pos_plan", "velmax_neg_plan",
                "boat_length", "goal", "goal_buffer", "error_tol", "sample_space", "goal_bias"):
        if key in ns:
            out["tbl_" + key] = np.array(ns[key], dtype=np.float64)
    out["x0"] = np.array(rl.x0_of(name, ns), dtype=np.float64)
    out["dt"] = np.float64(cfg["dt"])

    # erf
    xg = random_states(name, ns, rng, B)
    x = random_states(name, ns, rng, B)
    out["erf_xg"], out["erf_x"] = xg, x
    out["erf_e"] = np.array([ns["erf"](np.copy(a), np.copy(b)) for a, b in zip(xg, x)])

    # lqr (K only varies; S is a constant diagonal in every shipped demo)
    xs = random_states(name, ns, rng, B)
    SK = [ns["lqr"](np.copy(a), np.zeros(m)) for a in xs]
    out["lqr_x"] = xs
    out["lqr_S"] = np.array(SK[0][0], dtype=np.float64)
    out["lqr_K"] = np.array([k for (_, k) in SK], dtype=np.float64)

    # dynamics (u spans well past the actuator limits so saturation branches fire)
    xs = random_states(name, ns, rng, B)
    if name == "pendulum":
        us = rng.uniform(-300, 300, (B, m))
    else:
        us = rng.uniform(-1.0, 1.0, (B, m)) * np.array([1500.0, 1500.0, 4000.0])[[0, 2] if m == 2 else [0, 1, 2]]
    out["dyn_x"], out["dyn_u"] = xs, us
    out["dyn_xnext"] = np.array([ns["dynamics"](np.copy(a), np.copy(b), cfg["dt"]) for a, b in zip(xs, us)])

    # feasibility: random + rim-straddling
    xs = np.vstack((random_states(name, ns, rng, B), boundary_states(name, ns, rng, 2 * B)))
    us = rng.uniform(-1, 1, (len(xs), m)) * 500.0
    out["feas_x"], out["feas_u"] = xs, us
    out["feas_ok"] = np.array([bool(ns["is_feasible"](np.copy(a), np.copy(b))) for a, b in zip(xs, us)])

    # cost-to-go against a frozen node table (planner.py:340-350)
    planner = rl.make_planner(name, ns, 10)
    import tree as reftree
    nodes = random_states(name, ns, rng, 512)
    t = reftree.Tree(nodes[0], ns["lqr"](nodes[0], np.zeros(m)))
    t.state = np.array(nodes)
    t.size = len(nodes)
    planner.tree = t
    qs = random_states(name, ns, rng, 8)
    out["ctg_nodes"], out["ctg_x"] = nodes, qs
    out["ctg_costs"] = np.array([planner._costs_to_go(np.copy(q)) for q in qs])

    path = os.path.join(OUT, "ops_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, "feasible fraction %.2f" % out["feas_ok"].mean())


# --------------------------------------------------------------------------- trajectory level

def run_traj(name, max_nodes, keep_xrand=512, tag=None, min_time=None, horizon=None, pruning=True, tries=10, guide=None,
             finish_on_goal=False, teacher=False, stable_ties=True, edges_only=False):
    """teacher=True additionally stores EVERY iteration's xrand and tie flag (`xrand_all`, `tie_mask`), which is what the
    teacher-forced parity tests replay decision by decision.  stable_ties=False runs the reference with numpy's own
    (unspecified) argsort tie order, i.e. with nothing patched at all."""
    rl.TIES_STABLE = stable_ties
    ns = rl.load_demo(name, OBS_SEED)
    planner = rl.make_planner(name, ns, max_nodes, min_time=min_time)
    if horizon is not None:
        planner.set_resolution(horizon=horizon)            # (min, max) -> adaptive-horizon heuristic
    n = ns["nstates"]

    xrands, nearest, slen, ties, tie_any = [], [], [], [], []
    ctg, steer = planner._costs_to_go, planner._steer

    def ctg_spy(x):
        xrands.append(np.copy(x))
        c = ctg(x)
        ties.append(int(np.sum(c == c.min())) > 1)
        tie_any.append(len(np.unique(c)) < len(c))
        return c

    def steer_spy(ID, xtar, force_arrive=False):
        r = steer(ID, xtar, force_arrive)
        nearest.append(int(ID))
        slen.append(len(r[0]))
        return r

    planner._costs_to_go = ctg_spy
    planner._steer = steer_spy

    np.random.seed(PLAN_SEED)
    t0 = time.time()
    ret = planner.update_plan(rl.x0_of(name, ns), ns["sample_space"], goal_bias=ns["goal_bias"], xrand_gen=tries, pruning=pruning,
                              guide=guide, finish_on_goal=finish_on_goal)
    wall = time.time() - t0

    # how many (n+1)-double sampler candidates were consumed from the legacy global stream
    probe = np.random.sample()
    rs = np.random.RandomState(PLAN_SEED)
    stream = rs.random_sample((len(xrands) * 12 + 64) * (n + 1))
    pos = int(np.flatnonzero(stream == probe)[0])
    assert pos % (n + 1) == 0
    n_candidates = pos // (n + 1)

    tree = planner.tree
    iters = len(nearest)
    # the fallback plan evaluates one extra _costs_to_go-like contraction inline, not via the spy
    assert len(xrands) == iters
    edge_len = np.array([len(s) for s in tree.x_seq], dtype=np.int32)
    last_u = np.array([np.array(s[-1], dtype=np.float64) for s in tree.u_seq])
    out = dict(
        max_nodes=np.int64(max_nodes), min_time=np.float64(planner.min_time), iterations=np.int64(iters), n_candidates=np.int64(n_candidates),
        returned=np.bool_(ret), reached_goal=np.bool_(planner.plan_reached_goal),
        pID=np.array(tree.pID, dtype=np.int32), state=np.array(tree.state, dtype=np.float64),
        K=np.array([lk[1] for lk in tree.lqr], dtype=np.float64),
        edge_len=edge_len, last_u=last_u,
        nearest=np.array(nearest, dtype=np.int32), steer_len=np.array(slen, dtype=np.int16),
        xrand_head=np.array(xrands[:keep_xrand], dtype=np.float64),
        node_seq=np.array(planner.node_seq, dtype=np.int32),
        plan_x=np.array(planner.x_seq, dtype=np.float64), plan_u=np.array(planner.u_seq, dtype=np.float64),
        plan_T=np.float64(planner.T), pid_hash=np.array(pid_hash(tree.pID)),
        state_sum=np.float64(tree.state.sum()), ref_wall_s=np.float64(wall),
        tie_iterations=np.int64(np.sum(ties)), horizon_iters_final=np.int64(planner.horizon_iters),
        pruning=np.bool_(pruning), tries=np.int64(tries), finish_on_goal=np.bool_(finish_on_goal),
        guide=np.array(guide if guide is not None else [], dtype=np.float64),
        stable_ties=np.bool_(stable_ties),
    )
    if teacher:
        out["xrand_all"] = np.array(xrands, dtype=np.float64)
        out["tie_mask"] = np.array(ties, dtype=np.bool_)          # the MINIMUM cost is shared by several nodes
        out["tie_any_mask"] = np.array(tie_any, dtype=np.bool_)   # some cost value (not nec. the minimum) is repeated
    # a few complete edges (first, a middle one, the last) to pin x_seq/u_seq contents
    for tagid, ID in (("a", 1), ("b", tree.size // 2), ("c", tree.size - 1)):
        out["edge_%s_id" % tagid] = np.int32(ID)
        out["edge_%s_x" % tagid] = np.array(tree.x_seq[ID], dtype=np.float64)
        out["edge_%s_u" % tagid] = np.array(tree.u_seq[ID], dtype=np.float64)
    path = os.path.join(OUT, "traj_%s_%s.npz" % (name, tag or str(max_nodes)))
    if edges_only:
        # EVERY edge interior of the run (tree.x_seq / u_seq, tree.py:121-132) and the plan's interpolators at 64 times
        # (planner.py:451-464), in a file of their own next to the traj fixture of the SAME run (asserted)
        old = np.load(path)
        assert str(old["pid_hash"]) == str(out["pid_hash"]) and np.array_equal(old["state"], out["state"]), "not the committed run"
        e = dict(pid_hash=out["pid_hash"], edge_len=edge_len,
                 x_cat=np.concatenate([np.array(q, dtype=np.float64).reshape(-1, n) for q in tree.x_seq]),
                 u_cat=np.concatenate([np.array(q, dtype=np.float64).reshape(-1, ns["ncontrols"]) for q in tree.u_seq]))
        assert len(e["x_cat"]) == int(edge_len.sum())
        if np.isfinite(planner.T) and planner.T > 0:
            ts = np.concatenate((np.linspace(0.0, float(planner.T), 60), [-1.0, 1.25 * float(planner.T), 0.5 * planner.dt, float(planner.T) - 1e-9]))
            e["interp_t"] = ts
            e["interp_x"] = np.array([planner.get_state(t) for t in ts], dtype=np.float64)
            e["interp_u"] = np.array([planner.get_effort(t) for t in ts], dtype=np.float64)
        epath = os.path.join(OUT, "edges_%s_%s.npz" % (name, tag or str(max_nodes)))
        np.savez_compressed(epath, **e)
        print("wrote %s: %d edges, %d rows, plan T=%s, %d KB" % (epath, tree.size, len(e["x_cat"]), planner.T, os.path.getsize(epath) // 1024))
        return
    np.savez_compressed(path, **out)
    print("wrote %s: iters=%d cand=%d nodes=%d hash=%s sum=%r goal=%s ties=%d wall=%.1fs" % (
        path, iters, n_candidates, tree.size, out["pid_hash"], float(out["state_sum"]),
        bool(planner.plan_reached_goal), int(np.sum(ties)), wall))


def compact_unpatched(name, tag):
    """A 10k-node teacher fixture is 2.5 MB.  The unpatched twin of the headline run differs from the patched one in a handful
    of parents / nearest choices only (bit-equal-cost ties), so it is committed as: every array that differs, plus the list of
    arrays that are IDENTICAL to the patched fixture's (asserted here, on the full output of the unpatched run) -- the tests
    take those from the patched file."""
    full = os.path.join(OUT, "traj_%s_%s_unpatched.npz" % (name, tag))
    a = np.load(os.path.join(OUT, "traj_%s_%s.npz" % (name, tag)))
    b = np.load(full)
    assert not bool(b["stable_ties"]) and bool(a["stable_ties"])
    same, keep = [], {}
    for k in b.files:
        if k in a.files and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]):
            same.append(k)
        else:
            keep[k] = b[k]
    for k in ("state", "K", "xrand_all", "edge_len", "steer_len", "last_u", "tie_mask"):
        assert k in same, "%s differs between the patched and the unpatched run: keep the full fixture" % k
    keep["same_as_patched"] = np.array(sorted(same))
    np.savez_compressed(full, **keep)
    print("compacted %s: kept %s, %d arrays identical to the patched fixture" % (full, sorted(keep), len(same)))


# --------------------------------------------------------------------------- occupancy grid (8f-3)

def gen_ogrid():
    """
    Fixture for the ROS node's occupancy-grid feasibility (demos/lqrrt_ros/nodes/lqrrt_node.py:719-745).
    The node cannot be imported here (rospy, cv2), so only the text of that one method is compiled, at
    generation time, against a stand-in `self`; behaviors/params.py (numpy only) supplies the hull points.
    """
    import ast
    import importlib.util
    import textwrap
    import types
    node_path = os.path.join(rl.REF, "demos", "lqrrt_ros", "nodes", "lqrrt_node.py")
    src = open(node_path).read()
    fn = [n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "is_feasible"][0]
    text = textwrap.dedent("\n".join(src.split("\n")[fn.lineno - 1:fn.end_lineno]))
    spec = importlib.util.spec_from_file_location("ref_params", os.path.join(rl.REF, "demos", "lqrrt_ros", "behaviors", "params.py"))
    params = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(params)
    ns = {"np": np, "params": params}
    exec(compile(text, node_path, "exec"), ns)
    rng = np.random.RandomState(77)
    rows, cols, res = 300, 400, 0.25
    grid = np.zeros((rows, cols), dtype=np.int64)
    for _ in range(60):                                    # blobs of occupancy 0..100 and unknown (-1) patches
        r, c = rng.randint(rows), rng.randint(cols)
        h, w = rng.randint(2, 14), rng.randint(2, 14)
        grid[max(r - h, 0):r + h, max(c - w, 0):c + w] = rng.choice([100, 95, 91, 90, 89, 50, -1])
    me = types.SimpleNamespace(ogrid=grid, blind=False, ogrid_origin=np.array([-20.0, -15.0]), ogrid_cpm=1 / res,
                               ogrid_threshold=float("90"))
    xs = np.zeros((768, 6))
    xs[:, 0] = rng.uniform(-30, 90, len(xs))                # beyond the grid on every side (IndexError / wrap quirks)
    xs[:, 1] = rng.uniform(-25, 70, len(xs))
    xs[:, 2] = rng.uniform(-7, 7, len(xs))
    ok = np.array([bool(ns["is_feasible"](me, np.copy(x), np.zeros(3))) for x in xs])
    out = dict(grid=grid.astype(np.int8), origin=me.ogrid_origin, cpm=np.float64(me.ogrid_cpm),
               threshold=np.float64(me.ogrid_threshold), vps=np.array(params.vps, dtype=np.float64), x=xs, ok=ok)
    path = os.path.join(OUT, "ops_ogrid.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "feasible fraction %.2f, hull points %d" % (ok.mean(), params.vps.shape[1]))


# --------------------------------------------------------------------------- ROS behaviours (8f-2)

def _load_behavior(name):
    """Imports demos/lqrrt_ros/behaviors/<name>.py (it does `from params import *` and builds a Planner)."""
    import importlib.util
    rl.import_reference()
    bdir = os.path.join(rl.REF, "demos", "lqrrt_ros", "behaviors")
    if bdir not in sys.path:
        sys.path.insert(0, bdir)
    spec = importlib.util.spec_from_file_location("ref_behavior_" + name, os.path.join(bdir, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_ros_behaviors():
    """
    Operator and trajectory fixtures for the boat / car / escape behaviours of the ROS package, run through
    their own module-level Planner (adaptive horizon (0.1, 3), FPR 0) with the node's occupancy-grid
    feasibility (method text compiled as in gen_ogrid) and the node's erf (identical to the demos').
    """
    import ast
    import textwrap
    import types
    node_path = os.path.join(rl.REF, "demos", "lqrrt_ros", "nodes", "lqrrt_node.py")
    src = open(node_path).read()
    fn = [n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "is_feasible"][0]
    text = textwrap.dedent("\n".join(src.split("\n")[fn.lineno - 1:fn.end_lineno]))
    erf = rl.load_demo("boat_novice")["erf"]                # same formula as lqrrt_node.py:996-1016
    og = np.load(os.path.join(OUT, "ops_ogrid.npz"))
    grid = np.array(og["grid"], dtype=np.int64)
    cpm, origin = float(og["cpm"]), og["origin"]
    x0 = np.zeros(6)
    goal = np.array([30, 20, np.deg2rad(45), 0, 0, 0])
    for px, py in ((x0[0], x0[1]), (goal[0], goal[1])):      # free start and goal
        c, r = int(cpm * (px - origin[0])), int(cpm * (py - origin[1]))
        grid[max(r - 40, 0):r + 40, max(c - 40, 0):c + 40] = 0
    for name in ("boat", "car", "escape"):
        mod = _load_behavior(name)
        ns = {"np": np, "params": sys.modules["params"]}
        exec(compile(text, node_path, "exec"), ns)
        me = types.SimpleNamespace(ogrid=grid, blind=False, ogrid_origin=origin, ogrid_cpm=cpm, ogrid_threshold=90.0)
        feas = lambda x, u, _f=ns["is_feasible"], _me=me: _f(_me, x, u)
        rng = np.random.RandomState(5)
        out = {}
        # operators
        xs = np.zeros((256, 6))
        xs[:, :2] = rng.uniform(-10, 40, (256, 2))
        xs[:, 2] = rng.uniform(-7, 7, 256)
        xs[:, 3:] = rng.uniform(-1.3, 1.3, (256, 3))
        us = rng.uniform(-1, 1, (256, 3)) * np.array([900.0, 900.0, 3000.0])
        out["dyn_x"], out["dyn_u"] = xs, us
        out["dyn_xnext"] = np.array([mod.dynamics(np.copy(a), np.copy(b), mod.dt) for a, b in zip(xs, us)])
        if name == "boat":
            mod.focus = np.array([12.0, -3.0, 0.0])
            out["focus"] = mod.focus
            out["dyn_xnext_focus"] = np.array([mod.dynamics(np.copy(a), np.copy(b), mod.dt) for a, b in zip(xs, us)])
            mod.focus = None
        SK = [mod.lqr(np.copy(a), np.zeros(3)) for a in xs]
        out["lqr_S"] = np.array(SK[0][0], dtype=np.float64)
        out["lqr_K"] = np.array([k for _, k in SK])
        # trajectory through the module's own planner
        planner = mod.planner
        planner.set_system(erf=erf)
        planner.constraints.set_feasibility_function(feas)
        planner.set_runtime(min_time=2, max_time=3, max_nodes=300, sys_time=lambda: 0.0)
        planner.set_goal(goal)
        ss = mod.gen_ss(x0, goal)
        nearest, slen, xrands = [], [], []
        steer = planner._steer

        def spy(ID, xtar, force_arrive=False, _s=steer):
            r = _s(ID, xtar, force_arrive)
            nearest.append(int(ID)); slen.append(len(r[0])); xrands.append(np.copy(xtar))
            return r
        planner._steer = spy
        np.random.seed(PLAN_SEED)
        ret = planner.update_plan(x0, ss, goal_bias=[0.3, 0.3, 0, 0, 0, 0], xrand_gen=10)
        tree = planner.tree
        out.update(returned=np.bool_(ret), iterations=np.int64(len(nearest)), pID=np.array(tree.pID, dtype=np.int32),
                   state=np.array(tree.state), edge_len=np.array([len(e) for e in tree.x_seq], dtype=np.int32),
                   nearest=np.array(nearest, dtype=np.int32), steer_len=np.array(slen, dtype=np.int16),
                   K=np.array([lk[1] for lk in tree.lqr]), horizon_iters_final=np.int64(planner.horizon_iters),
                   reached_goal=np.bool_(planner.plan_reached_goal), node_seq=np.array(planner.node_seq, dtype=np.int32),
                   plan_x=np.array(planner.x_seq), plan_T=np.float64(planner.T), sample_space=np.array(ss, dtype=np.float64),
                   goal=goal, goal_buffer=np.array(planner.constraints.goal_buffer), error_tol=np.array(planner.error_tol),
                   grid=grid.astype(np.int8), origin=origin, cpm=np.float64(cpm), threshold=np.float64(90.0),
                   pid_hash=np.array(pid_hash(tree.pID)), xrand_all=np.array(xrands, dtype=np.float64))
        path = os.path.join(OUT, "ros_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("wrote %s: iters=%d nodes=%d goal=%s horizon_iters=%d edges<=%d" % (
            path, len(nearest), tree.size, bool(planner.plan_reached_goal), planner.horizon_iters, out["edge_len"].max()))


# --------------------------------------------------------------------------- config 5

def gen_config5(max_nodes=600, n_boxes=3000, tag=None):
    """
    BASELINE config 5 is not in the reference; SURVEY 8(d) names its oracle: "the reference Planner driven by the
    build's own NumPy callbacks for this system".  That is what runs here: the REFERENCE's Planner / Constraints /
    Tree classes with oracle/systems_np.DoubleIntegrator's dynamics / lqr / is_feasible (erf = np.subtract, the
    reference's default), S and K from scipy.linalg.solve_discrete_are.
    """
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from systems_np import DoubleIntegrator
    lq = rl.import_reference()
    s = DoubleIntegrator(n_boxes=n_boxes, seed=OBS_SEED)
    cons = lq.Constraints(nstates=s.nstates, ncontrols=s.ncontrols, goal_buffer=s.goal_buffer, is_feasible=s.is_feasible)
    planner = lq.Planner(s.dynamics, s.lqr, cons, error_tol=s.error_tol, min_time=2, max_time=3, max_nodes=max_nodes,
                         goal0=s.goal, printing=False, sys_time=lambda: 0.0, **s.plan_kwargs)
    # (copies taken BEFORE planning: the reference's fallback plan zeroes columns of the S its lqr returns, in place,
    #  planner.py:314-316)
    S_before, K_before = np.array(s.S), np.array(s.K)
    xrands, nearest, slen, ties = [], [], [], []
    ctg, steer = planner._costs_to_go, planner._steer

    def ctg_spy(x):
        xrands.append(np.copy(x))
        c = ctg(x)
        ties.append(len(np.unique(c)) < len(c))
        return c

    def steer_spy(ID, xtar, force_arrive=False):
        r = steer(ID, xtar, force_arrive)
        nearest.append(int(ID)); slen.append(len(r[0]))
        return r
    planner._costs_to_go, planner._steer = ctg_spy, steer_spy
    np.random.seed(PLAN_SEED)
    t0 = time.time()
    ret = planner.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    wall = time.time() - t0
    n = s.nstates
    probe = np.random.sample()
    stream = np.random.RandomState(PLAN_SEED).random_sample((len(xrands) * 12 + 64) * (n + 1))
    pos = int(np.flatnonzero(stream == probe)[0])
    assert pos % (n + 1) == 0
    tree = planner.tree
    out = dict(max_nodes=np.int64(max_nodes), n_boxes=np.int64(n_boxes), box_seed=np.int64(OBS_SEED), min_time=np.float64(2),
               iterations=np.int64(len(nearest)), n_candidates=np.int64(pos // (n + 1)), returned=np.bool_(ret),
               reached_goal=np.bool_(planner.plan_reached_goal), pID=np.array(tree.pID, dtype=np.int32),
               state=np.array(tree.state), K=np.array([lk[1] for lk in tree.lqr]), S=S_before, Kconst=K_before,
               edge_len=np.array([len(e) for e in tree.x_seq], dtype=np.int32), nearest=np.array(nearest, dtype=np.int32),
               steer_len=np.array(slen, dtype=np.int16), xrand_all=np.array(xrands), tie_any_mask=np.array(ties),
               node_seq=np.array(planner.node_seq, dtype=np.int32), plan_x=np.array(planner.x_seq), plan_u=np.array(planner.u_seq),
               plan_T=np.float64(planner.T), pid_hash=np.array(pid_hash(tree.pID)), box_lo_sum=np.float64(s.box_lo.sum()),
               box_hi_sum=np.float64(s.box_hi.sum()), ref_wall_s=np.float64(wall), pruning=np.bool_(True), tries=np.int64(10))
    for tagid, ID in (("a", 1), ("b", tree.size // 2), ("c", tree.size - 1)):
        out["edge_%s_id" % tagid] = np.int32(ID)
        out["edge_%s_x" % tagid] = np.array(tree.x_seq[ID], dtype=np.float64)
        out["edge_%s_u" % tagid] = np.array(tree.u_seq[ID], dtype=np.float64)
    path = os.path.join(OUT, "traj_double_integrator_%s.npz" % (tag or str(max_nodes)))
    np.savez_compressed(path, **out)
    print("wrote %s: iters=%d cand=%d nodes=%d hash=%s goal=%s ties=%d wall=%.1fs" % (
        path, len(nearest), out["n_candidates"], tree.size, out["pid_hash"], bool(planner.plan_reached_goal), int(np.sum(ties)), wall))


def gen_pendulum_lqr(max_nodes=120, tag=None, eps=None):
    return gen_riccati("pendulum_lqr", max_nodes, tag, eps=eps)


def gen_riccati(name, max_nodes, tag=None, eps=None):
    """
    The north-star steer pipeline on the reference itself: the REFERENCE's Planner with the callbacks of
    oracle/systems_np.PendulumLqr / BoatNoviceLqr, whose lqr linearises the demo's dynamics by central differences and calls
    scipy.linalg.solve_discrete_are for every rollout step, new node and sample.
    """
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from systems_np import SYSTEMS
    lq = rl.import_reference()
    s = SYSTEMS[name](OBS_SEED) if eps is None else SYSTEMS[name](OBS_SEED, eps=eps)     # eps: step of the central differences
    cons = lq.Constraints(nstates=s.nstates, ncontrols=s.ncontrols, goal_buffer=s.goal_buffer, is_feasible=s.is_feasible)
    planner = lq.Planner(s.dynamics, s.lqr, cons, error_tol=s.error_tol, erf=s.erf, min_time=60, max_time=61, max_nodes=max_nodes,
                         goal0=s.goal, printing=False, sys_time=lambda: 0.0, **s.plan_kwargs)
    xrands, nearest, slen, Ssamp = [], [], [], []
    ctg, steer, lqr = planner._costs_to_go, planner._steer, planner.lqr

    def ctg_spy(x):
        xrands.append(np.copy(x))
        Ssamp.append(np.array(lqr(np.copy(x), np.zeros(s.ncontrols))[0]))
        return ctg(x)

    def steer_spy(ID, xtar, force_arrive=False):
        r = steer(ID, xtar, force_arrive)
        nearest.append(int(ID)); slen.append(len(r[0]))
        return r
    planner._costs_to_go, planner._steer = ctg_spy, steer_spy
    np.random.seed(PLAN_SEED)
    t0 = time.time()
    ret = planner.update_plan(s.x0, s.sample_space, goal_bias=s.goal_bias, xrand_gen=10)
    wall = time.time() - t0
    n = s.nstates
    probe = np.random.sample()
    stream = np.random.RandomState(PLAN_SEED).random_sample((len(xrands) * 12 + 64) * (n + 1))
    pos = int(np.flatnonzero(stream == probe)[0])
    tree = planner.tree
    out = dict(max_nodes=np.int64(max_nodes), min_time=np.float64(60), iterations=np.int64(len(nearest)),
               n_candidates=np.int64(pos // (n + 1)), returned=np.bool_(ret), reached_goal=np.bool_(planner.plan_reached_goal),
               pID=np.array(tree.pID, dtype=np.int32), state=np.array(tree.state), K=np.array([lk[1] for lk in tree.lqr]),
               S_nodes=np.array([lk[0] for lk in tree.lqr]), S_samples=np.array(Ssamp),
               edge_len=np.array([len(e) for e in tree.x_seq], dtype=np.int32), nearest=np.array(nearest, dtype=np.int32),
               steer_len=np.array(slen, dtype=np.int16), xrand_all=np.array(xrands), node_seq=np.array(planner.node_seq, dtype=np.int32),
               plan_x=np.array(planner.x_seq), plan_u=np.array(planner.u_seq), plan_T=np.float64(planner.T),
               pid_hash=np.array(pid_hash(tree.pID)), Q=s.Q, R=s.R, eps=np.float64(s.eps), ref_wall_s=np.float64(wall),
               pruning=np.bool_(True), tries=np.int64(10))
    for tagid, ID in (("a", 1), ("b", tree.size // 2), ("c", tree.size - 1)):
        out["edge_%s_id" % tagid] = np.int32(ID)
        out["edge_%s_x" % tagid] = np.array(tree.x_seq[ID], dtype=np.float64)
        out["edge_%s_u" % tagid] = np.array(tree.u_seq[ID], dtype=np.float64)
    path = os.path.join(OUT, "traj_%s_%s.npz" % (name, tag or str(max_nodes)))
    np.savez_compressed(path, **out)
    print("wrote %s: iters=%d cand=%d nodes=%d hash=%s goal=%s wall=%.1fs" % (
        path, len(nearest), out["n_candidates"], tree.size, out["pid_hash"], bool(planner.plan_reached_goal), wall))


def gen_chain(name="car", plans=3, max_nodes=400, frac=0.75):
    """The ROS node's tree chain, made deterministic (lqrrt_node.py:444-484; planner.py:172,311-334,451-464): `plans` chained
    update_plan calls on ONE reference Planner, fake clock, each ended by the node limit, plan k+1 seeded at
    get_state(frac * T_k) of plan k (the node seeds at get_ref(next_runtime)), the obstacle table swapped after the first plan
    (the node writes module globals of the plugin file between plans, :260-263).  Fixture: every tree (parents, states, gains,
    edge lengths), every seed, every plan (node_seq, x_seq, u_seq, T), iterations and sampler rows per plan, the swapped table."""
    rl.TIES_STABLE = True
    ns = rl.load_demo(name, OBS_SEED)
    planner = rl.make_planner(name, ns, max_nodes)
    n = ns["nstates"]
    obs_a = np.array(ns["obs"], dtype=np.float64)
    obs_b = np.copy(obs_a)
    obs_b[:, 0] = np.round(obs_a[:, 0] + 2.5, 2)                 # the same field shifted: the old plan now grazes obstacles
    obs_b[::2, 1] = np.round(obs_a[::2, 1] - 1.75, 2)
    out = dict(plans=np.int64(plans), max_nodes=np.int64(max_nodes), frac=np.float64(frac), obs_a=obs_a, obs_b=obs_b,
               plan_seeds=np.array([PLAN_SEED + 100 + k for k in range(plans)], dtype=np.int64))
    x0 = np.array(rl.x0_of(name, ns), dtype=np.float64)
    steer = planner._steer
    count = [0]

    def steer_spy(ID, xtar, force_arrive=False):
        count[0] += 1
        return steer(ID, xtar, force_arrive)
    planner._steer = steer_spy
    for k in range(plans):
        if k == 1:
            ns["obs"] = obs_b                                    # the plugin's global table rebound, as the node rewrites module globals
                                                                 # (NOT in place: demo_car.py's table is an int64 array and would truncate)
        count[0] = 0
        np.random.seed(int(out["plan_seeds"][k]))
        ret = planner.update_plan(x0, ns["sample_space"], goal_bias=ns["goal_bias"], xrand_gen=10)
        probe = np.random.sample()
        rs = np.random.RandomState(int(out["plan_seeds"][k]))
        stream = rs.random_sample((count[0] * 12 + 64) * (n + 1))
        pos = int(np.flatnonzero(stream == probe)[0])
        assert pos % (n + 1) == 0
        tree = planner.tree
        pre = "p%d_" % k
        out[pre + "x0"] = np.copy(x0)
        out[pre + "returned"] = np.bool_(ret)
        out[pre + "reached_goal"] = np.bool_(planner.plan_reached_goal)
        out[pre + "iterations"] = np.int64(count[0])
        out[pre + "n_candidates"] = np.int64(pos // (n + 1))
        out[pre + "pID"] = np.array(tree.pID, dtype=np.int32)
        out[pre + "state"] = np.array(tree.state, dtype=np.float64)
        out[pre + "K"] = np.array([lk[1] for lk in tree.lqr], dtype=np.float64)
        out[pre + "edge_len"] = np.array([len(q) for q in tree.x_seq], dtype=np.int32)
        out[pre + "node_seq"] = np.array(planner.node_seq, dtype=np.int32)
        out[pre + "plan_x"] = np.array(planner.x_seq, dtype=np.float64)
        out[pre + "plan_u"] = np.array(planner.u_seq, dtype=np.float64)
        out[pre + "plan_T"] = np.float64(planner.T)
        ts = np.array([0.0, 0.25 * planner.T, frac * planner.T, planner.T, 1.5 * planner.T])
        out[pre + "interp_t"] = ts
        out[pre + "interp_x"] = np.array([planner.get_state(t) for t in ts], dtype=np.float64)
        out[pre + "interp_u"] = np.array([planner.get_effort(t) for t in ts], dtype=np.float64)
        print("plan %d: seed state %s -> %d nodes, %d iterations, goal=%s, T=%.2f, hash=%s" % (
            k, np.round(x0, 3), tree.size, count[0], bool(planner.plan_reached_goal), planner.T, pid_hash(tree.pID)))
        x0 = np.array(planner.get_state(frac * planner.T), dtype=np.float64)
    path = os.path.join(OUT, "chain_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote %s (%d KB)" % (path, os.path.getsize(path) // 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true", help="also run boat_advanced to 10k nodes (~20 min)")
    ap.add_argument("--only", default=None, help="comma list: ops,traj")
    ap.add_argument("--job", default=None, help="one teacher / tie-audit job: adv10k, adv10ku, adv3000, car500u, car2000u, pend150u, "
                                                "car500t, car2000t, pend150t (u = unpatched reference, t = teacher data of the patched run)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    what = set((args.only or "ops,traj").split(","))
    jobs = {
        # every iteration's xrand next to nearest / steer_len: input of the teacher-forced parity tests
        "adv10k": lambda: run_traj("boat_advanced", 10000, keep_xrand=64, tag="10k", teacher=True),
        "adv3000": lambda: run_traj("boat_advanced", 3000, keep_xrand=64, teacher=True),
        # every edge interior + the interpolators of runs whose traj fixture is committed (same run, asserted)
        "edges_adv3000": lambda: run_traj("boat_advanced", 3000, keep_xrand=64, teacher=True, edges_only=True),
        "edges_car2000": lambda: run_traj("car", 2000, keep_xrand=64, teacher=True, edges_only=True),
        "car500t": lambda: run_traj("car", 500, teacher=True),
        "car2000t": lambda: run_traj("car", 2000, keep_xrand=64, teacher=True),
        "pend150t": lambda: run_traj("pendulum", 150, teacher=True),
        "int300t": lambda: run_traj("boat_intermediate", 300, teacher=True),
        "nov300t": lambda: run_traj("boat_novice", 300, teacher=True),
        # the reference with NOTHING patched (numpy's own argsort tie order): the tie audit
        "car500u": lambda: run_traj("car", 500, keep_xrand=64, tag="500_unpatched", teacher=True, stable_ties=False),
        "car2000u": lambda: run_traj("car", 2000, keep_xrand=64, tag="2000_unpatched", teacher=True, stable_ties=False),
        "pend150u": lambda: run_traj("pendulum", 150, keep_xrand=64, tag="150_unpatched", teacher=True, stable_ties=False),
        # the headline run (BASELINE config 4, 10k nodes) with nothing patched either (~20 min)
        "adv10ku": lambda: (run_traj("boat_advanced", 10000, keep_xrand=64, tag="10k_unpatched", teacher=True, stable_ties=False),
                            compact_unpatched("boat_advanced", "10k")),
        # BASELINE config 5 on the reference's Planner with the build's NumPy callbacks
        "di600": lambda: gen_config5(600, 3000),
        "di2500": lambda: gen_config5(2500, 3000),
        # finite-difference linearise -> DARE -> K rollout on the reference's Planner (scipy.linalg.solve_discrete_are)
        "plqr120": lambda: gen_pendulum_lqr(120),
        "plqr600": lambda: gen_pendulum_lqr(600),
        # the same run with a linearisation step of 1e-4 instead of 1e-6: the finite differences of a dt = 1 ms model then carry
        # 1e-12 instead of 1e-10 of rounding noise, which the ill-conditioned Riccati equation amplifies (tests/test_pendulum_lqr.py)
        "plqr600e4": lambda: gen_pendulum_lqr(600, tag="600_eps1e-4", eps=1e-4),
        # the same pipeline at the metric's dimension: demo_boat_novice dynamics, 6 states / 3 controls, Riccati lqr about (x, 0)
        "bnlqr400": lambda: gen_riccati("boat_novice_lqr", 400),
        # three chained plans on the car with a map swap in between (lqrrt_node.py:444-484), every tree / seed / plan
        "chain_car": lambda: gen_chain("car"),
    }
    if args.job:
        jobs[args.job]()
        return
    if args.long:
        jobs["adv10k"]()
        return
    if "ops" in what:
        for name in rl.DEMOS:
            gen_ops(name)
    if "ogrid" in what or "ops" in what:
        gen_ogrid()
    if "ros" in what or "traj" in what:
        gen_ros_behaviors()
    if "traj" in what:
        run_traj("boat_advanced", 200)
        run_traj("boat_intermediate", 300)
        run_traj("boat_novice", 300)
        run_traj("car", 500)
        run_traj("pendulum", 150)
        run_traj("car", 2000, keep_xrand=64)
        run_traj("car", 2000, keep_xrand=64, tag="firstgoal", min_time=0)
        run_traj("boat_novice", 1000, keep_xrand=64, tag="firstgoal", min_time=0)
    if "modes" in what or "traj" in what:
        # update_plan's other switches on the reference itself: pruning=False (argmin instead of the ignore-aware
        # argsort, planner.py:239-247) and xrand_gen=1 (a single sampler try, :188-211)
        run_traj("car", 600, keep_xrand=64, tag="nopruning", pruning=False)
        run_traj("boat_novice", 400, keep_xrand=64, tag="nopruning", pruning=False)
        run_traj("car", 600, keep_xrand=64, tag="tries1", tries=1)
        run_traj("boat_intermediate", 300, keep_xrand=64, tag="tries1", tries=1)
    if "plans" in what or "traj" in what:
        # plan extraction off the main line: the guide fallback when the goal was not reached (planner.py:311-328)
        run_traj("car", 60, keep_xrand=64, tag="guide", guide=[35.0, 35.0, 0.0, 0.0, 0.0])
        run_traj("boat_intermediate", 50, keep_xrand=64, tag="guide", guide=[30.0, 10.0, 0.5, 0.0, 0.0, 0.0])
        # (finish_on_goal cannot be pinned this way: with the frozen clock the reference's force-arrive steer, whose
        #  only other exit is np.allclose to the goal, never returns -- its real exit is the wall-clock timeout)
    if "adaptive" in what or "traj" in what:
        run_traj("boat_intermediate", 400, keep_xrand=64, tag="adaptive", horizon=(0.1, 3))
        run_traj("car", 400, keep_xrand=64, tag="adaptive", horizon=(0.1, 3))
        run_traj("boat_advanced", 3000, keep_xrand=64)


if __name__ == "__main__":
    main()
