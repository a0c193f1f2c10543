"""
Load the reference (jnez71/lqRRT at /root/reference) in this build container.  No reference file is modified.  By
default ONE name is rebound after import: `np` inside the reference's `planner` module becomes a proxy whose argsort is
stable (_StableSortNumpy below), which only decides between nodes of bit-equal cost (tests/test_teacher_cpu.py
test_tie_audit_unpatched_reference); TIES_STABLE = False / import_reference(stable_ties=False) leaves everything untouched
(the `*_unpatched` fixtures).

This module is test-infrastructure tooling: it only works where /root/reference exists
(the build container) and is used by tools/gen_golden.py to produce the committed fixtures
under tests/golden/.  Nothing in lqrrt_amd/, bench.py or the gpu tests imports it.

Recipe:
  * the reference is Python-2 dialect with implicit relative imports
    (lqrrt/__init__.py:1-2, lqrrt/planner.py:17-18) -> put BOTH the repo root and the
    package directory on sys.path so `constraints`, `planner`, `tree` resolve top-level;
  * demo scripts plan+plot at module level -> exec only the text above the
    "################################################# PLAN" banner;
  * always pass xrand_gen=10 (Py3 raises on `None > 0`, planner.py:188; 10 is what Py2 selects).
"""
import os
import sys

import numpy as np

REF = os.environ.get("LQRRT_REFERENCE", "/root/reference")
PLAN_BANNER = "################################################# PLAN"

# Planner kwargs copied from each demo's PLAN section (file:line cited).
DEMOS = {
    # demo_boat_advanced.py:245-249
    "boat_advanced": dict(file="demo_boat_advanced.py", horizon=2, dt=0.1, FPR=0.9, min_time=2, max_time=3),
    # demo_boat_intermediate.py:228-232
    "boat_intermediate": dict(file="demo_boat_intermediate.py", horizon=2, dt=0.1, FPR=0.5, min_time=2, max_time=3),
    # demo_boat_novice.py:182-186
    "boat_novice": dict(file="demo_boat_novice.py", horizon=2, dt=0.1, FPR=0.5, min_time=1, max_time=2),
    # demo_car.py:200-204 (FPR not passed -> ctor default 0, planner.py:86)
    "car": dict(file="demo_car.py", horizon=5, dt=0.1, FPR=0, min_time=2, max_time=3),
    # demo_pendulum.py:172-176 passes horizon=0 which raises ValueError (planner.py:548-553);
    # the build picks horizon=0.05 (H=50) and documents it.
    "pendulum": dict(file="demo_pendulum.py", horizon=0.05, dt=0.001, FPR=0.5, min_time=60, max_time=61),
}


class _StableSortNumpy(object):
    """
    Stand-in for the name `np` inside the reference's planner module ONLY (numpy itself is
    untouched): identical to numpy except that argsort defaults to kind="stable".

    Why: planner.py:240 orders candidate parents with np.argsort(costs), whose order among
    EXACTLY equal costs is unspecified (introsort / AVX-512 sorting networks, CPU dependent).
    Exact ties are real -- e.g. a stopped car adds a child whose state equals its parent's
    (tests/golden/traj_car_500: nodes 16 and 17) -- so the fixtures pin the tie order to the
    one well-defined choice, lowest node id first, which is also what planner.py:247
    (np.argmin, pruning=False) does.  Any tie order is a valid behaviour of the reference.
    """

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    def argsort(self, a, *args, **kwargs):
        if not args and "kind" not in kwargs:
            kwargs["kind"] = "stable"
        return self._real.argsort(a, *args, **kwargs)


TIES_STABLE = True     # module-wide default for import_reference(); tools/gen_golden.py flips it for the unpatched runs


def import_reference(stable_ties=None):
    """Returns the reference `lqrrt` module (Constraints, Planner).  stable_ties=False leaves the reference's
    planner module entirely untouched (numpy's own argsort)."""
    if stable_ties is None:
        stable_ties = TIES_STABLE
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present at %s (build container only)" % REF)
    for p in (os.path.join(REF, "lqrrt"), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import lqrrt  # noqa: E402  (the reference package)
    import planner as refplanner  # noqa: E402  (the reference's planner module)
    if stable_ties:
        if not isinstance(refplanner.np, _StableSortNumpy):
            refplanner.np = _StableSortNumpy(np)
    elif isinstance(refplanner.np, _StableSortNumpy):
        refplanner.np = np
    return lqrrt


def load_demo(name, obstacle_seed=0):
    """
    Executes the definition section of a reference demo and returns its namespace
    (dynamics, lqr, erf, is_feasible, x0/q, goal, goal_buffer, error_tol, sample_space,
    goal_bias, obs, vps, ...).
    """
    import_reference()
    cfg = DEMOS[name]
    path = os.path.join(REF, "demos", cfg["file"])
    with open(path) as f:
        text = f.read()
    head = text.split(PLAN_BANNER)[0]
    ns = {"__name__": "refdemo_" + name}
    np.random.seed(obstacle_seed)
    exec(compile(head, path, "exec"), ns)
    return ns


def make_planner(name, ns, max_nodes, fake_clock=True, min_time=None):
    """
    Builds the reference Planner for a demo namespace with the demo's own kwargs.  With the
    fake clock (time never advances) and the demo's min_time > 0 the only exit is
    tree.size > max_nodes (planner.py:311); min_time=0 instead stops at the first goal hit
    (planner.py:293).
    """
    lq = import_reference()
    cfg = DEMOS[name]
    cons = lq.Constraints(nstates=ns["nstates"], ncontrols=ns["ncontrols"],
                          goal_buffer=ns["goal_buffer"], is_feasible=ns["is_feasible"])
    kw = dict(horizon=cfg["horizon"], dt=cfg["dt"], FPR=cfg["FPR"],
              error_tol=ns["error_tol"], erf=ns["erf"],
              min_time=cfg["min_time"] if min_time is None else min_time,
              max_time=cfg["max_time"], max_nodes=max_nodes,
              goal0=ns["goal"], printing=False)
    if fake_clock:
        kw["sys_time"] = lambda: 0.0
    return lq.Planner(ns["dynamics"], ns["lqr"], cons, **kw)


def x0_of(name, ns):
    return ns["q"] if name == "pendulum" else ns["x0"]
