"""
Host logic of callback mode (lqrrt_amd/callback.py), no GPU: which plugins select it, how `erf` is classified, that a planner with
plain Python plugins can be built and configured anywhere and refuses to PLAN without a device (there is no CPU path).
"""
import numpy as np
import pytest

import lqrrt_amd
from lqrrt_amd import callback


def _wrap(d):
    return np.arctan2(np.sin(d), np.cos(d))


def test_erf_classification():
    assert callback.classify_erf(np.subtract, 6) == ()
    assert callback.classify_erf(lambda g, x: g - x, 4) == ()

    def car_erf(g, x):                                   # demo_car.py:115-126's form
        e = np.subtract(g, x)
        c, s, cg, sg = np.cos(x[2]), np.sin(x[2]), np.cos(g[2]), np.sin(g[2])
        e[2] = np.arctan2(sg * c - cg * s, cg * c + sg * s)
        return e
    assert callback.classify_erf(car_erf, 5) == (2,)
    assert callback.classify_erf(car_erf, 5, declared=(2,)) == (2,)

    def pend_erf(g, x):                                  # demo_pendulum.py:130-142's form
        e = np.subtract(g, x)
        e[:2] = _wrap(e[:2])
        return e
    assert callback.classify_erf(pend_erf, 4) == (0, 1)
    with pytest.raises(ValueError):
        callback.classify_erf(pend_erf, 4, declared=(0,))
    with pytest.raises(ValueError):
        callback.classify_erf(np.subtract, 4, declared=(1,))
    with pytest.raises(ValueError):
        callback.classify_erf(pend_erf, 4, declared=(0, 7))
    # not of the subtract-and-wrap form: scaled, coupled, saturated, wrong shape, raising
    assert callback.classify_erf(lambda g, x: 2.0 * (g - x), 3) is None
    assert callback.classify_erf(lambda g, x: (g - x)[::-1], 3) is None
    assert callback.classify_erf(lambda g, x: np.clip(g - x, -1, 1), 3) is None
    assert callback.classify_erf(lambda g, x: (g - x)[:2], 3) is None
    assert callback.classify_erf(lambda g, x: 1 / 0, 3) is None
    # the probe has a stream of its own
    np.random.seed(9)
    a = np.random.sample()
    np.random.seed(9)
    callback.classify_erf(pend_erf, 4)
    assert np.random.sample() == a


def test_plain_callables_select_callback_mode():
    boat = lqrrt_amd.systems.BoatAdvanced(0)
    feas = lambda x, u: True
    cons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, feas)
    assert cons.system is None and cons.is_feasible is feas
    np.testing.assert_array_equal(cons.feasible_batch(np.zeros((3, 6))), [True, True, True])
    dyn = lambda x, u, dt: x + dt * np.concatenate((x[3:], u))
    lqr = lambda x, u: (np.eye(6), np.hstack((np.eye(3), np.eye(3))))
    p = lqrrt_amd.Planner(dyn, lqr, cons, horizon=2, dt=0.1, printing=False)
    assert p.callback_mode and p.system is None
    assert p.horizon_iters == 20 and p.nstates == 6 and p.ncontrols == 3
    # one plain callable among native handles is enough
    ncons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
    assert lqrrt_amd.Planner(dyn, boat.lqr, ncons, horizon=2, dt=0.1, erf=boat.erf, printing=False).callback_mode
    assert lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, horizon=2, dt=0.1, erf=boat.erf, printing=False).callback_mode
    assert lqrrt_amd.Planner(boat.dynamics, boat.lqr, ncons, horizon=2, dt=0.1, erf=lambda g, x: g - x, printing=False).callback_mode
    native = lqrrt_amd.Planner(boat.dynamics, boat.lqr, ncons, horizon=2, dt=0.1, erf=boat.erf, printing=False)
    assert not native.callback_mode and native.system is boat
    # mistakes among NATIVE handles still raise
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, boat.lqr, ncons, horizon=2, dt=0.1)      # np.subtract erf on an angular native system
    car = lqrrt_amd.systems.Car()
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, car.lqr, ncons, horizon=2, dt=0.1, erf=boat.erf)
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(dyn, 3, cons, horizon=2)                                 # planner.py:572
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(dyn, lqr, cons, horizon=2, erf=3)                        # planner.py:590
    with pytest.raises(ValueError):
        lqrrt_amd.Constraints(6, 3, boat.goal_buffer, 3)                           # constraints.py:61
    # the mode follows the plugins through set_system
    native.set_system(dyn, lqr)
    assert native.callback_mode
    native.set_system(boat.dynamics, boat.lqr)
    assert not native.callback_mode
    # ... and through a feasibility function swapped on the Constraints object itself (lqrrt_node.py:65), at the next update_plan
    assert not native.callback_mode
    native.set_goal(None)
    ncons.set_feasibility_function(feas)
    assert native.update_plan(boat.x0, boat.sample_space) is False and native.callback_mode
    ncons.set_feasibility_function(boat.is_feasible)
    assert native.update_plan(boat.x0, boat.sample_space) is False and not native.callback_mode
    # update_plans is for native planners
    p.set_goal(boat.goal)
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([dict(planner=p, x0=boat.x0, sample_space=boat.sample_space)])


def test_callback_mode_without_a_device():
    from lqrrt_amd import _native as nat
    if nat.available():
        pytest.skip("a device is present")
    dyn = lambda x, u, dt: x + dt * u
    lqr = lambda x, u: (np.eye(2), np.eye(2))
    cons = lqrrt_amd.Constraints(2, 2, [0.1, 0.1], lambda x, u: True)
    p = lqrrt_amd.Planner(dyn, lqr, cons, horizon=1, dt=0.1, printing=False)
    assert p.warm_up_error is not None
    assert p.update_plan([0, 0], [(0, 1), (0, 1)]) is False        # no goal: the reference's answer, no device needed (planner.py:157-161)
    np.testing.assert_array_equal(p.get_state(0.5), [0, 0])
    p.set_goal([1, 1])
    with pytest.raises(RuntimeError):
        p.update_plan([0, 0], [(0, 1), (0, 1)])                    # planning needs the GPU: no CPU path


def test_held_elsewhere_counts_references():
    """tree.held_elsewhere decides whether the previous plan's tree must be copied out of HBM before the engine is reused: only when
    somebody besides the planner's attribute still holds the tree or one of its feature sequences."""
    from lqrrt_amd.tree import Tree, held_elsewhere

    class Owner(object):
        pass
    o = Owner()
    o.tree = Tree(np.zeros(3), None)
    assert held_elsewhere(o, "tree") is False
    kept = o.tree
    assert held_elsewhere(o, "tree") is True
    del kept
    for name in ("x_seq", "u_seq", "lqr"):
        rows = getattr(o.tree, name)
        assert held_elsewhere(o, "tree") is True, name
        del rows
    box = [o.tree]
    assert held_elsewhere(o, "tree") is True
    del box
    assert held_elsewhere(o, "tree") is False
