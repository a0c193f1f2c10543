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
