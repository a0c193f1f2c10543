"""The documented way to add a problem is exercised end to end as far as a machine without a GPU
can: the out-of-tree example header builds into a complete engine with hipcc (done by __graft_entry__.build()), the library
loads through the same binding, exports the whole ABI, knows LQRRT_MODEL_USER (the stock library does not), and its kernels
obey the same resource rules as the built-in ones."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USER_LIB = os.path.join(ROOT, "examples", "user_system", "liblqrrt_unicycle.so")


def _create_rc(libpath, model, n, m):
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from lqrrt_amd import _native as nat\n"
            "d = nat.SystemDesc(); d.model, d.nstates, d.ncontrols, d.n_params = %d, %d, %d, 6\n"
            "h = C.c_void_p(); rc = nat.lib().lqrrt_engine_create(C.byref(d), 0, 64, 64, C.byref(h))\n"
            "missing = [k for k in nat.SIGNATURES if not hasattr(nat.lib(), k)]\n"
            "print(rc, nat.lib().lqrrt_abi_version(), len(missing), nat.lib().lqrrt_last_error().decode())\n" % (ROOT, model, n, m))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LQRRT_LIB=libpath), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    rc, abi, missing, msg = out.stdout.strip().split(" ", 3)
    return int(rc), int(abi), int(missing), msg


def test_user_system_library_builds_loads_and_knows_the_model():
    import lqrrt_amd
    from lqrrt_amd import _native as nat
    if not os.path.exists(USER_LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(USER_LIB), "examples/user_system/liblqrrt_unicycle.so was not built"
    import torch
    gpu = torch.cuda.is_available()
    rc, abi, missing, msg = _create_rc(USER_LIB, nat.MODEL_USER, 4, 2)
    assert abi == nat.lib().lqrrt_abi_version() and missing == 0
    assert rc == (0 if gpu else nat.E_NODEVICE), (rc, msg)        # the model is known: only the device is missing here
    rc, _, _, msg = _create_rc(USER_LIB, nat.MODEL_USER, 6, 3)      # the dimensions are the header's: N = 4, M = 2
    assert rc == nat.E_ARG and "nstates=4" in msg
    rc, _, _, msg = _create_rc(nat.LIB_PATH, nat.MODEL_USER, 4, 2)  # the stock library has no user problem
    assert rc == nat.E_ARG and "unknown model" in msg


def test_user_system_host_description():
    sys.path.insert(0, os.path.join(ROOT, "examples", "user_system"))
    import plan_unicycle
    s = plan_unicycle.make_system()
    d, keep = s.desc()
    assert (d.model, d.nstates, d.ncontrols, d.n_params, d.n_obstacles) == (100, 4, 2, 6, 24)
    import lqrrt_amd
    native = lqrrt_amd.Planner(s.dynamics, s.lqr, lqrrt_amd.Constraints(4, 2, s.goal_buffer, s.is_feasible), horizon=2, erf=s.erf)
    assert not native.callback_mode and native.system is s
    mixed = lqrrt_amd.Planner(lambda x, u, dt: x, s.lqr, lqrrt_amd.Constraints(4, 2, s.goal_buffer, s.is_feasible), horizon=2, erf=s.erf)
    assert mixed.callback_mode                       # one plain Python callable: the host loop (lqrrt_amd/callback.py)


USER_ORACLE = os.path.join(ROOT, "examples", "user_system", "liblqrrt_unicycle_oracle.so")


def test_user_system_gets_a_sequential_oracle():
    """The same header compiled for the host (tools/build_user_system.py --oracle) drives the sequential C oracle: the
    out-of-tree problem plans on CPU, its edges re-simulate with its own dynamics callback and its states are feasible under its
    own is_feasible (64 lanes emulated one after the other).  The GPU suite then asserts HIP == this oracle bit for bit
    (tests/test_user_system_gpu.py), the net every built-in system has."""
    assert os.path.exists(USER_ORACLE), "examples/user_system/liblqrrt_unicycle_oracle.so missing: __graft_entry__.build() makes it"
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import coracle, plan_unicycle
assert coracle.use_user_model(%r) == (4, 2)
s = plan_unicycle.make_system()
o = coracle.make(s, 400, seed=1)
o.extend(max_nodes=400)
assert o.size == 401 and o.parents()[0] == -1
st, pid = o.states(), o.parents()
for i in range(1, o.size, 5):
    xs, us = o.edge(i)
    assert 1 <= len(xs) <= 20 and np.array_equal(xs[-1], st[i])
    prev = np.vstack((st[pid[i]][None, :], xs[:-1]))
    for a, u, b in zip(prev, us, xs):
        assert np.array_equal(o.dynamics(a, u), b)
        assert o.feasible(b, u)
centres = s.obs
d = np.sqrt(((st[:, None, :2] - centres[None, :, :2]) ** 2).sum(-1))
assert np.all(d[1:] > centres[None, :, 2])                 # no node inside an (inflated) obstacle
print("OK", o.iterations, o.hits)
""" % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "examples", "user_system"), USER_ORACLE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
