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
    with pytest.raises(ValueError):
        import lqrrt_amd
        lqrrt_amd.Planner(lambda x, u, dt: x, s.lqr, lqrrt_amd.Constraints(4, 2, s.goal_buffer, s.is_feasible), horizon=2)
