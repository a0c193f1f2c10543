"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/lqrrt_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os
import sys
import re

import numpy as np
import pytest

import lqrrt_amd
from lqrrt_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "lqrrt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lqrrt_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    lib = nat.lib()
    names = _declared()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "liblqrrt_hip.so does not export %s" % name
        assert name in nat.SIGNATURES, "%s has no ctypes signature" % name
    for name in nat.SIGNATURES:
        assert name in names, "%s bound but not declared in the header" % name
    assert lib.lqrrt_abi_version() == 1


def test_struct_sizes_match_header():
    # sizes implied by include/lqrrt_hip.h (natural alignment)
    import ctypes as C
    assert C.sizeof(nat.SystemDesc) == 16 + 8 * 96 + 16 + 16 + 8 + 8 + 16 + 8 + 8
    assert C.sizeof(nat.Resolution) == 16 + 8 + 4 * 8 * 12 + 16
    assert C.sizeof(nat.SamplerDesc) == 3 * 8 * 12 + 8
    assert C.sizeof(nat.ExtendStats) == 9 * 8 + 8


def test_mismatched_native_handles_are_rejected():
    """(Plain Python callables select callback mode: tests/test_callback_cpu.py.)"""
    boat = lqrrt_amd.systems.BoatAdvanced(0)
    cons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, horizon=2, dt=0.1)      # np.subtract erf on an angular system
    car = lqrrt_amd.systems.Car()
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, car.lqr, cons, horizon=2, dt=0.1, erf=boat.erf)


def test_reference_error_conventions():
    boat = lqrrt_amd.systems.BoatAdvanced(0)
    with pytest.raises(ValueError):
        lqrrt_amd.Constraints(6, 3, [1, 2, 3], boat.is_feasible)                  # constraints.py:49
    cons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
    kw = dict(error_tol=boat.error_tol, erf=boat.erf, printing=False)
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, horizon=0, dt=0.1, **kw)  # planner.py:553
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, horizon=2, dt=0.1, min_time=3, max_time=1, **kw)  # :504
    with pytest.raises(ValueError):
        lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, horizon=2, dt=0.1, goal0=[1, 2], **kw)            # :480
    p = lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, horizon=0.3, dt=0.1, **kw)
    assert p.horizon_iters == int(0.3 / 0.1) == 2                                  # planner.py:549 float floor
    assert p.update_plan(boat.x0, boat.sample_space) is False                      # no goal -> False, :157-161
    np.testing.assert_array_equal(p.get_state(0.3), boat.x0)


def test_no_gpu_means_no_compute():
    if nat.device_count() > 0:
        pytest.skip("a GPU is present")
    boat = lqrrt_amd.systems.BoatAdvanced(0)
    cons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
    p = lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, error_tol=boat.error_tol, erf=boat.erf,
                          goal0=boat.goal, printing=False, **boat.plan_kwargs)
    with pytest.raises(nat.NativeError):
        p.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)
    with pytest.raises(nat.NativeError):
        boat.dynamics(boat.x0, np.zeros(3), 0.1)


def test_system_tables_match_reference_fixtures(golden_dir):
    for name, cls in lqrrt_amd.systems.SYSTEMS.items():
        path = os.path.join(golden_dir, "ops_%s.npz" % name)
        if not os.path.exists(path):
            continue
        g = np.load(path)
        s = cls(0)
        if "vps" in g.files:
            np.testing.assert_array_equal(s.vps, g["vps"])
        if "obs" in g.files and name != "pendulum":
            np.testing.assert_array_equal(np.asarray(s.obs).reshape(-1, 3), g["obs"])
        for key in ("B", "invB", "D_pos", "D_neg", "D", "invM", "u_max"):
            if "tbl_" + key in g.files:
                np.testing.assert_array_equal(np.asarray(getattr(s, key), dtype=np.float64), g["tbl_" + key])
        np.testing.assert_array_equal(np.asarray(s.goal, dtype=np.float64), g["tbl_goal"])
        np.testing.assert_array_equal(np.asarray(s.error_tol, dtype=np.float64), g["tbl_error_tol"])
        np.testing.assert_array_equal(np.asarray(s.sample_space, dtype=np.float64), g["tbl_sample_space"])


def test_dare_matches_scipy():
    import scipy.linalg
    from lqrrt_amd.dare import dare_doubling
    rng = np.random.RandomState(3)
    for n, m in ((2, 1), (4, 2), (12, 6)):
        if n == 12:
            dt = 0.1
            A = np.eye(n); A[:6, 6:] = dt * np.eye(6)
            B = np.vstack((np.zeros((6, 6)), dt * np.eye(6)))
        else:
            A = np.eye(n) + 0.1 * rng.randn(n, n)
            B = rng.randn(n, m)
        Q, R = np.eye(n), np.eye(m)
        S, K = dare_doubling(A, B, Q, R)
        S_ref = scipy.linalg.solve_discrete_are(A, B, Q, R)
        K_ref = np.linalg.solve(R + B.T @ S_ref @ B, B.T @ S_ref @ A)
        np.testing.assert_allclose(S, S_ref, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(K, K_ref, rtol=1e-10, atol=1e-10)


def test_integration_stub_matches_binding():
    """INTEGRATION.md shows the ctypes struct a maintainer would write; it must list the fields of the real binding
    (lqrrt_amd/_native.py SystemDesc == include/lqrrt_hip.h lqrrt_system_desc) in order."""
    import re
    from lqrrt_amd import _native as nat
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class _SystemDesc\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n", text, re.S)
    assert m, "stub not found"
    names = re.findall(r'\("(\w+)"', m.group(1))
    assert names == [f[0] for f in nat.SystemDesc._fields_]


def test_reference_package_name_and_tree_constructor():
    """`import lqrrt` resolves to this build (lqrrt/__init__.py:1-2 of the reference exports Constraints, Planner) and
    Tree has the reference's constructor Tree(seed_state, seed_lqr) (tree.py:50) with its host-side behaviour."""
    import lqrrt
    import lqrrt_amd
    assert lqrrt.Planner is lqrrt_amd.Planner and lqrrt.Constraints is lqrrt_amd.Constraints and lqrrt.Tree is lqrrt_amd.Tree
    S, K = np.eye(3), np.arange(6.0).reshape(2, 3)
    t = lqrrt.Tree([1.0, 2.0, 3.0], (S, K))
    assert (t.size, t.nstates, t.ncontrols, t.pID) == (1, 3, 2, [-1]) and not t.on_device
    assert t.state.shape == (1, 3) and t.lqr[0][1] is K
    assert np.array_equal(t.x_seq[0][0], [1.0, 2.0, 3.0]) and np.array_equal(t.u_seq[0][0], np.zeros(2))   # tree.py:69-70
    t.add_node(0, [2.0, 2.0, 2.0], (S, K + 1), [np.ones(3), 2 * np.ones(3)], [np.zeros(2), np.ones(2)])
    t.add_node(1, [3.0, 3.0, 3.0], (S, K + 2), [3 * np.ones(3)], [np.ones(2)])
    assert t.size == 3 and t.climb(2) == [0, 1, 2] and t.state.shape == (3, 3)
    xs, us = t.trajectory([0, 1, 2])
    assert len(xs) == 4 and len(us) == 4 and np.array_equal(xs[-1], 3 * np.ones(3))
    with pytest.raises(ValueError, match="doesn't exist"):
        t.add_node(3, [0, 0, 0], None, [], [])                    # tree.py:83-84
    with pytest.raises(ValueError, match="doesn't exist"):
        t.climb(9)                                                 # tree.py:109-110


def test_steer_kernels_use_no_scratch():
    """the rollout kernels must not spill to scratch memory.  hipcc cross-compiles for gfx950 here and
    reports the private segment size of every kernel (-Rpass-analysis=kernel-resource-usage)."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    if not shutil.which(kr.HIPCC) and not os.path.exists(kr.HIPCC):
        pytest.skip("hipcc not available")
    # (compiled with the out-of-tree example problem in: a user's kernels obey the same rule)
    user = os.path.join(ROOT, "examples", "user_system", "unicycle.hpp")
    rows = kr.parse(kr.remarks(["-DLQRRT_USER_SYSTEM=\"%s\"" % user]))
    steer = [r for r in rows if "k_steer<" in r["name"]]
    assert any("UserSystem" in r["name"] for r in steer)
    scan = [r for r in rows if "k_nn_scan<" in r["name"]]
    assert len(steer) >= 14 and len(scan) >= 20
    # No spilling: ScratchSize 0 -- or, where the register allocator left a small frame RESERVED (these kernels live at the SGPR
    # limit; an emergency slot of a few dozen bytes comes and goes with unrelated edits), not one instruction that touches it.
    framed = [r for r in steer if r["scratch"] != 0]
    assert all(r["scratch"] <= 64 for r in framed), [(r["name"][:60], r["scratch"]) for r in framed]
    if framed:
        touched = kr.private_memory_instructions([r["mangled"] for r in framed], ["-DLQRRT_USER_SYSTEM=\"%s\"" % user])
        assert all(v == 0 for v in touched.values()), touched
    # the scan is fed by the scalar unit: no LDS at all -- except the opt-in two-level form (template argument WPB = 4,
    # LQRRT_NN_WG4), whose four wavefronts combine their minima through 3 KB of it
    two_level = [r for r in scan if r["name"].split("(")[0].rstrip(">").endswith(", 4")]
    assert two_level and all(r["scratch"] == 0 and r["lds"] == 4 * 64 * (8 + 4) for r in two_level)
    assert all(r["scratch"] == 0 and r["lds"] == 0 for r in scan if r not in two_level)
    # Round 5: the scan's speed is its occupancy (profiles/r05_nn_regression.txt -- in round 4 an innocent
    # `by * WPB + (threadIdx.x >> 6)` made the node loop's index a vector value: vector loads instead of s_load_dwordx8, 117 -> 155
    # VGPRs, 4 -> 3 wavefronts per SIMD, 13 -> 27 us for the full-size launch, and nothing noticed).  Wavefronts per SIMD that the
    # instantiations of the bench configurations must keep: <system, S form, in-wave, patch, WPB>.
    def occ(sys_, dense, tri, patch, wpb):
        key = "lq::k_nn_scan<lq::%s, %d, %s, %s, %d>" % (sys_, dense, "true" if tri else "false", "true" if patch else "false", wpb)
        hit = [r for r in scan if r["name"].startswith("void " + key)]
        assert len(hit) == 1, key
        return hit[0]["occupancy"]
    for sys_ in ("BoatAdvanced", "BoatIntermediate", "BoatNovice", "RosBoat", "BoatNoviceLqr"):
        assert occ(sys_, 0, False, False, 1) >= 4 and occ(sys_, 0, False, True, 1) >= 4 and occ(sys_, 0, True, False, 1) >= 5, sys_
        assert occ(sys_, 0, False, False, 4) >= 4, sys_
    assert occ("Car", 0, False, False, 1) >= 4 and occ("Car", 0, True, False, 1) >= 4
    assert occ("Pendulum", 0, False, False, 1) >= 6 and occ("Pendulum", 0, True, False, 1) >= 6
    assert occ("DoubleIntegratorT<6>", 3, False, False, 1) >= 4 and occ("DoubleIntegratorT<6>", 3, True, False, 1) >= 7


def test_scan_node_loop_is_fed_by_the_scalar_unit():
    """The ISA of the headline tree scan: its node fetches are s_load_dwordx8 (four nodes of one state component), and vector
    memory instructions appear only outside the node loop (samples in, partial minima out, the ignore words)."""
    import shutil
    import subprocess
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not shutil.which(hipcc) and not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    d = tempfile.mkdtemp()
    src = os.path.join(d, "scan_only.hip")
    with open(src, "w") as f:
        f.write('#include <hip/hip_runtime.h>\n#include "kernels.hpp"\nnamespace lq {\n'
                'template __global__ void k_nn_scan<BoatAdvanced, 0, false, false, 1>(NodeView, const double*, const double*, int, '
                'const double*, int, Part*, int*, int, int, IgnPatch);\n'
                'template __global__ void k_nn_scan<BoatAdvanced, 0, false, false, 4>(NodeView, const double*, const double*, int, '
                'const double*, int, Part*, int*, int, int, IgnPatch);\n}\n')
    asm = os.path.join(d, "scan_only.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I", os.path.join(ROOT, "lqrrt_amd", "csrc"), "-I", os.path.join(ROOT, "include"), src, "-o", asm])
    text = open(asm).read()
    for wpb in (1, 4):
        import re
        m = re.search(r"\n(_ZN2lq9k_nn_scanINS_12BoatAdvancedELi0ELb0ELb0ELi%dEEEvNS_8NodeViewE\w+):" % wpb, text)
        body = text[m.start():]
        body = body[:body.index("s_endpgm")]
        n_s8 = body.count("s_load_dwordx8")
        n_vec = body.count("global_load_") + body.count("flat_load_") + body.count("buffer_load_")
        assert n_s8 >= 12, (wpb, n_s8)              # three modes x (6 state components + angle data) in two loop forms
        assert n_vec <= 16, (wpb, n_vec)            # round 4's broken build: 99
        # one partial minimum = ONE 16-byte store (round 6; two scattered stores were 69 % of the scan's physical traffic)
        assert body.count("global_store_dwordx4") == 1 and body.count("global_store_dword ") + body.count("global_store_dwordx2") == 0, wpb


def test_planner_call_budget_follows_the_clock():
    """Host logic of the time budget: a native call commits at most four waves, at most what fits in half of
    the time left at the measured rate, at least a small wave; synchronous mode only whole waves."""
    boat = lqrrt_amd.systems.BoatAdvanced(0)
    cons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)
    p = lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal,
                          printing=False, wave_size=256, **boat.plan_kwargs)
    assert p._attempt_budget(None, 1.0) == 256                         # no rate yet: one wave
    assert p._attempt_budget(4e5, np.inf) == 1024                      # no deadline: four waves
    assert p._attempt_budget(4e5, 1.0) == 1024                         # plenty of time
    assert p._attempt_budget(4e5, 1e-3) == 200                         # 0.5 * 4e5/s * 1 ms
    assert p._attempt_budget(4e5, 0.0) == 32 and p._attempt_budget(4e5, -1.0) == 32
    q = lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal,
                          printing=False, wave_size=256, wave_mode="synchronous", **boat.plan_kwargs)
    assert q._attempt_budget(4e5, 1e-3) == 256 and q._attempt_budget(4e5, 2e-3) == 256 and q._attempt_budget(4e5, 1.0) == 1024
    assert p.tree is None and p._engine is None                         # no GPU here: nothing was created, nothing raised
    assert isinstance(p.warm_up_error, Exception)                       # ... and the reason is on record (diagnosable before the first plan)


def test_update_plans_argument_rules():
    """Host logic of lqrrt_amd.update_plans (several planners through shared native calls): which combinations it refuses, before
    anything touches a device."""
    boat = lqrrt_amd.systems.BoatAdvanced(0)
    cons = lqrrt_amd.Constraints(6, 3, boat.goal_buffer, boat.is_feasible)

    def mk(**over):
        kw = dict(error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal, printing=False, wave_size=256, **boat.plan_kwargs)
        kw.update(over)
        return lqrrt_amd.Planner(boat.dynamics, boat.lqr, cons, **kw)
    assert lqrrt_amd.update_plans([]) == []
    a, b = mk(), mk()
    job = lambda p, **kw: dict(planner=p, x0=boat.x0, sample_space=boat.sample_space, goal_bias=boat.goal_bias, **kw)
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), job(a)])                                     # the same planner twice
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), job(mk(max_nodes=5000))])                    # different node limits
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), job(mk(wave_size=128))])
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), job(b, pruning=False)])
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), job(mk(wave_mode="synchronous"))])
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), job(b, nonsense=1)])
    car = lqrrt_amd.systems.Car()
    ccons = lqrrt_amd.Constraints(car.nstates, car.ncontrols, car.goal_buffer, car.is_feasible)
    c = lqrrt_amd.Planner(car.dynamics, car.lqr, ccons, error_tol=car.error_tol, erf=car.erf, goal0=car.goal, printing=False,
                          wave_size=256, **car.plan_kwargs)
    with pytest.raises(ValueError):
        lqrrt_amd.update_plans([job(a), dict(planner=c, x0=car.x0, sample_space=car.sample_space)])   # two system types
    a.set_goal(None)
    b.set_goal(None)
    assert lqrrt_amd.update_plans([job(a), job(b)]) == [False, False]                # no goal: update_plan's answer, per planner
