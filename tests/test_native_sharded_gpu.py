"""
GPU: lqrrt_engine_extend_sharded -- the native loop with one collective per wave (include/lqrrt_hip.h) -- must give exactly
the tree of the single-engine loop, for every rank of every world size.

One GPU checks the whole data path through the loopback communicator: a process plays rank r of G, the slices of the other
ranks are speculated here INTO THEIR ALL-GATHER BLOCKS, their local records are wiped, and the wave continues from what
k_shard_unpack_prep takes out of the blocks (headers, compacted edges, "tail full" -> re-steer).  A second test runs the real
RCCL communicator (ncclCommInitRank through the dlopen'ed librccl of this process) with a world of one.  N > 1 processes on N
GPUs are the driver's SCALE run (bench.py --gpus N uses this very loop).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(name, cap, wave, seed=1, sync=False, device=0, **sys_kw):
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.DoubleIntegrator(**sys_kw) if name == "double_integrator" else lqrrt_amd.systems.SYSTEMS[name](0)
    eng = Engine(s, capacity=cap, max_wave=wave, device=device)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(seed).get_state()
    eng.set_mt19937(st[1], st[2])
    if sync:
        eng.set_wave_mode("synchronous")
    eng.tree_reset(s.x0)
    return s, eng


def _same_tree(a, b):
    assert a.size == b.size
    np.testing.assert_array_equal(a.parents(), b.parents())
    np.testing.assert_array_equal(a.states(), b.states())
    np.testing.assert_array_equal(a.gains(), b.gains())
    np.testing.assert_array_equal(a.edge_lengths(), b.edge_lengths())
    np.testing.assert_array_equal(a.ignored(), b.ignored())
    xa, ua, la = a.edges()
    xb, ub, lb = b.edges()
    for i in range(a.size):                      # the edges travelled compacted: every recorded step must have arrived
        np.testing.assert_array_equal(xa[i, :la[i]], xb[i, :lb[i]])
        np.testing.assert_array_equal(ua[i, :la[i]], ub[i, :lb[i]])
    assert a.plan_best() == b.plan_best()


@pytest.mark.parametrize("name,nodes,wave,world,rank,scheme", [
    ("boat_advanced", 1500, 256, 2, 0, "sample"), ("boat_advanced", 1500, 256, 2, 1, "sample"),
    ("boat_advanced", 2500, 256, 8, 5, "sample"), ("car", 1200, 256, 3, 2, "sample"),
    ("boat_advanced", 1500, 1024, 4, 1, "sample"),                     # waves beyond the fused rounds' 256
    ("boat_advanced", 1500, 256, 4, 3, "tree"), ("boat_intermediate", 900, 128, 8, 0, "tree"),
    ("double_integrator", 1200, 512, 4, 2, "tree"), ("double_integrator", 1200, 256, 2, 1, "sample"),
    # round 4: Riccati systems (K per recorded step, S per sample) are sample-sharded too, through the fused rounds
    ("boat_novice_lqr", 500, 64, 2, 1, "sample"), ("pendulum_lqr", 150, 64, 4, 0, "sample"), ("boat_novice_lqr", 400, 64, 3, 2, "tree")])
def test_loopback_rank_matches_single_engine(name, nodes, wave, world, rank, scheme):
    from lqrrt_amd.parallel import NativeComm
    _, ref = _make(name, nodes + wave + 8, wave)
    rs = ref.extend(wave, node_limit=nodes)
    _, eng = _make(name, nodes + wave + 8, wave)
    comm = NativeComm(rank, world)
    st = eng.extend_sharded(comm, scheme, wave, node_limit=nodes)
    assert (st.attempts, st.accepted, st.goal_hits) == (rs.attempts, rs.accepted, rs.goal_hits)
    _same_tree(eng, ref)
    comm.close()


def test_loopback_tail_overflow_resteers(monkeypatch):
    """LQRRT_SHARD_TAIL=0: only one edge fits into a block's tail, every other accepted sample arrives without its edge and is
    re-steered in the first round -- by the receivers AND by its owner (whose own record is complete).  Same tree as the single
    engine, edges included, for every emulated rank.  (WHICH edge fits is decided by an atomic in the producing launch, so two
    loopback processes that each recompute all blocks may see different overflow sets; that every rank of ONE world runs the same
    rounds -- ADVICE r03 -- is asserted where the ranks share their blocks: the two-process test below, run with the tail at 0.)"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import test_native_sharded_gpu as t
        from lqrrt_amd.parallel import NativeComm
        _, ref = t._make("boat_advanced", 1300, 256); ref.extend(256, node_limit=1000)
        for rank in range(4):
            _, eng = t._make("boat_advanced", 1300, 256)
            c = NativeComm(rank, 4); st = eng.extend_sharded(c, "sample", 256, node_limit=1000)
            t._same_tree(eng, ref)
            assert st.resteers > 0
            c.close()
        print("OK")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, LQRRT_SHARD_TAIL="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_loopback_synchronous_mode_sharded():
    """The synchronous wave mode has a sample-sharded path too: every sample sees the wave-start snapshot, nothing is repaired,
    so the shards are independent up to the commit."""
    from lqrrt_amd.parallel import NativeComm
    _, ref = _make("boat_advanced", 3000, 512, sync=True)
    ref.extend(512, max_attempts=8 * 512)
    _, eng = _make("boat_advanced", 3000, 512, sync=True)
    comm = NativeComm(2, 4)
    eng.extend_sharded(comm, "sample", 512, max_attempts=8 * 512)
    _same_tree(eng, ref)


def test_rccl_world_of_one_both_schemes():
    """The real communicator: ncclGetUniqueId / ncclCommInitRank / ncclAllGather resolved from the librccl.so in this process."""
    import torch
    from lqrrt_amd.parallel import NativeComm

    class _Solo(object):                         # the 128-byte id has nobody to travel to
        def get_backend(self):
            return "solo"

        def broadcast(self, t, src):
            return None
    torch.cuda.init()
    for scheme in ("sample", "tree"):
        _, ref = _make("boat_intermediate", 900, 128)
        ref.extend(128, node_limit=600)
        _, eng = _make("boat_intermediate", 900, 128)
        comm = NativeComm(0, 1, device=0, dist=_Solo())
        eng.extend_sharded(comm, scheme, 128, node_limit=600)
        torch.cuda.synchronize()
        _same_tree(eng, ref)
        comm.close()


_TWO_RANK_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import test_native_sharded_gpu as t
from lqrrt_amd.parallel import NativeComm
rank, world = int(sys.argv[1]), int(sys.argv[2])
uid = bytes.fromhex(os.environ["LQRRT_TEST_UID"])
comm = NativeComm(rank, world, device=0, uid=uid)          # lqrrt_comm_create -> ncclCommInitRank of the library LQRRT_RCCL names
out = []
for name, nodes, wave, scheme, sync in (("boat_advanced", 1500, 256, "sample", False), ("car", 900, 256, "sample", False),
                                         ("boat_advanced", 1200, 256, "tree", False), ("boat_advanced", 2500, 512, "sample", True)):
    _, ref = t._make(name, nodes + wave + 8, wave, sync=sync)
    kw = dict(max_attempts=6 * wave) if sync else dict(node_limit=nodes)
    rs = ref.extend(wave, **kw)
    _, eng = t._make(name, nodes + wave + 8, wave, sync=sync)
    st = eng.extend_sharded(comm, scheme, wave, **kw)
    assert (st.attempts, st.accepted, st.goal_hits) == (rs.attempts, rs.accepted, rs.goal_hits), (name, scheme)
    t._same_tree(eng, ref)
    out.append((st.waves, st.fix_rounds, st.resteers, st.attempts))
comm.close()
print("RANK", rank, "STATS", out, "OK")
"""


@pytest.mark.parametrize("tail", ["default", "0"])
def test_two_processes_through_the_collective_entry_point(tail):
    """Two ranks in two PROCESSES run lqrrt_engine_extend_sharded against each other on this one GPU: the library's RCCL entry
    points are resolved from tests/stub_rccl/libstub_rccl.so (LQRRT_RCCL), an all-gather over POSIX shared memory -- real RCCL
    refuses two ranks on one device, and the loopback communicator skips ncclAllGather altogether.  What only this test executes:
    the in-place all-gather offsets of a rank > 0, the per-rank tail cursor across waves, a rank that only ever sees the other
    rank's samples through the blocks, and the same W / round sequence on both sides (or the collective would not match up).
    Sample-sharded (fused rounds, car and boat), tree-sharded, synchronous mode; each rank == the single engine, bit for bit.
    tail = "0": LQRRT_SHARD_TAIL=0, nearly every accepted edge overflows its block's tail and is re-steered -- by the receiver and,
    since round 4, by the owner too, so that fix_rounds / resteers and with them the wave-size controller stay replicated."""
    import subprocess, sys
    import ctypes as C
    stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub_rccl", "libstub_rccl.so")
    if not os.path.exists(stub):
        pytest.fail("tests/stub_rccl/libstub_rccl.so is missing: __graft_entry__.build() compiles it")
    lib = C.CDLL(stub)
    uid = (C.c_char * 128)()
    assert lib.ncclGetUniqueId(uid) == 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _TWO_RANK_CHILD % dict(root=root, tests=os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LQRRT_RCCL=stub, LQRRT_TEST_UID=bytes(uid).hex())
    if tail != "default":
        env["LQRRT_SHARD_TAIL"] = tail
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0 and "OK" in o, o[-1500:] + e[-2500:]
    stats = [o.split("STATS")[1].split("OK")[0].strip() for _, o, _ in outs]
    assert stats[0] == stats[1], stats                # same waves / rounds / re-steers on both ranks


_TWO_DEVICE_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import torch
import test_native_sharded_gpu as t
from lqrrt_amd.parallel import NativeComm
rank, world = int(sys.argv[1]), int(sys.argv[2])
dev = rank if os.environ["LQRRT_TEST_PATH"] == "rccl" else 0
torch.cuda.set_device(dev)
uid = bytes.fromhex(os.environ["LQRRT_TEST_UID"])
comm = NativeComm(rank, world, device=dev, uid=uid)
out = []
for name, nodes, wave, scheme in (("car", 1200, 256, "sample"), ("double_integrator", 2500, 256, "tree"), ("boat_advanced", 1500, 256, "sample")):
    _, ref = t._make(name, nodes + wave + 8, wave, device=dev)
    rs = ref.extend(wave, node_limit=nodes)
    _, eng = t._make(name, nodes + wave + 8, wave, device=dev)
    st = eng.extend_sharded(comm, scheme, wave, node_limit=nodes)
    assert (st.attempts, st.accepted, st.goal_hits) == (rs.attempts, rs.accepted, rs.goal_hits), (name, scheme)
    t._same_tree(eng, ref)
    out.append((st.waves, st.fix_rounds, st.resteers, st.attempts))
comm.close()
print("RANK", rank, "DEVICE", dev, "STATS", out, "OK")
"""


def test_two_ranks_on_two_devices_through_real_rccl():
    """What an 8-GPU box would run, in its smallest form: two ranks in two processes on devices 0 and 1, the
    REAL communicator (ncclCommInitRank / ncclAllGather of the librccl PyTorch ships, resolved by the engine itself), sample-sharded
    car-1200 and boat-1500, tree-sharded double-integrator-2500 -- each rank's parents / states / gains / edges / ignore set / best plan
    equal to the single engine on its device, and waves / fix_rounds / resteers equal on both ranks (the wave-size controller, hence the
    all-gather sizes, must be replicated).  On a box with ONE device the same child runs both ranks on device 0 through the
    shared-memory double of librccl (tests/stub_rccl): never a skip; the path taken is printed."""
    import subprocess, sys
    import ctypes as C
    import torch
    from lqrrt_amd.parallel import NativeComm
    two = torch.cuda.device_count() >= 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LQRRT_TEST_PATH="rccl" if two else "stub")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if two:
        env.pop("LQRRT_RCCL", None)
        uid = NativeComm.unique_id()                              # ncclGetUniqueId of the real library, made here, carried by the environment
    else:
        stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub_rccl", "libstub_rccl.so")
        if not os.path.exists(stub):
            pytest.fail("tests/stub_rccl/libstub_rccl.so is missing: __graft_entry__.build() compiles it")
        lib = C.CDLL(stub)
        buf = (C.c_char * 128)()
        assert lib.ncclGetUniqueId(buf) == 0
        uid = bytes(buf)
        env["LQRRT_RCCL"] = stub
    env["LQRRT_TEST_UID"] = bytes(uid).hex()
    print("two-rank parity path: %s" % ("real RCCL, devices 0 and 1" if two else "one device: shared-memory double of librccl"))
    code = _TWO_DEVICE_CHILD % dict(root=root, tests=os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0 and "OK" in o, o[-1500:] + e[-2500:]
        print(o.strip().splitlines()[-1])
    stats = [o.split("STATS")[1].split("OK")[0].strip() for _, o, _ in outs]
    assert stats[0] == stats[1], stats


def test_legacy_repair_path_for_waves_beyond_256():
    """The repair path that is still shipped for waves beyond the fused rounds' 256 samples and for Riccati systems -- k_decide +
    in-wave scan of the records + listed re-steers -- in the single-engine loop and in a gathered wave (with the
    exact-mode cap of 256 in force nothing reached it any more).  LQRRT_EXACT_WAVE_MAX=1024 lifts the cap, LQRRT_MATRIX_MAX_W=0
    takes the in-wave matrix (and with it the fused rounds) away from the smaller waves too, so every wave of the run uses it."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np
        import test_native_sharded_gpu as t
        import coracle, lqrrt_amd
        from lqrrt_amd.parallel import NativeComm
        s, ref = t._make("boat_advanced", 3200, 1024)
        rs = ref.extend(1024, node_limit=2000)
        assert rs.fix_rounds > 50 and rs.resteers > 200, (rs.fix_rounds, rs.resteers)
        o = coracle.make(s, 3200, seed=1); o.extend(max_nodes=2000)
        assert ref.size == o.size and np.array_equal(ref.parents(), o.parents()) and np.array_equal(ref.states(), o.states())
        _, eng = t._make("boat_advanced", 3200, 1024)
        c = NativeComm(1, 4); eng.extend_sharded(c, "sample", 1024, node_limit=2000)
        t._same_tree(eng, ref); print("OK", rs.attempts, rs.waves)
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
           os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LQRRT_EXACT_WAVE_MAX="1024", LQRRT_MATRIX_MAX_W="0"), capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
