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


def _make(name, cap, wave, seed=1, sync=False, **sys_kw):
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.DoubleIntegrator(**sys_kw) if name == "double_integrator" else lqrrt_amd.systems.SYSTEMS[name](0)
    eng = Engine(s, capacity=cap, max_wave=wave)
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
    ("double_integrator", 1200, 512, 4, 2, "tree"), ("double_integrator", 1200, 256, 2, 1, "sample")])
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
    """LQRRT_SHARD_TAIL=0: only one edge fits into a block's tail, every other accepted sample of another rank arrives
    without its edge and is re-steered by the receiver -- same tree, edges included."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import test_native_sharded_gpu as t
        from lqrrt_amd.parallel import NativeComm
        _, ref = t._make("boat_advanced", 1300, 256); ref.extend(256, node_limit=1000)
        _, eng = t._make("boat_advanced", 1300, 256)
        c = NativeComm(1, 4); st = eng.extend_sharded(c, "sample", 256, node_limit=1000)
        t._same_tree(eng, ref); print("resteers", st.resteers, "OK")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, LQRRT_SHARD_TAIL="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
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
