"""
world_size-2 and -4 gloo tests of the sample-sharded and the tree-sharded wave logic (lqrrt_amd/parallel.py) on CPU.

There is no GPU here, so the engine is replaced by a stand-in that fills its slice of the record
buffer with a deterministic function of (sample index, tree size) and "commits" by hashing the
gathered records.  What is verified: shard bounds cover the wave exactly once, the all-gather puts
every rank's rows where the commit expects them, and all ranks end with identical state -- the
properties the bit-identical replicas rely on.
"""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def torch_from(a):
    import torch
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


class FakeStats(object):
    def __init__(self, attempts, accepted):
        self.attempts, self.accepted = attempts, accepted


class FakeEngine(object):
    """Mimics Engine.wave_speculate / wave_commit / size on a torch CPU tensor, with the boat's real record width
    (csrc make_layout: 4 + n + 2*nw + m*n + H*(n+m) = 210 doubles for n=6, m=3, nw=1, H=20)."""
    R = 210

    def __init__(self, max_wave):
        import torch
        self.rec = torch.zeros((max_wave, self.R), dtype=torch.float64)
        self.size = 600
        self.cursor = 0
        self.digest = hashlib.sha1()
        self.device = 0
        self.max_wave = max_wave

    def wave_speculate(self, W, lo, hi):
        self.rec[:W] = -1.0                                   # poison everything this rank does not own
        for t in range(lo, hi):
            k = self.cursor + t
            self.rec[t] = torch_from(np.float64(k) * 10.0 + np.arange(self.R) + self.size * 1e-3)

    def wave_commit(self, W, max_commit, node_limit, pruning=True):
        rows = self.rec[:W].numpy()
        assert (rows[:, 0] >= 0).all(), "a slice was not gathered"
        want = np.array([(self.cursor + t) * 10.0 + self.size * 1e-3 for t in range(W)])
        np.testing.assert_array_equal(rows[:, 0], want)
        self.digest.update(rows.tobytes())
        C = min(W, max_commit)
        self.cursor += C
        acc = C // 3
        self.size += acc
        return FakeStats(C, acc)


class FakeTreeEngine(object):
    """Stand-in for the tree-sharded wave: node i has cost f(sample, i); every rank must end up steering each sample
    from the global arg-min over ALL nodes although it only scanned its own range."""

    def __init__(self, max_wave, size=1000):
        self.max_wave, self.size, self.cursor, self.device = max_wave, size, 0, 0
        self.digest = hashlib.sha1()
        self.seen = []

    @staticmethod
    def cost(k, i):
        return float((k * 7919 + i * 104729) % 1000003)

    def wave_scan_nodes(self, W, lo, hi, best_ptr):
        import ctypes
        assert lo % 64 == 0 and 0 <= lo <= hi <= self.size
        out = np.frombuffer((ctypes.c_double * (2 * W)).from_address(best_ptr), dtype=np.float64).reshape(W, 2)
        for t in range(W):
            k = self.cursor + t
            c = [self.cost(k, i) for i in range(lo, hi)]
            j = int(np.argmin(c)) if c else -1
            out[t] = (c[j], lo + j) if c else (0.0, -1.0)

    def wave_steer_candidates(self, W, parts, best_ptr):
        import ctypes
        buf = np.frombuffer((ctypes.c_double * (2 * W * parts)).from_address(best_ptr), dtype=np.float64).reshape(parts, W, 2)
        self.parents = []
        for t in range(W):
            cand = [(buf[p, t, 0], int(buf[p, t, 1])) for p in range(parts) if buf[p, t, 1] >= 0]
            self.parents.append(min(cand)[1])
        want = [int(np.argmin([self.cost(self.cursor + t, i) for i in range(self.size)])) for t in range(W)]
        assert self.parents == want, "the gathered candidates do not reproduce the global arg-min"

    def wave_commit(self, W, max_commit, node_limit, pruning=True):
        self.digest.update(np.array(self.parents, dtype=np.int64).tobytes())
        C = min(W, max_commit)
        self.cursor += C
        self.size += C // 3
        return FakeStats(C, C // 3)


def _tree_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lqrrt_amd.parallel import TreeShardedWave
    eng = FakeTreeEngine(64)
    sw = TreeShardedWave(eng, dist, rank, world, best=torch.zeros((world, 64, 2), dtype=torch.float64))
    total = 0
    for want in (64, 40, 7, 64):
        total += sw.wave(want, max_commit=want).attempts
    out.put((rank, total, eng.size, eng.digest.hexdigest()))
    dist.destroy_process_group()


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lqrrt_amd.parallel import ShardedWave
    eng = FakeEngine(256)
    sw = ShardedWave(eng, dist, rank, world, records=eng.rec)
    total = 0
    for want in (256, 200, 77, 1, 130):
        st = sw.wave(want, max_commit=want)
        total += st.attempts
    out.put((rank, total, eng.size, eng.digest.hexdigest()))
    dist.destroy_process_group()


def _comm_worker(rank, world, port, out):
    """The communicator handshake of the native sharded loop: rank 0's RCCL unique id reaches every rank through
    torch.distributed (gloo here); without a GPU lqrrt_comm_create must then refuse with LQRRT_E_NODEVICE, not crash."""
    import hashlib
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lqrrt_amd import _native as nat
    from lqrrt_amd import parallel
    import ctypes as C
    uid = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        nat.check(nat.lib().lqrrt_comm_unique_id(uid.ctypes.data_as(C.c_void_p)))
    t = torch.from_numpy(uid)
    dist.broadcast(t, 0)
    digest = hashlib.sha1(t.numpy().tobytes()).hexdigest()
    refused = None
    try:
        parallel.NativeComm(rank, world, device=0, dist=dist)
        refused = torch.cuda.is_available() and "created"
    except nat.NativeError as ex:
        refused = ex.args[0] if ex.args else True
    loop = parallel.NativeComm(rank, world)         # the loopback double needs no device
    loop.close()
    out.put((rank, int(uid.any() or rank != 0), digest, str(refused)))
    dist.destroy_process_group()


def test_native_comm_handshake_gloo():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 2
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == 1, "rank 0 got no unique id from librccl"
    assert res[0][2] == res[1][2], "the ranks hold different ids"
    if not torch.cuda.is_available():
        assert all("-3" in r[3] or "not available" in r[3] for r in res), res      # LQRRT_E_NODEVICE


def test_shard_bounds_cover_wave():
    from lqrrt_amd.parallel import pick_wave, shard_bounds
    for W in (1, 7, 64, 100, 1024):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                per, lo, hi = shard_bounds(W, r, world)
                seen.extend(range(lo, hi))
                assert hi - lo <= per
            assert seen == list(range(W))
    assert pick_wave(10000, 1024) == 1024 and pick_wave(600, 1024) == 64 and pick_wave(10, 1024) == 8


def test_node_ranges_cover_tree():
    from lqrrt_amd.parallel import node_range
    for size in (1, 63, 64, 65, 1000, 50000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = node_range(size, r, world)
                assert lo % 64 == 0 or lo == size
                seen.extend(range(lo, hi))
            assert seen == list(range(size))


@pytest.mark.parametrize("world,worker", [(2, "_worker"), (4, "_worker"), (4, "_tree_worker")])
def test_sharded_wave_gloo(world, worker):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    import socket
    with socket.socket() as sk:                       # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=globals()[worker], args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    # generous: a spawned interpreter has to import torch, which takes minutes in a cold container
    res = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1:] == res[0][1:] for r in res), "replicas diverged: %r" % (res,)
