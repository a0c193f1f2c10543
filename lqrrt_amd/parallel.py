"""
Sample-sharded waves over the GPUs of one node (one process per GPU, torch.distributed with
backend "nccl" = RCCL over xGMI).

Within a wave, given the same tree snapshot, the W sample -> nearest-neighbour -> steer problems
are independent.  Every rank holds a full replica of the tree and the same
sample stream (same MT19937 state).  Rank g evaluates samples [g*W/G, (g+1)*W/G) speculatively
(lqrrt_wave_speculate), the fixed-size per-sample records are exchanged with ONE all-gather,
and every rank then runs the same deterministic exact-mode commit (lqrrt_wave_commit), so the
replicas stay bit-identical without further traffic.  The collective is the only exchange step
of the path; payload = W * record_doubles * 8 B (boat: 1712 B/record, 1.75 MB per 1024-wave),
latency- rather than bandwidth-bound on 7 x 153 GB/s xGMI links.

ShardedWave is the sample-sharded scheme above.  TreeShardedWave is SURVEY 8(e)'s alternative for large trees
(BASELINE config 5, where the nearest-neighbour scan over 50k nodes dominates a wave): every rank scans only its 1/G of
the NODES for all W samples, the per-sample (cost, id) candidates -- 16*W bytes per rank -- are all-gathered, and every
rank then steers and commits the whole wave itself.  No records travel at all; the replicas stay bit-identical because
the winner by (cost, id) over ascending node ranges is exactly the node a single scan returns.

Both classes are engine-agnostic (anything with the wave_* methods and the buffers), which is how
tests/test_parallel_cpu.py drives them with gloo on CPU.
"""
import ctypes as C

import numpy as np


class NativeComm(object):
    """Communicator of the native sharded loop (include/lqrrt_hip.h lqrrt_comm_*).  `dist` given: an RCCL communicator of its
    own for this world -- rank 0 creates the unique id, the 128 bytes travel through torch.distributed (any backend), every
    rank joins.  `uid` given (128 bytes from lqrrt_comm_unique_id / NativeComm.unique_id() on rank 0, carried by the caller's
    own means): the same without torch.distributed.  Neither: the loopback double (one process plays `rank` of `world`; tests)."""

    @staticmethod
    def unique_id():
        from . import _native as nat
        uid = np.zeros(128, dtype=np.uint8)
        nat.check(nat.lib().lqrrt_comm_unique_id(uid.ctypes.data_as(C.c_void_p)))
        return uid.tobytes()

    def __init__(self, rank, world, device=0, dist=None, uid=None):
        from . import _native as nat
        self._nat = nat
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        if uid is not None:
            uid = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
            if uid.size != 128:
                raise ValueError("uid must be the 128 bytes of lqrrt_comm_unique_id")
            nat.check(nat.lib().lqrrt_comm_create(uid.ctypes.data_as(C.c_void_p), self.rank, self.world, int(device), C.byref(h)))
        elif dist is None:
            nat.check(nat.lib().lqrrt_comm_create_loopback(self.rank, self.world, C.byref(h)))
        else:
            import torch
            uid = np.zeros(128, dtype=np.uint8)
            if self.rank == 0:
                nat.check(nat.lib().lqrrt_comm_unique_id(uid.ctypes.data_as(C.c_void_p)))
            on_gpu = hasattr(dist, "get_backend") and dist.get_backend() == "nccl"
            t = torch.from_numpy(uid)
            t = t.cuda(device) if on_gpu else t
            dist.broadcast(t, 0)
            uid = np.ascontiguousarray(t.cpu().numpy(), dtype=np.uint8)
            nat.check(nat.lib().lqrrt_comm_create(uid.ctypes.data_as(C.c_void_p), self.rank, self.world, int(device), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self._nat.lib().lqrrt_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeSharded(object):
    """Sharded waves without a host language in the loop: engine.extend_sharded runs speculate -> ncclAllGather -> commit on one
    stream for as many waves as the call covers.

    NOT the one-wave-per-call interface of ShardedWave / TreeShardedWave: `extend_to(wave_cap, max_attempts, ...)` commits up to
    `max_attempts` attempts over however many waves that takes (`wave_cap` bounds a wave) and returns their aggregated
    ExtendStats (`waves` says how many).  Callers that need a per-wave hook -- a rewind window, stop-on-goal handling of their own
    -- pass max_attempts <= wave_cap or use the Python classes below.  `wave` is kept as an alias for bench.py's loop, which
    treats all three classes as "commit up to this many attempts"."""

    def __init__(self, engine, comm, scheme="sample"):
        self.e, self.comm, self.scheme = engine, comm, scheme

    def extend_to(self, wave_cap, max_attempts, node_limit=-1, pruning=True):
        if max_attempts is None or max_attempts < 0:
            if node_limit is None or node_limit < 0:
                raise ValueError("NativeSharded: unbounded call (max_attempts < 0 and no node_limit): it would run until the tree "
                                 "capacity is exhausted; bound it or use extend(wave, until_size=...)")
            max_attempts = -1
        return self.e.extend_sharded(self.comm, self.scheme, wave_cap, max_attempts=max_attempts, node_limit=node_limit, pruning=pruning)

    def wave(self, want, max_commit, node_limit=-1, pruning=True):
        return self.extend_to(want, max_commit, node_limit=node_limit, pruning=pruning)

    def extend(self, wave, **kw):
        return self.e.extend_sharded(self.comm, self.scheme, wave, **kw)


class _DevBlob(object):
    """Exposes engine-owned HBM through the CUDA array interface so torch can view it (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def records_tensor(engine):
    """torch view [max_wave][record_doubles] of the engine's wave record buffer."""
    import torch
    R = int(engine.record_layout()[0])
    blob = _DevBlob(engine.wave_records_ptr(), (engine.max_wave, R))
    return torch.as_tensor(blob, device="cuda:%d" % engine.device)


def pick_wave(tree_size, wave_cap):
    """Open-loop part of the engine's wave-size policy (W ~ N/6 keeps in-wave conflicts rare); the
    engine's lqrrt_wave_suggest adds feedback from the last commits and is used when available."""
    W = max(tree_size // 6, 8)
    W = min(W, wave_cap)
    if W >= 64:
        W = (W // 64) * 64
    return W


def shard_bounds(W, rank, world):
    per = (W + world - 1) // world
    lo = min(W, rank * per)
    hi = min(W, lo + per)
    return per, lo, hi


class ShardedWave(object):
    def __init__(self, engine, dist, rank, world, records=None):
        self.e, self.dist, self.rank, self.world = engine, dist, rank, world
        self.rec = records_tensor(engine) if records is None else records
        self.max_wave = self.rec.shape[0]
        self.in_place = dist.get_backend() == "nccl" if hasattr(dist, "get_backend") else False

    def wave(self, want, max_commit, node_limit=-1, pruning=True):
        """One wave of up to `want` samples; returns the commit's ExtendStats."""
        suggest = getattr(self.e, "wave_suggest", None)
        W = min(suggest(self.max_wave) if suggest else pick_wave(self.e.size, self.max_wave), want)
        per, lo, hi = shard_bounds(W, self.rank, self.world)
        while per * self.world > self.max_wave:        # gather buffer must hold world * per rows
            W -= 1
            per, lo, hi = shard_bounds(W, self.rank, self.world)
        self.e.wave_speculate(W, lo, hi)
        full = self.rec[: per * self.world]
        send = self.rec[self.rank * per: (self.rank + 1) * per]
        _gather(self, full.view(-1), send.view(-1))
        return self.e.wave_commit(W, max_commit, node_limit, pruning)


def _gather(sw, out, chunk):
    """all_gather_into_tensor with `chunk` = this rank's own slice of `out`.  RCCL gathers IN PLACE in that case (no
    staging copy); backends or versions that refuse aliased buffers get a clone, once and for all."""
    if sw.in_place:
        try:
            sw.dist.all_gather_into_tensor(out, chunk)
            return
        except (RuntimeError, ValueError):
            sw.in_place = False
    sw.dist.all_gather_into_tensor(out, chunk.clone())


def node_range(size, rank, world):
    """Rank `rank`'s slice [lo, hi) of `size` nodes, boundaries on multiples of 64 (aligned scalar loads of the scan)."""
    per = ((size + world - 1) // world + 63) // 64 * 64
    lo = min(size, rank * per)
    return lo, min(size, lo + per)


class TreeShardedWave(object):
    def __init__(self, engine, dist, rank, world, best=None):
        self.e, self.dist, self.rank, self.world = engine, dist, rank, world
        if best is None:
            import torch
            best = torch.empty((world, engine.max_wave, 2), dtype=torch.float64, device="cuda:%d" % engine.device)
        self.best = best                          # [world][max_wave][2] (cost, id) candidates; rank r's chunk is row r
        self.max_wave = best.shape[1]
        self.in_place = dist.get_backend() == "nccl" if hasattr(dist, "get_backend") else False

    def wave(self, want, max_commit, node_limit=-1, pruning=True):
        """One wave of up to `want` samples, all of them steered and committed on every rank."""
        suggest = getattr(self.e, "wave_suggest", None)
        W = min(suggest(self.max_wave) if suggest else pick_wave(self.e.size, self.max_wave), want)
        lo, hi = node_range(self.e.size, self.rank, self.world)
        buf = self.best[:, :W, :] if W == self.max_wave else self.best.view(-1)[: self.world * W * 2].view(self.world, W, 2)
        mine = buf[self.rank]
        self.e.wave_scan_nodes(W, lo, hi, mine.data_ptr())
        _gather(self, buf.view(-1), mine.view(-1))                          # 16 * W bytes per rank: the one collective
        self.e.wave_steer_candidates(W, self.world, buf.data_ptr())
        return self.e.wave_commit(W, max_commit, node_limit, pruning)
