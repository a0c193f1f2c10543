"""
Thin object wrapper over the C ABI (include/lqrrt_hip.h).  Device memory for the operator
inputs/outputs is owned by PyTorch-ROCm tensors whose data_ptr() is handed to the HIP
library; launches go onto torch's current HIP stream.
"""
import ctypes as C

import numpy as np

from . import _native as nat


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise nat.NativeError(nat.E_NODEVICE, "torch sees no HIP device: lqrrt_amd needs an MI355X (no CPU fallback)")
    return torch


class Engine(object):
    """One native engine handle = one problem on one GPU."""

    def __init__(self, system, capacity, max_wave=1024, device=0):
        nat.require_device()
        self.system = system
        self.n, self.m = system.nstates, system.ncontrols
        self.device = device
        self.capacity = int(capacity)
        self.max_wave = int(max_wave)
        desc, keep = system.desc()
        self._keep = keep
        h = C.c_void_p()
        nat.check(nat.lib().lqrrt_engine_create(C.byref(desc), device, self.capacity, self.max_wave, C.byref(h)))
        self.h = h
        self._geometry_revision = getattr(system, "revision", 0)
        if system.S is not None:
            S = nat.as_f64(system.S, (self.n, self.n))
            nat.check(nat.lib().lqrrt_engine_set_dense_S(self.h, nat.ptr(S)))
        self.horizon_iters = None
        self.generation = 0           # bumped whenever the tree is replaced (reset / load): Tree views check it
        self.epoch = 0                # bumped whenever nodes can change behind a view's back (truncate / rewind / re-layout)

    def _stream(self):
        """torch's current HIP stream on THIS engine's device."""
        return nat.current_stream(self.device)

    def set_wave_mode(self, mode):
        """'exact' (default: the reference's sequential result) or 'synchronous' (all samples of a wave see the
        wave-start snapshot; fixed wave size; parity target oracle/lqrrt_oracle.c orc_extend_sync)."""
        modes = {"exact": 0, "synchronous": 1}
        if mode not in modes:
            raise ValueError("wave mode must be 'exact' or 'synchronous'")
        nat.check(nat.lib().lqrrt_engine_set_wave_mode(self.h, modes[mode]))
        self.wave_mode = mode

    def set_cu_mask(self, xcds=None, mask=None, n_cus=256):
        """Runs this engine's native loops on a stream restricted to some of the GPU's compute units (include/lqrrt_hip.h
        lqrrt_engine_set_cu_mask).  `xcds`: iterable of XCD numbers 0..7 (all CUs of those XCDs), or `mask`: the raw words;
        neither = the whole chip again.  Speed only."""
        if mask is None and xcds is not None:
            want = set(int(x) & 7 for x in xcds)
            words = np.zeros((n_cus + 31) // 32, dtype=np.uint32)
            for b in range(n_cus):
                if (b & 7) in want:
                    words[b >> 5] |= np.uint32(1 << (b & 31))
            mask = words
        if mask is None:
            nat.check(nat.lib().lqrrt_engine_set_cu_mask(self.h, None, 0))
            return
        mask = np.ascontiguousarray(mask, dtype=np.uint32)
        nat.check(nat.lib().lqrrt_engine_set_cu_mask(self.h, nat.ptr(mask), int(mask.size)))

    def footprint(self):
        """dict(device_bytes, pinned_bytes): what this engine holds in HBM and in page-locked host memory (lqrrt_engine_footprint)."""
        d, h = C.c_int64(), C.c_int64()
        nat.check(nat.lib().lqrrt_engine_footprint(self.h, C.byref(d), C.byref(h)))
        return dict(device_bytes=d.value, pinned_bytes=h.value)

    def sync_geometry(self):
        """Re-uploads parameters, hull points, obstacles and occupancy grid if the system object changed since
        this engine last saw it (system.revision; e.g. set_occupancy_grid between two plans)."""
        rev = getattr(self.system, "revision", 0)
        if rev != self._geometry_revision:
            desc, keep = self.system.desc()
            nat.check(nat.lib().lqrrt_engine_set_geometry(self.h, C.byref(desc), self._stream()))
            self._keep = keep
            self._geometry_revision = rev
            return True
        return False

    def close(self):
        if getattr(self, "h", None):
            nat.lib().lqrrt_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------------
    def set_resolution(self, dt, FPR, horizon_iters, error_tol, goal, goal_buffer, adaptive=False, hspan_min=1,
                       horizon_iters_state=1):
        r = nat.Resolution()
        r.dt, r.FPR, r.horizon_iters = float(dt), float(FPR), int(horizon_iters)
        r.adaptive, r.hspan_min, r.horizon_iters_state = (1 if adaptive else 0), int(hspan_min), int(horizon_iters_state)
        tol = np.broadcast_to(np.asarray(error_tol, dtype=np.float64), (self.n,))
        for i in range(nat.MAX_STATES):
            r.error_tol[i] = tol[i] if i < self.n else np.inf
            r.goal[i] = 0.0
            r.goal_lo[i], r.goal_hi[i] = -np.inf, np.inf
        r.has_goal = 0
        if goal is not None:
            g = np.asarray(goal, dtype=np.float64)
            b = np.asarray(goal_buffer, dtype=np.float64)
            r.has_goal = 1
            for i in range(self.n):
                r.goal[i] = g[i]
                r.goal_lo[i] = g[i] - b[i]        # planner.py:482-484
                r.goal_hi[i] = g[i] + b[i]
        nat.check(nat.lib().lqrrt_engine_set_resolution(self.h, C.byref(r)))
        if self.horizon_iters is not None and int(horizon_iters) != self.horizon_iters:
            self.generation += 1      # the edge pools were re-laid out and the tree emptied: bound Tree views are dead
        self.horizon_iters = int(horizon_iters)
        self.epoch += 1

    def horizon_iters_state(self):
        return nat.check(nat.lib().lqrrt_engine_horizon_iters(self.h))

    def set_sampler(self, centers, spans, goal_bias, tries_limit):
        s = nat.SamplerDesc()
        for i in range(self.n):
            s.centers[i], s.spans[i], s.goal_bias[i] = centers[i], spans[i], goal_bias[i]
        s.tries_limit = int(tries_limit)
        nat.check(nat.lib().lqrrt_engine_set_sampler(self.h, C.byref(s)))

    def set_mt19937(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        nat.check(nat.lib().lqrrt_engine_set_mt19937(self.h, nat.ptr(key), int(pos)))

    def get_mt19937(self):
        key = np.zeros(624, dtype=np.uint32)
        pos = C.c_int()
        nat.check(nat.lib().lqrrt_engine_get_mt19937(self.h, nat.ptr(key), C.byref(pos)))
        return key, pos.value

    def seed_from_numpy_global(self):
        st = np.random.get_state()
        self.set_mt19937(st[1], st[2])

    def sync_numpy_global(self):
        """Leaves np.random exactly where the reference's sampler would have left it."""
        key, pos = self.get_mt19937()
        st = np.random.get_state()
        np.random.set_state((st[0], key, pos, 0, 0.0))

    # -- tree -----------------------------------------------------------------------------------
    def tree_reset(self, x0):
        x0 = nat.as_f64(x0, (self.n,))
        nat.check(nat.lib().lqrrt_tree_reset(self.h, nat.ptr(x0), self._stream()))
        self.generation += 1

    def tree_load(self, states, K, pID, edge_len=None, xedge=None, uedge=None, ignored=None):
        """Puts an existing tree on the device (lqrrt_tree_load): states (N, n), K (N, m, n), pID (N,), optionally the
        edge lengths (N,) with the packed edges (sum(len), n) / (sum(len), m) and the ignore flags (N,)."""
        states = nat.as_f64(states)
        N = len(states)
        states = nat.as_f64(states, (N, self.n))
        K = nat.as_f64(K, (N, self.m, self.n))
        pID = np.ascontiguousarray(pID, dtype=np.int32)
        if pID.shape != (N,):
            raise ValueError("expected %d parent IDs" % N)
        keep = [states, K, pID]

        def opt(a, dtype, shape=None):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dtype)
            if shape is not None and a.shape != shape:
                raise ValueError("expected array of shape %r, got %r" % (shape, a.shape))
            keep.append(a)
            return nat.ptr(a)
        el = opt(edge_len, np.int32, (N,))
        rows = int(np.sum(edge_len)) if edge_len is not None else N
        nat.check(nat.lib().lqrrt_tree_load(self.h, N, nat.ptr(states), nat.ptr(K), nat.ptr(pID), el,
                                            opt(xedge, np.float64, (rows, self.n)), opt(uedge, np.float64, (rows, self.m)),
                                            opt(ignored, np.uint8, (N,)), self._stream()))
        self.generation += 1

    def tree_truncate(self, size):
        nat.check(nat.lib().lqrrt_tree_truncate(self.h, int(size)))
        self.epoch += 1

    def set_ignored(self, flags, first=0):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        nat.check(nat.lib().lqrrt_tree_set_ignored(self.h, int(first), len(flags), nat.ptr(flags)))

    def edges(self, first=0, count=None, pinned=False):
        """(x [count][H][n], u [count][H][m], len [count]) in three copies instead of one per node.  pinned: the two big arrays
        are views of page-locked host memory (torch's caching host allocator) -- a copy out of HBM at PCIe speed instead of the
        ~0.7 GB/s a pageable destination gets; for snapshots of whole trees (Tree._detach: 18 MB per 12k nodes)."""
        count = self.size - first if count is None else count
        H = max(self.horizon_iters or 1, 1)
        if pinned:
            torch = _torch()
            x = torch.empty((count, H, self.n), dtype=torch.float64, pin_memory=True).numpy()
            u = torch.empty((count, H, self.m), dtype=torch.float64, pin_memory=True).numpy()
        else:
            x = np.empty((count, H, self.n))
            u = np.empty((count, H, self.m))
        nat.check(nat.lib().lqrrt_tree_get_edges(self.h, int(first), int(count), nat.ptr(x), nat.ptr(u)))
        return x, u, self.edge_lengths(first, count)

    def tree_mark(self):
        nat.check(nat.lib().lqrrt_tree_mark(self.h))

    def set_rewind_above(self, size):
        """Measurement aid: extend_multi rewinds this engine to its mark whenever a wave would begin above `size` nodes (0: off)."""
        nat.check(nat.lib().lqrrt_tree_set_rewind_above(self.h, int(size)))

    def tree_rewind(self):
        nat.check(nat.lib().lqrrt_tree_rewind(self.h))
        self.epoch += 1

    @property
    def size(self):
        return nat.check(nat.lib().lqrrt_tree_size(self.h))

    def states(self, first=0, count=None):
        count = self.size - first if count is None else count
        out = np.empty((count, self.n))
        nat.check(nat.lib().lqrrt_tree_get_states(self.h, first, count, nat.ptr(out)))
        return out

    def gains(self, first=0, count=None):
        count = self.size - first if count is None else count
        out = np.empty((count, self.m, self.n))
        nat.check(nat.lib().lqrrt_tree_get_gains(self.h, first, count, nat.ptr(out)))
        return out

    def parents(self, first=0, count=None):
        count = self.size - first if count is None else count
        out = np.empty(count, dtype=np.int32)
        nat.check(nat.lib().lqrrt_tree_get_parents(self.h, first, count, nat.ptr(out)))
        return out

    def edge_lengths(self, first=0, count=None):
        count = self.size - first if count is None else count
        out = np.empty(count, dtype=np.int32)
        nat.check(nat.lib().lqrrt_tree_get_edge_lengths(self.h, first, count, nat.ptr(out)))
        return out

    def edge(self, ID):
        H = max(self.horizon_iters or 1, 1)
        x = np.empty((H, self.n))
        u = np.empty((H, self.m))
        ln = nat.check(nat.lib().lqrrt_tree_get_edge(self.h, int(ID), nat.ptr(x), nat.ptr(u)))
        return x[:ln].copy(), u[:ln].copy()

    def ignored(self, first=0, count=None):
        count = self.size - first if count is None else count
        out = np.empty(count, dtype=np.uint8)
        nat.check(nat.lib().lqrrt_tree_get_ignored(self.h, first, count, nat.ptr(out)))
        return out.astype(bool)

    def climb(self, ID):
        """Node ids from the seed down to ID (tree.py:100-117), from the engine's host mirror of the parents: no device access."""
        cap = 4096
        while True:
            out = np.empty(cap, dtype=np.int32)
            rc = nat.lib().lqrrt_tree_climb(self.h, int(ID), nat.ptr(out), cap)
            if rc == nat.E_CAPACITY and cap < self.size + 1:
                cap = min(4 * cap, self.size + 1)
                continue
            return out[:nat.check(rc)].tolist()

    def edges_of(self, ids):
        """[(x [len][n], u [len][m])] of the listed nodes: one gather on the device, two copies out (lqrrt_tree_get_edges_of)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        H = max(self.horizon_iters or 1, 1)
        x = np.empty((len(ids), H, self.n))
        u = np.empty((len(ids), H, self.m))
        ln = np.empty(len(ids), dtype=np.int32)
        nat.check(nat.lib().lqrrt_tree_get_edges_of(self.h, nat.ptr(ids), len(ids), nat.ptr(x), nat.ptr(u), nat.ptr(ln)))
        return [(x[k, :ln[k]].copy(), u[k, :ln[k]].copy()) for k in range(len(ids))]

    # -- batched operators (host arrays in, host arrays out; device memory via torch) ----------
    def _dev(self, a, shape):
        torch = _torch()
        a = nat.as_f64(a, shape)
        return torch.from_numpy(a).to("cuda:%d" % self.device)

    def feasible_batch(self, x, u=None):
        torch = _torch()
        B = len(x)
        dx = self._dev(x, (B, self.n))
        du = self._dev(u, (B, self.m)) if u is not None else None
        ok = torch.empty(B, dtype=torch.uint8, device=dx.device)
        nat.check(nat.lib().lqrrt_feasible_batch(self.h, dx.data_ptr(), du.data_ptr() if du is not None else None,
                                                 B, ok.data_ptr(), self._stream()))
        return ok.cpu().numpy().astype(bool)

    def dynamics_batch(self, x, u):
        torch = _torch()
        B = len(x)
        dx, du = self._dev(x, (B, self.n)), self._dev(u, (B, self.m))
        out = torch.empty_like(dx)
        nat.check(nat.lib().lqrrt_dynamics_batch(self.h, dx.data_ptr(), du.data_ptr(), B, out.data_ptr(), self._stream()))
        return out.cpu().numpy()

    def gain_batch(self, x, u=None):
        torch = _torch()
        B = len(x)
        dx = self._dev(x, (B, self.n))
        du = self._dev(u, (B, self.m)) if u is not None else None
        K = torch.empty((B, self.m, self.n), dtype=torch.float64, device=dx.device)
        nat.check(nat.lib().lqrrt_gain_batch(self.h, dx.data_ptr(), du.data_ptr() if du is not None else None,
                                             B, K.data_ptr(), self._stream()))
        return K.cpu().numpy()

    def erf_batch(self, xg, x):
        torch = _torch()
        B = len(x)
        dg, dx = self._dev(xg, (B, self.n)), self._dev(x, (B, self.n))
        e = torch.empty_like(dx)
        nat.check(nat.lib().lqrrt_erf_batch(self.h, dg.data_ptr(), dx.data_ptr(), B, e.data_ptr(), self._stream()))
        return e.cpu().numpy()

    def lqr_dare_batch(self, x, u, Q, R, eps=1e-6):
        """(S, K, A, B, iterations) of the finite-difference-linearised DARE at every (x[i], u[i])."""
        torch = _torch()
        Bn = len(x)
        dev = "cuda:%d" % self.device
        dx, du = self._dev(x, (Bn, self.n)), self._dev(u, (Bn, self.m))
        dQ, dR = self._dev(Q, (self.n, self.n)), self._dev(R, (self.m, self.m))
        S = torch.empty((Bn, self.n, self.n), dtype=torch.float64, device=dev)
        K = torch.empty((Bn, self.m, self.n), dtype=torch.float64, device=dev)
        A = torch.empty((Bn, self.n, self.n), dtype=torch.float64, device=dev)
        Bm = torch.empty((Bn, self.n, self.m), dtype=torch.float64, device=dev)
        it = torch.empty(Bn, dtype=torch.int32, device=dev)
        nat.check(nat.lib().lqrrt_lqr_dare_batch(self.h, dx.data_ptr(), du.data_ptr(), Bn, dQ.data_ptr(), dR.data_ptr(),
                                                 float(eps), S.data_ptr(), K.data_ptr(), A.data_ptr(), Bm.data_ptr(),
                                                 it.data_ptr(), self._stream()))
        return S.cpu().numpy(), K.cpu().numpy(), A.cpu().numpy(), Bm.cpu().numpy(), it.cpu().numpy()

    def nn_argmin(self, xs, S=None, use_ignore=True):
        torch = _torch()
        W = len(xs)
        dxs = self._dev(xs, (W, self.n))
        dS = self._dev(S, (self.n, self.n)) if S is not None else None
        ids = torch.empty(W, dtype=torch.int32, device=dxs.device)
        cost = torch.empty(W, dtype=torch.float64, device=dxs.device)
        nat.check(nat.lib().lqrrt_nn_argmin(self.h, dxs.data_ptr(), W, dS.data_ptr() if dS is not None else None,
                                            1 if use_ignore else 0, ids.data_ptr(), cost.data_ptr(), self._stream()))
        return ids.cpu().numpy(), cost.cpu().numpy()

    def costs_to_go(self, x, S=None):
        torch = _torch()
        dx = self._dev(x, (self.n,))
        dS = self._dev(S, (self.n, self.n)) if S is not None else None
        cost = torch.empty(self.size, dtype=torch.float64, device=dx.device)
        nat.check(nat.lib().lqrrt_costs_to_go(self.h, dx.data_ptr(), dS.data_ptr() if dS is not None else None,
                                              cost.data_ptr(), self._stream()))
        return cost.cpu().numpy()

    def steer_batch(self, parents, xtar):
        torch = _torch()
        W = len(xtar)
        H = self.horizon_iters
        dev = "cuda:%d" % self.device
        dp = torch.from_numpy(np.ascontiguousarray(parents, dtype=np.int32)).to(dev)
        dx = self._dev(xtar, (W, self.n))
        ln = torch.empty(W, dtype=torch.int32, device=dev)
        xs = torch.empty((W, H, self.n), dtype=torch.float64, device=dev)
        us = torch.empty((W, H, self.m), dtype=torch.float64, device=dev)
        xe = torch.empty((W, self.n), dtype=torch.float64, device=dev)
        Ke = torch.empty((W, self.m, self.n), dtype=torch.float64, device=dev)
        nat.check(nat.lib().lqrrt_steer_batch(self.h, dp.data_ptr(), dx.data_ptr(), W, ln.data_ptr(), xs.data_ptr(),
                                              us.data_ptr(), xe.data_ptr(), Ke.data_ptr(), self._stream()))
        return ln.cpu().numpy(), xs.cpu().numpy(), us.cpu().numpy(), xe.cpu().numpy(), Ke.cpu().numpy()

    def steer_force(self, parent, xtar, max_steps, rtol=1e-4, atol=1e-4):
        torch = _torch()
        dev = "cuda:%d" % self.device
        dx = self._dev(xtar, (self.n,))
        ln = torch.zeros(1, dtype=torch.int32, device=dev)
        xs = torch.empty((max_steps, self.n), dtype=torch.float64, device=dev)
        us = torch.empty((max_steps, self.m), dtype=torch.float64, device=dev)
        nat.check(nat.lib().lqrrt_steer_force(self.h, int(parent), dx.data_ptr(), int(max_steps), float(rtol), float(atol),
                                              ln.data_ptr(), xs.data_ptr(), us.data_ptr(), self._stream()))
        k = int(ln.cpu()[0])
        return xs[:k].cpu().numpy(), us[:k].cpu().numpy()

    def push_samples(self, xs):
        xs = nat.as_f64(xs)
        if xs.ndim != 2 or xs.shape[1] != self.n:
            raise ValueError("expected samples of shape (count, %d)" % self.n)
        nat.check(nat.lib().lqrrt_engine_push_samples(self.h, nat.ptr(xs), len(xs)))

    def queued_samples(self):
        return nat.check(nat.lib().lqrrt_engine_queued_samples(self.h))

    # -- wave engine -----------------------------------------------------------------------------
    def record_layout(self):
        lay = np.zeros(11, dtype=np.int32)
        nat.check(nat.lib().lqrrt_record_layout(self.h, nat.ptr(lay)))
        return lay

    def wave_records_ptr(self):
        p = C.c_void_p()
        nat.check(nat.lib().lqrrt_wave_records(self.h, C.byref(p)))
        return p.value

    def wave_suggest(self, wave_cap):
        return nat.check(nat.lib().lqrrt_wave_suggest(self.h, int(wave_cap)))

    def wave_speculate(self, W, lo, hi):
        nat.check(nat.lib().lqrrt_wave_speculate(self.h, W, lo, hi, self._stream()))

    def wave_scan_nodes(self, W, node_lo, node_hi, best_ptr):
        """Tree-sharded wave, phase 1: this rank's (cost, id) candidates [W][2] over nodes [node_lo, node_hi)."""
        nat.check(nat.lib().lqrrt_wave_scan_nodes(self.h, int(W), int(node_lo), int(node_hi), best_ptr, self._stream()))

    def wave_steer_candidates(self, W, parts, best_ptr):
        """Tree-sharded wave, phase 2: nearest node per sample from the gathered candidates [parts][W][2], then the steer."""
        nat.check(nat.lib().lqrrt_wave_steer_candidates(self.h, int(W), int(parts), best_ptr, self._stream()))

    def wave_commit(self, W, max_commit, node_limit, pruning=True):
        st = nat.ExtendStats()
        nat.check(nat.lib().lqrrt_wave_commit(self.h, W, int(max_commit), int(node_limit), 1 if pruning else 0,
                                              C.byref(st), self._stream()))
        return st

    def extend(self, wave, max_attempts=-1, node_limit=-1, until_size=0, pruning=True, stop_on_goal=False):
        st = nat.ExtendStats()
        nat.check(nat.lib().lqrrt_engine_extend(self.h, int(wave), int(max_attempts), int(node_limit), int(until_size),
                                                1 if pruning else 0, 1 if stop_on_goal else 0, C.byref(st),
                                                self._stream()))
        return st

    @staticmethod
    def extend_multi(engines, wave, max_attempts=-1, node_limit=-1, until_size=0, pruning=True, stop_on_goal=False):
        """lqrrt_engine_extend_multi: n independent engines (trees) advanced in lock step by one native loop, two launches per
        step whatever n is.  Every tree is what its engine grows alone with `extend`; returns the n stats blocks."""
        engines = list(engines)
        n = len(engines)
        if n < 1:
            raise ValueError("no engines")
        handles = (C.c_void_p * n)(*[e.h for e in engines])
        stats = (nat.ExtendStats * n)()
        nat.check(nat.lib().lqrrt_engine_extend_multi(handles, n, int(wave), int(max_attempts), int(node_limit), int(until_size),
                                                       1 if pruning else 0, 1 if stop_on_goal else 0, stats, engines[0]._stream()))
        return [stats[i] for i in range(n)]

    def extend_sharded(self, comm, scheme, wave, max_attempts=-1, node_limit=-1, until_size=0, pruning=True, stop_on_goal=False):
        """lqrrt_engine_extend_sharded: the native loop with one RCCL all-gather per wave (comm: parallel.NativeComm)."""
        st = nat.ExtendStats()
        nat.check(nat.lib().lqrrt_engine_extend_sharded(self.h, comm.h, {"sample": 0, "tree": 1}[scheme], int(wave), int(max_attempts),
                                                       int(node_limit), int(until_size), 1 if pruning else 0,
                                                       1 if stop_on_goal else 0, C.byref(st), self._stream()))
        return st

    def allgather_nodes(self, comm, W):
        nat.check(nat.lib().lqrrt_allgather_nodes(self.h, comm.h, int(W), self._stream()))

    def plan_best(self):
        end, steps, hits = C.c_int32(), C.c_int64(), C.c_int64()
        nat.check(nat.lib().lqrrt_plan_best(self.h, C.byref(end), C.byref(steps), C.byref(hits)))
        return end.value, steps.value, hits.value

    def counters(self):
        st = nat.ExtendStats()
        nat.check(nat.lib().lqrrt_engine_counters(self.h, C.byref(st)))
        return st

    def profile_enable(self, on=True, steer=True, every=1):
        """HIP-event timing of the NN scan launches (every `every`-th one) and of the steer launches when `steer`."""
        nat.check(nat.lib().lqrrt_profile_enable(self.h, ((2 if steer else 1) + 16 * (max(1, int(every)) - 1)) if on else 0))

    def profile_read(self):
        a, b, c, d, f = C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_int64()
        nat.check(nat.lib().lqrrt_profile_read(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(f)))
        return dict(nn_ms=a.value, nn_launches=b.value, nn_bytes=c.value, steer_ms=d.value, steer_launches=f.value)


class NodeTable(object):
    """
    An LQRRT_MODEL_GENERIC engine (include/lqrrt_hip.h): the node table of a tree whose plugins are host callables -- SoA states,
    cos/sin of the angular ones, parents, ignore set -- and the nearest-neighbour stage over it (planner.py:239-247, 340-350).
    No dynamics, lqr or feasibility is compiled in; gains and edges stay with the caller (lqrrt_amd/callback.py).
    """

    def __init__(self, nstates, ncontrols, angle_dims=(), capacity=100008, device=0, max_wave=64):
        nat.require_device()
        self.n, self.m, self.device = int(nstates), int(ncontrols), device
        if not 1 <= self.n <= 64:
            raise ValueError("the device node table holds 1 to 64 states per node, got %d" % self.n)
        self.angle_dims = tuple(sorted(int(d) for d in angle_dims))
        d = nat.SystemDesc()
        d.model, d.nstates, d.ncontrols = nat.MODEL_GENERIC, self.n, self.m
        d.n_params = 1 + len(self.angle_dims)
        d.params[0] = float(len(self.angle_dims))
        for k, dim in enumerate(self.angle_dims):
            d.params[1 + k] = float(dim)
        h = C.c_void_p()
        nat.check(nat.lib().lqrrt_engine_create(C.byref(d), device, int(capacity), int(max_wave), C.byref(h)))
        self.h = h
        self.capacity, self.max_wave = int(capacity), int(max_wave)
        self._id, self._cost = C.c_int32(), C.c_double()
        self._lib = nat.lib()
        self._stream = nat.current_stream(device)

    def close(self):
        if getattr(self, "h", None):
            nat.lib().lqrrt_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return nat.check(self._lib.lqrrt_tree_size(self.h))

    def reset(self, x0):
        """Tree(seed_state, ...) (tree.py:50): the table holds the seed only, the ignore set is empty."""
        x0 = nat.as_f64(x0, (self.n,))
        self._stream = nat.current_stream(self.device)
        nat.check(self._lib.lqrrt_tree_reset(self.h, nat.ptr(x0), self._stream))

    def load(self, states, pID, ignored=None):
        states = nat.as_f64(states)
        N = len(states)
        states = nat.as_f64(states, (N, self.n))
        pID = np.ascontiguousarray(pID, dtype=np.int32)
        ign = None if ignored is None else np.ascontiguousarray(ignored, dtype=np.uint8)
        self._stream = nat.current_stream(self.device)
        nat.check(self._lib.lqrrt_tree_load(self.h, N, nat.ptr(states), None, nat.ptr(pID), None, None, None,
                                            nat.ptr(ign) if ign is not None else None, self._stream))

    def append(self, parent, state):
        """Tree.add_node's device half (tree.py:77-96): asynchronous, ordered before the next query."""
        state = nat.as_f64(state, (self.n,))
        nat.check(self._lib.lqrrt_tree_append(self.h, int(parent), nat.ptr(state), None, 1, None, None, self._stream))

    def ignore(self, ids):
        """planner.py:270: the nodes of a finished path join the ignore set."""
        one = np.ones(1, dtype=np.uint8)
        for i in ids:
            nat.check(self._lib.lqrrt_tree_set_ignored(self.h, int(i), 1, nat.ptr(one)))

    def ignored(self):
        out = np.empty(self.size, dtype=np.uint8)
        nat.check(self._lib.lqrrt_tree_get_ignored(self.h, 0, len(out), nat.ptr(out)))
        return out.astype(bool)

    def truncate(self, size):
        nat.check(self._lib.lqrrt_tree_truncate(self.h, int(size)))

    def states(self):
        out = np.empty((self.size, self.n))
        nat.check(self._lib.lqrrt_tree_get_states(self.h, 0, len(out), nat.ptr(out)))
        return out

    def parents(self):
        out = np.empty(self.size, dtype=np.int32)
        nat.check(self._lib.lqrrt_tree_get_parents(self.h, 0, len(out), nat.ptr(out)))
        return out

    def nearest(self, x, S=None, use_ignore=True):
        """(id, cost) of the nearest eligible node to x under S (None = identity): one synchronous call, no copies."""
        x = nat.as_f64(x, (self.n,))
        Sp = None
        if S is not None:
            S = nat.as_f64(S, (self.n, self.n))
            Sp = nat.ptr(S)
        nat.check(self._lib.lqrrt_nn_argmin_host(self.h, nat.ptr(x), Sp, 1 if use_ignore else 0, C.byref(self._id), C.byref(self._cost),
                                                 self._stream))
        return self._id.value, self._cost.value

    def nearest_from_errors(self, errors, S=None, use_ignore=True):
        """The same selection for error rows erf(x, node i) the caller evaluated itself ([size][n])."""
        errors = nat.as_f64(errors, (self.size, self.n))
        Sp = None
        if S is not None:
            S = nat.as_f64(S, (self.n, self.n))
            Sp = nat.ptr(S)
        nat.check(self._lib.lqrrt_nn_argmin_errors(self.h, nat.ptr(errors), Sp, 1 if use_ignore else 0, C.byref(self._id),
                                                   C.byref(self._cost), self._stream))
        return self._id.value, self._cost.value

    def nn_argmin(self, xs, S=None, use_ignore=True):
        """Batched device form (lqrrt_nn_argmin): xs [W][n] -> ids [W], costs [W]."""
        torch = _torch()
        W = len(xs)
        dev = "cuda:%d" % self.device
        dxs = torch.from_numpy(nat.as_f64(xs, (W, self.n))).to(dev)
        dS = torch.from_numpy(nat.as_f64(S, (self.n, self.n))).to(dev) if S is not None else None
        ids = torch.empty(W, dtype=torch.int32, device=dev)
        cost = torch.empty(W, dtype=torch.float64, device=dev)
        nat.check(self._lib.lqrrt_nn_argmin(self.h, dxs.data_ptr(), W, dS.data_ptr() if dS is not None else None,
                                            1 if use_ignore else 0, ids.data_ptr(), cost.data_ptr(), nat.current_stream(self.device)))
        return ids.cpu().numpy(), cost.cpu().numpy()

    def costs_to_go(self, x, S=None):
        torch = _torch()
        dev = "cuda:%d" % self.device
        dx = torch.from_numpy(nat.as_f64(x, (self.n,))).to(dev)
        dS = torch.from_numpy(nat.as_f64(S, (self.n, self.n))).to(dev) if S is not None else None
        cost = torch.empty(self.size, dtype=torch.float64, device=dev)
        nat.check(self._lib.lqrrt_costs_to_go(self.h, dx.data_ptr(), dS.data_ptr() if dS is not None else None, cost.data_ptr(),
                                              nat.current_stream(self.device)))
        return cost.cpu().numpy()
