"""
ORACLE (test infrastructure) -- ctypes wrapper of oracle/lqrrt_oracle.c, the plain-C sequential
restatement of the reference's extend path.  Used by tests/, smoke() and bench.py's cpu_baseline
leg only.  Takes the same plain-data problem description as the HIP engine (model id, parameter
block, hull/obstacle tables) so both sides are fed identical inputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liblqrrt_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        for name in ("orc_iterations", "orc_candidates", "orc_hits", "orc_best_steps"):
            getattr(L, name).restype = C.c_longlong
        L.orc_extend.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
        L.orc_extend_sync.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
        L.orc_enable_trace.argtypes = [C.c_void_p, C.c_longlong]
        L.orc_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.orc_get_trace_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
        L.orc_set_resolution.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 4
        L.orc_set_ogrid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_load_tree.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_set_ignored.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_nearest_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_costs_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_steer_from.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _UserModel(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("nw", C.c_int), ("wd", C.c_int * 2),
                ("gain", C.c_void_p), ("step", C.c_void_p), ("feasible", C.c_void_p)]


_user_libs = []


def use_user_model(path):
    """Registers the host build of an out-of-tree problem (tools/build_user_system.py --oracle) as the oracle's model 100
    (LQRRT_MODEL_USER): the callbacks are the engine's own header, the sequential loop is oracle/lqrrt_oracle.c's."""
    u = C.CDLL(path)
    _user_libs.append(u)                                    # keep it loaded: the oracle calls into it
    n, m, nw = C.c_int(), C.c_int(), C.c_int()
    wd = (C.c_int * 2)()
    u.lq_user_dims(C.byref(n), C.byref(m), C.byref(nw), wd)
    um = _UserModel(n.value, m.value, nw.value, wd, C.cast(u.lq_user_gain, C.c_void_p), C.cast(u.lq_user_step, C.c_void_p),
                    C.cast(u.lq_user_feasible, C.c_void_p))
    lib().orc_register_user(C.byref(um))
    return n.value, m.value


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class COracle(object):
    """Sequential planner for one problem (system = an object with model/params()/vps/obs/obs_stride)."""

    def __init__(self, system, capacity):
        self.n, self.m = system.nstates, system.ncontrols
        params = _f(system.params())
        vps = _f(system.vps)
        obs = _f(system.obs).reshape(-1, system.obs_stride)
        self.system = system
        self.h = C.c_void_p(lib().orc_create(system.model, _p(params), params.size, _p(vps), vps.shape[1],
                                             _p(obs), obs.shape[0], system.obs_stride, int(capacity)))
        if not self.h:
            raise ValueError("unknown model")
        og = getattr(system, "ogrid", None)
        if og is not None:
            grid = np.ascontiguousarray(og["grid"], dtype=np.int8)
            lib().orc_set_ogrid(self.h, _p(grid), grid.shape[0], grid.shape[1], float(og["origin"][0]),
                                float(og["origin"][1]), float(og["cpm"]), float(og["threshold"]))
        self.H = 1
        self.S = None if getattr(system, "S", None) is None else _f(system.S)

    def __del__(self):
        try:
            lib().orc_destroy(self.h)
        except Exception:
            pass

    def configure(self, dt, FPR, horizon_iters, error_tol, goal, goal_buffer, sample_space, goal_bias, tries=10):
        tol = _f(np.broadcast_to(np.abs(np.asarray(error_tol, dtype=np.float64)), (self.n,)))
        g = _f(goal)
        b = np.abs(_f(goal_buffer))
        lo, hi = _f(g - b), _f(g + b)
        lib().orc_set_resolution(self.h, float(dt), float(FPR), int(horizon_iters), _p(tol), _p(g), _p(lo), _p(hi))
        self.H = int(horizon_iters)
        space = _f(sample_space)
        centers, spans, bias = _f(np.mean(space, axis=1)), _f(np.diff(space).flatten()), _f(goal_bias)
        lib().orc_set_sampler(self.h, _p(centers), _p(spans), _p(bias), int(tries))

    def set_adaptive(self, hspan_min, hspan_max, state=1):
        """horizon=(min,max) mode; call after configure (H = hspan_max) and before reset."""
        lib().orc_set_adaptive(self.h, int(hspan_min), int(hspan_max), int(state))
        self.H = int(hspan_max)

    @property
    def horizon_iters(self):
        return lib().orc_horizon_iters(self.h)

    def seed(self, seed):
        st = np.random.RandomState(seed).get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        lib().orc_set_mt19937(self.h, _p(key), int(st[2]))

    def set_mt19937(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        lib().orc_set_mt19937(self.h, _p(key), int(pos))

    def reset(self, x0):
        x0 = _f(x0)
        lib().orc_reset(self.h, _p(x0))

    def enable_trace(self, cap):
        lib().orc_enable_trace(self.h, int(cap))

    def extend(self, max_iters=-1, max_nodes=-1, pruning=True, stop_on_goal=False):
        return lib().orc_extend(self.h, int(max_iters), int(max_nodes), 1 if pruning else 0, 1 if stop_on_goal else 0,
                                _p(self.S) if self.S is not None else None)

    def extend_sync(self, wave, max_iters=-1, max_nodes=-1, pruning=True, stop_on_goal=False):
        """Synchronous wave mode (every sample of a wave searches the wave-start snapshot); wave=1 == extend."""
        return lib().orc_extend_sync(self.h, int(wave), int(max_iters), int(max_nodes), 1 if pruning else 0,
                                     1 if stop_on_goal else 0, _p(self.S) if self.S is not None else None)

    @property
    def size(self):
        return lib().orc_size(self.h)

    @property
    def iterations(self):
        return lib().orc_iterations(self.h)

    @property
    def candidates(self):
        return lib().orc_candidates(self.h)

    @property
    def hits(self):
        return lib().orc_hits(self.h)

    def best(self):
        return lib().orc_best_end(self.h), lib().orc_best_steps(self.h)

    def states(self):
        out = np.empty((self.size, self.n))
        lib().orc_get_states(self.h, _p(out))
        return out

    def gains(self):
        out = np.empty((self.size, self.m, self.n))
        lib().orc_get_gains(self.h, _p(out))
        return out

    def parents(self):
        out = np.empty(self.size, dtype=np.int32)
        lib().orc_get_parents(self.h, _p(out))
        return out

    def edge_lengths(self):
        out = np.empty(self.size, dtype=np.int32)
        lib().orc_get_edge_lengths(self.h, _p(out))
        return out

    def ignored(self):
        out = np.empty(self.size, dtype=np.uint8)
        lib().orc_get_ignored(self.h, _p(out))
        return out.astype(bool)

    def edge(self, ID):
        x = np.empty((self.H, self.n))
        u = np.empty((self.H, self.m))
        ln = lib().orc_get_edge(self.h, int(ID), _p(x), _p(u))
        return x[:ln].copy(), u[:ln].copy()

    def trace(self):
        k = self.iterations
        near = np.empty(k, dtype=np.int32)
        ln = np.empty(k, dtype=np.int32)
        lib().orc_get_trace(self.h, _p(near), _p(ln), k)
        return near, ln

    def trace_samples(self):
        """the sample of every traced iteration, (iterations, n)"""
        xs = np.empty((self.iterations, self.n))
        lib().orc_get_trace_samples(self.h, _p(xs), self.iterations)
        return xs

    def nearest(self, x, S=None, pruning=False):
        x = _f(x)
        S = None if S is None else _f(S)
        return lib().orc_nearest(self.h, _p(x), _p(S) if S is not None else None, 1 if pruning else 0)

    # teacher forcing: somebody else's tree resident, single decisions replayed (tests/test_teacher_cpu.py)
    def load_tree(self, states, K, pID, ignored=None):
        states, K = _f(states), _f(K)
        pID = np.ascontiguousarray(pID, dtype=np.int32)
        ign = None if ignored is None else np.ascontiguousarray(ignored, dtype=np.uint8)
        if lib().orc_load_tree(self.h, len(states), _p(states), _p(K), _p(pID), _p(ign) if ign is not None else None) != 0:
            raise ValueError("tree does not fit (reset first; capacity %d)" % len(states))

    def set_ignored(self, ignored):
        ign = np.ascontiguousarray(ignored, dtype=np.uint8)
        lib().orc_set_ignored(self.h, _p(ign), len(ign))

    def nearest_prefix(self, x, count, pruning=True):
        x = _f(x)
        return lib().orc_nearest_prefix(self.h, _p(x), _p(self.S) if self.S is not None else None, 1 if pruning else 0, int(count))

    def costs_prefix(self, x, count):
        x = _f(x)
        out = np.empty(int(count))
        lib().orc_costs_prefix(self.h, _p(x), _p(self.S) if self.S is not None else None, int(count), _p(out))
        return out

    def steer_from(self, ID, xtar):
        xtar = _f(xtar)
        xs, us, K = np.empty((self.H + 1, self.n)), np.empty((self.H + 1, self.m)), np.empty((self.m, self.n))
        ln = lib().orc_steer_from(self.h, int(ID), _p(xtar), _p(xs), _p(us), _p(K))
        return ln, xs[:ln].copy(), us[:ln].copy(), K

    # single-call operators
    def dynamics(self, x, u):
        x, u = _f(x), _f(u)
        out = np.empty(self.n)
        lib().orc_dynamics(self.h, _p(x), _p(u), _p(out))
        return out

    def feasible(self, x, u):
        x, u = _f(x), _f(u)
        return bool(lib().orc_feasible(self.h, _p(x), _p(u)))

    def gain(self, x, u):
        x, u = _f(x), _f(u)
        out = np.empty((self.m, self.n))
        lib().orc_gain(self.h, _p(x), _p(u), _p(out))
        return out

    def lqr(self, x, u):
        """(S, K, iterations) of the Riccati lqr (pendulum_lqr only)."""
        x, u = _f(x), _f(np.atleast_1d(u))
        S, K = np.empty((self.n, self.n)), np.empty((self.m, self.n))
        it = lib().orc_lqr(self.h, _p(x), _p(u), _p(S), _p(K))
        return S, K, it

    def erf(self, xg, x):
        xg, x = _f(xg), _f(x)
        out = np.empty(self.n)
        lib().orc_erf(self.h, _p(xg), _p(x), _p(out))
        return out


def make(system, max_nodes, seed=1, tries=10, horizon=None):
    """COracle configured with the demo's PLAN kwargs (or an explicit horizon / (min,max) pair), seeded, reset at x0."""
    o = COracle(system, capacity=int(max_nodes) + 8)
    kw = system.plan_kwargs
    horizon = kw["horizon"] if horizon is None else horizon
    if hasattr(horizon, "__len__"):
        hspan = np.divide(horizon, kw["dt"]).astype(np.int64)
        o.configure(kw["dt"], kw["FPR"], int(hspan[1]), system.error_tol, system.goal, system.goal_buffer,
                    system.sample_space, system.goal_bias, tries)
        o.set_adaptive(hspan[0], hspan[1], 1)
    else:
        o.configure(kw["dt"], kw["FPR"], int(horizon / kw["dt"]), system.error_tol, system.goal, system.goal_buffer,
                    system.sample_space, system.goal_bias, tries)
    o.seed(seed)
    o.reset(system.x0)
    return o
