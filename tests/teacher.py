"""
Teacher forcing: replay a run of the REFERENCE planner decision by decision.

A `traj_*.npz` fixture written with teacher data (tools/gen_golden.py --job ...) holds, for every iteration t of
the reference's loop (planner.py:233-290): the sample `xrand_all[t]`, the node the reference chose `nearest[t]`
and the length of the edge its steer returned `steer_len[t]`; and the final tree (state, K, pID).  Nodes are only
ever appended and never modified, so the tree the reference searched at iteration t is the prefix of the final
tree of size 1 + #{s < t : steer_len[s] > 0}, and its `ignores` set (planner.py:270) is the union of the root
paths of the prefix nodes that lie in the goal region -- both reconstructed here from the reference's own numbers,
with no planner of ours in the loop.  A backend (C oracle, HIP engine) then has to make the reference's decision
from the reference's tree for every t: errors cannot accumulate, so chaotic dynamics (boat_advanced near
standstill) show up as isolated, countable mismatches instead of a divergence point.
"""
import os

import numpy as np


class Fixture(dict):
    """dict of a fixture's arrays with np.load's `.files`."""

    @property
    def files(self):
        return list(self.keys())


def load_fixture(path):
    """np.load of a traj_* fixture.  A compact `*_unpatched` fixture (tools/gen_golden.py compact_unpatched) lists the arrays
    that are identical to its patched twin's in `same_as_patched`; they are taken from that file."""
    g = np.load(path)
    if "same_as_patched" not in g.files:
        return g
    twin = np.load(path.replace("_unpatched", ""))
    out = Fixture({k: g[k] for k in g.files if k != "same_as_patched"})
    for k in g["same_as_patched"]:
        out[str(k)] = twin[str(k)]
    return out


class Schedule(object):
    def __init__(self, fx, goal, goal_buffer):
        self.xrand = np.asarray(fx["xrand_all"], dtype=np.float64)
        self.nearest = np.asarray(fx["nearest"], dtype=np.int64)
        self.steer_len = np.asarray(fx["steer_len"], dtype=np.int64)
        self.state = np.asarray(fx["state"], dtype=np.float64)
        self.K = np.asarray(fx["K"], dtype=np.float64)
        self.pID = np.asarray(fx["pID"], dtype=np.int64)
        self.iters = len(self.nearest)
        assert len(self.xrand) == self.iters == len(self.steer_len)
        added = self.steer_len > 0
        # tree size seen by iteration t, id of the node it adds (or -1)
        self.size_before = 1 + np.concatenate(([0], np.cumsum(added)[:-1]))
        self.new_node = np.where(added, self.size_before, -1)
        assert int(self.size_before[-1] + added[-1]) == len(self.state)
        assert np.all(self.nearest < self.size_before)
        # goal hits (strict box, planner.py:442-447) in node order; ignores grows at each of them
        g, b = np.asarray(goal, dtype=np.float64), np.asarray(goal_buffer, dtype=np.float64)
        inside = np.all((g - b < self.state) & (self.state < g + b), axis=1)
        inside[0] = False                                       # the seed is never goal-tested
        self.hit_nodes = np.flatnonzero(inside)
        self.epochs = []                                        # (first tree size at which these flags hold, flags)
        flags = np.zeros(len(self.state), dtype=np.uint8)
        self.epochs.append((1, flags.copy()))
        for node in self.hit_nodes:
            v = int(node)
            while v != -1:
                flags[v] = 1
                v = int(self.pID[v])
            self.epochs.append((int(node) + 1, flags.copy()))   # in force once the tree holds node `node`

    def ignored_at(self, size):
        """ignore flags of the tree prefix of `size` nodes (uint8, full length; entries >= size are 0)"""
        k = 0
        for j, (first, _) in enumerate(self.epochs):
            if first <= size:
                k = j
        return self.epochs[k][1]

    def groups(self):
        """(size, t0, t1): iterations t0..t1-1 all searched the prefix of `size` nodes; ascending in time"""
        out = []
        t0 = 0
        for t in range(1, self.iters + 1):
            if t == self.iters or self.size_before[t] != self.size_before[t0]:
                out.append((int(self.size_before[t0]), t0, t))
                t0 = t
        return out


def summarize(tag, sched, got_near, got_len, got_xend, cost_gap, speed_floor=1e-2, pos_cols=None):
    """Scores one backend's replay.  Returns a dict of counts and worst errors (also printed by the tests)."""
    it = sched.iters
    miss = np.flatnonzero(got_near != sched.nearest)
    gaps = np.array([cost_gap(int(t)) for t in miss]) if len(miss) else np.zeros(0)
    len_bad = np.flatnonzero(got_len != sched.steer_len)
    vel = sched.state[sched.nearest][:, 3:5] if sched.state.shape[1] >= 5 else None
    speed = np.hypot(vel[:, 0], vel[:, 1]) if vel is not None else np.full(it, np.inf)
    slow = speed <= speed_floor
    both = (got_len > 0) & (sched.steer_len > 0) & (got_len == sched.steer_len)
    err = np.zeros(it)
    idx = np.flatnonzero(both)
    err[idx] = np.abs(got_xend[idx] - sched.state[sched.new_node[idx]]).max(axis=1)
    out = dict(tag=tag, iterations=it, nearest_exact=int(it - len(miss)), nearest_miss=int(len(miss)),
               nearest_miss_max_rel_gap=float(gaps.max()) if len(gaps) else 0.0,
               steer_len_mismatch=int(len(len_bad)), steer_len_mismatch_slow_start=int(np.sum(slow[len_bad])),
               steer_len_mismatch_fast_start=int(np.sum(~slow[len_bad])),
               end_state_compared=int(len(idx)), end_state_max_err=float(err.max()) if len(idx) else 0.0,
               end_state_over_1e9=int(np.sum(err > 1e-9)), end_state_over_1e9_fast_start=int(np.sum((err > 1e-9) & ~slow)),
               end_state_median_err=float(np.median(err[idx])) if len(idx) else 0.0)
    return out
