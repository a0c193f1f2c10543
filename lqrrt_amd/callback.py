"""
Callback mode -- the reference's plugin API proper: dynamics / lqr / erf / is_feasible given as ARBITRARY Python callables
(planner.py:35-59, constraints.py:27), e.g. the functions of a demo script, unchanged.

A GPU cannot call Python, so in this mode the loop of planner.py:233-290 runs on the host, in the reference's order of events
(sample -> nearest -> steer with a feasibility test per step -> add node -> goal bookkeeping -> clock / kill flag), and calls
the user's functions exactly where the reference calls them.  What moves to the MI355X is the stage the reference spends
74 % (3k nodes) to 95 % (8k nodes) of its time in: Planner._costs_to_go and the nearest selection
(planner.py:239-247, 340-350).  The node table -- SoA states, cos/sin of the angular ones, parents, the ignore set -- lives in
HBM behind an LQRRT_MODEL_GENERIC engine (csrc/generic.hpp); every iteration makes ONE query (lqrrt_nn_argmin_host: the
sample and S = lqr(sample, 0)[0] travel as kernel arguments, the answer returns through mapped pinned memory) and at most one
append (lqrrt_tree_append: the new state as kernel arguments).

The engine has to know what `erf` does to evaluate it for every node.  It is PROBED (64 seeded random pairs, np.random
untouched) for the two forms every shipped plugin has (demo_car.py:115-126, demo_pendulum.py:130-142, np.subtract):
      e[d] = xgoal[d] - x[d]                    or, for an angular state,   e[d] = that difference wrapped to (-pi, pi]
and a caller may declare the angular states (`Planner(..., angle_dims=(2,))`): the probe must then agree, ValueError
otherwise.  An erf of any other form is evaluated on the host, node by node, as planner.py:588 does, and the device does the
contraction with S and the selection on the uploaded error rows -- the reference's cost of N Python calls per iteration stays,
which is said once when `printing` is on.

Results: the tree of the reference's own Planner on the same np.random stream -- parents, edge lengths and the generator's
end state exactly, floating point to what the user's NumPy functions reproduce (tests/test_callback_gpu.py replays the
reference's fixtures from plain Python plugins).  Up to 64 states per node (12 with compile-time kernels, beyond that the state
dimension is a run-time value of the scan).  There is no CPU path: without the HIP library and a device this mode raises.
"""
from __future__ import division

import numpy as np

from . import _native as nat
from .engine import NodeTable


# -------------------------------------------------------------------------------------------------- erf classification

def classify_erf(erf, nstates, declared=None, pairs=64, scale=10.0):
    """
    Which states `erf(xgoal, x)` treats as angles: a tuple of indices, or None when erf is not of the subtract-and-wrap form.
    `declared`: the caller's statement (Planner's angle_dims); a probe that disagrees with it raises ValueError.
    """
    if erf is np.subtract:
        found = ()
    else:
        rs = np.random.RandomState(20240607)                    # a stream of its own: the planner's draws are not disturbed
        A = rs.uniform(-scale, scale, (pairs, nstates))
        B = rs.uniform(-scale, scale, (pairs, nstates))
        found = None
        try:
            E = np.array([np.asarray(erf(np.copy(a), np.copy(b)), dtype=np.float64) for a, b in zip(A, B)])
        except Exception:
            E = None
        if E is not None and E.shape == (pairs, nstates) and np.all(np.isfinite(E)):
            D = A - B
            plain = np.all(np.abs(E - D) <= 1e-12 * scale, axis=0)
            turns = np.round((E - D) / (2 * np.pi))
            angular = np.all((np.abs(E - D - 2 * np.pi * turns) <= 1e-9) & (np.abs(E) <= np.pi + 1e-9), axis=0)
            if np.all(plain | angular):
                found = tuple(int(d) for d in range(nstates) if not plain[d])
    if declared is not None:
        want = tuple(sorted(int(d) for d in declared))
        if any(d < 0 or d >= nstates for d in want) or len(set(want)) != len(want):
            raise ValueError("angle_dims must be distinct state indices below %d." % nstates)
        if found is None or tuple(found) != want:
            raise ValueError("erf does not behave as declared by angle_dims=%r: probing it on %d random state pairs found %s."
                             % (want, pairs, "angular states %r" % (found,) if found is not None else "no subtract-and-wrap form"))
    return found


def _is_identity(S, n):
    return S.shape == (n, n) and np.array_equal(S, np.eye(n))


# -------------------------------------------------------------------------------------------------- the loop

class CallbackRun(object):
    """One update_plan of a planner whose plugins are host callables.  Holds the device node table and the host mirrors."""

    def __init__(self, planner):
        self.p = planner
        self.table = None

    # -- device side -----------------------------------------------------------------------------------------------------
    def _table(self):
        p = self.p
        cap = int(p.max_nodes) + 8
        key = (p.nstates, tuple(p._erf_angles or ()), cap, p.device)
        if self.table is None or self.table.key != key:
            if self.table is not None:
                self.table.close()
            self.table = NodeTable(p.nstates, p.ncontrols, p._erf_angles or (), capacity=cap, device=p.device)
            self.table.key = key
        return self.table

    def nearest(self, x, pruning):
        """planner.py:239-247: the node with the least cost-to-go to x among those not on a finished path."""
        p = self.p
        S = np.asarray(p.lqr(x, np.zeros(p.ncontrols))[0], dtype=np.float64)          # S about the SAMPLE, planner.py:344-345
        S_arg = None if _is_identity(S, p.nstates) else np.ascontiguousarray(S)
        if p._erf_angles is not None:
            return self.table.nearest(x, S_arg, use_ignore=pruning)[0]
        # erf of unknown form: one Python call per node (planner.py:588), contraction and selection on the device
        diffs = -np.apply_along_axis(p.erf, 1, self.states[:self.size], x)
        return self.table.nearest_from_errors(diffs, S_arg, use_ignore=pruning)[0]

    # -- host side -------------------------------------------------------------------------------------------------------
    def steer(self, ID, xtar, force_arrive=False):
        """
        planner.py:354-438 with the user's callables: K-gain rollout from node ID toward xtar; returns (x_seq, u_seq) without the
        start state.  Order per step: error, effort, dynamics on copies, feasibility (failure keeps int(FPR * len) steps), stop
        rules (horizon / tolerance, or allclose / wall clock when force_arrive), record, next gain.
        """
        p = self.p
        K = np.copy(p.tree.lqr[ID][1])
        x = np.copy(self.states[ID])
        xs, us = [], []
        steps, before = 0, np.inf
        t0 = p.sys_time()
        while True:
            e = p.erf(np.copy(xtar), np.copy(x))
            u = K.dot(e)
            x = p.dynamics(np.copy(x), np.copy(u), p.dt)
            if not p.constraints.is_feasible(x, u):
                keep = int(p.FPR * len(xs))
                xs, us = xs[:keep], us[:keep]
                break
            if force_arrive:
                if p.sys_time() - t0 > np.clip(p.min_time / 2, 0.1, np.inf):            # planner.py:402-406
                    if p.printing:
                        print("(exact goal-convergence timed-out)")
                    break
                if np.allclose(x, xtar, rtol=1E-4, atol=1E-4):
                    break
            else:
                steps += 1
                emag = np.abs(e)
                if p.hfactor:                                                          # adaptive horizon, planner.py:418-425
                    if np.all(emag >= before):
                        xs, us = [], []
                        p.horizon_iters = int(np.clip(p.horizon_iters / p.hfactor, p.hspan[0], p.hspan[1]))
                        break
                    if steps == p.horizon_iters:
                        p.horizon_iters = int(np.clip(p.hfactor * p.horizon_iters, p.hspan[0], p.hspan[1]))
                    before = emag
                if steps > p.horizon_iters or np.all(emag <= p.error_tol):
                    break
            xs.append(x)
            us.append(u)
            K = p.lqr(x, u)[1]
        return xs, us

    def add_node(self, parent, state, lqr, xs, us):
        p = self.p
        p.tree.add_node(parent, state, lqr, xs, us)
        if self.size == len(self.states):
            self.states = np.vstack((self.states, np.empty_like(self.states)))
        self.states[self.size] = state
        self.size += 1

    def default_sampler(self, sample_space, goal_bias, tries_limit):
        """planner.py:176-211: uniform over the sample space, goal-biased per state, up to tries_limit feasibility tests."""
        p = self.p
        space = np.array(sample_space, dtype=np.float64)
        if space.shape != (p.nstates, 2):
            raise ValueError("Expected sample_space to be list of nstates tuples.")
        centers, spans = np.mean(space, axis=1), np.diff(space).flatten()
        n, zero_u = p.nstates, np.zeros(p.ncontrols)

        def draw(planner):
            tries = 0
            while tries < tries_limit:
                x = centers + spans * (np.random.sample(n) - 0.5)
                gate = np.random.sample()
                for i, biased in enumerate(np.greater(goal_bias, gate)):
                    if biased:
                        x[i] = p.goal[i]
                if p.constraints.is_feasible(x, zero_u):
                    return x
                tries += 1
            return x
        return draw

    def run(self, x0, sample_space, goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time, resume=None):
        p = self.p
        x0 = np.array(x0, dtype=np.float64)
        if p.goal is None:
            print("No goal has been set yet!")
            p.get_state = lambda t: x0
            p.get_effort = lambda t: np.zeros(p.ncontrols)
            return False
        min_time, max_time = (p.min_time, p.max_time) if specific_time is None else (specific_time, specific_time)

        from .tree import Tree
        table = self._table()
        if resume is None:
            p.tree = Tree(x0, p.lqr(x0, np.zeros(p.ncontrols)))                        # planner.py:172
            table.reset(x0)
            self.states = np.empty((1024, p.nstates))
            self.states[0] = x0
            self.size = 1
        else:                                                                          # (measurement: continue on a given tree)
            tree, ign = resume if isinstance(resume, tuple) else (resume, None)
            p.tree = tree
            st = np.ascontiguousarray(tree.state, dtype=np.float64)
            table.load(st, np.asarray(tree.pID, dtype=np.int32), ign)
            self.states = np.vstack((st, np.empty_like(st)))
            self.size = len(st)

        if xrand_gen is None or type(xrand_gen) is int:
            if goal_bias is None:
                goal_bias = [0] * p.nstates
            elif hasattr(goal_bias, '__contains__'):
                if len(goal_bias) != p.nstates:
                    raise ValueError("Expected goal_bias to be scalar or have same length as state.")
            else:
                goal_bias = [goal_bias] * p.nstates
            tries_limit = xrand_gen if (xrand_gen is not None and xrand_gen > 0) else 10   # Py2's None > 0 is False: 10 tries
            xrand_gen = self.default_sampler(sample_space, goal_bias, tries_limit)
        elif not hasattr(xrand_gen, '__call__'):
            raise ValueError("Expected xrand_gen to be None, an integer >= 1,  or a function.")

        p.xguide = np.copy(p.goal) if guide is None else np.array(guide, dtype=np.float64)
        if p.printing:
            print("\n...planning...")
            if p._erf_angles is None and not getattr(p, "_erf_note_given", False):
                print("(erf is not of the subtract-and-wrap form: it is evaluated on the host for every node, every iteration)")
                p._erf_note_given = True
        p.plan_reached_goal = False
        p.T = np.inf
        elapsed, t_start = 0, p.sys_time()
        self.iterations = 0
        zero_u = np.zeros(p.ncontrols)

        while True:
            xrand = xrand_gen(p)
            near = self.nearest(np.asarray(xrand, dtype=np.float64), pruning)
            xs, us = self.steer(near, xrand)
            self.iterations += 1

            if len(xs) > 0:
                xnew = np.copy(xs[-1])
                self.add_node(near, xnew, p.lqr(xnew, np.copy(us[-1])), xs, us)
                table.append(near, xnew)
                if p._in_goal(xnew):                                                   # planner.py:260-283
                    p.plan_reached_goal = True
                    path = p.tree.climb(p.tree.size - 1)
                    px, pu = p.tree.trajectory(path)
                    table.ignore(path)                                                 # nodes of a finished path are not extended again
                    T = len(px) * p.dt
                    if T < p.T:
                        p.T, p.node_seq, p.x_seq, p.u_seq = T, path, px, pu
                        p.t_seq = np.arange(len(px)) * p.dt
                        if p.printing:
                            print("Found plan at elapsed time: {} s".format(np.round(elapsed, 6)))

            elapsed = p.sys_time() - t_start
            if p.killed:
                break
            if p.plan_reached_goal and elapsed >= min_time:
                if finish_on_goal:                                                     # planner.py:294-303
                    gx, gu = self.steer(p.node_seq[-1], p.goal, force_arrive=True)
                    if len(gx) > 0:
                        self.add_node(p.node_seq[-1], p.goal, None, gx, gu)
                        table.append(p.node_seq[-1], p.goal)
                        p.node_seq.append(p.tree.size - 1)
                        p.x_seq.extend(gx)
                        p.u_seq.extend(gu)
                        p.t_seq = np.arange(len(p.x_seq)) * p.dt
                if p.printing:
                    print("Tree size: {0}\nETA: {1} s".format(p.tree.size, np.round(p.T, 2)))
                p._prepare_interpolators()
                break
            if elapsed >= max_time or p.tree.size > p.max_nodes:
                # nearest node to the guide state, states with an infinite goal buffer not counted (planner.py:311-323)
                # (the columns are zeroed IN PLACE in whatever the user's lqr returned, as planner.py:314-316 does: a plugin that hands out
                #  one shared S -- behaviors/car.py:65,78 -- sees it changed in later plans, with the reference and here alike)
                Sguide = p.lqr(p.xguide, zero_u)[0]
                for i, gbuf in enumerate(p.constraints.goal_buffer):
                    if np.isinf(gbuf):
                        Sguide[:, i] = 0
                Sguide = np.array(Sguide, dtype=np.float64)
                if p._erf_angles is not None:
                    closest = table.nearest(p.xguide, np.ascontiguousarray(Sguide), use_ignore=False)[0]
                else:
                    diffs = -np.apply_along_axis(p.erf, 1, self.states[:self.size], p.xguide)
                    closest = table.nearest_from_errors(diffs, np.ascontiguousarray(Sguide), use_ignore=False)[0]
                p.node_seq = p.tree.climb(closest)
                p.x_seq, p.u_seq = p.tree.trajectory(p.node_seq)
                p.T = len(p.x_seq) * p.dt
                p.t_seq = np.arange(len(p.x_seq)) * p.dt
                if p.printing:
                    print("Didn't reach goal.\nTree size: {0}\nETA: {1} s".format(p.tree.size, np.round(p.T, 2)))
                p._prepare_interpolators()
                break

        p.stats = dict(attempts=self.iterations, accepted=p.tree.size - 1, tree_size=p.tree.size)
        if p.killed or p.tree.size > p.max_nodes:
            if p.printing:
                print("Plan update terminated abruptly!")
            p.killed = False
            return False
        return True


def require_device():
    """Callback mode computes its nearest-neighbour stage on the GPU and nowhere else."""
    if not nat.available():
        raise RuntimeError("lqrrt_amd: callback mode keeps the tree and the nearest-neighbour stage on an MI355X; "
                           "no HIP device / library found and there is no CPU path.")
