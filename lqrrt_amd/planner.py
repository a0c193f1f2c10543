"""
Planner -- drop-in for the reference's lqrrt.Planner (lqrrt/planner.py) whose extend path
runs on an MI355X.

Create an instance of Planner and then call update_plan to generate a plan internal to the
instance.  To get the state or effort at some time t, use get_state(t) and get_effort(t).

What is the same as the reference: constructor / update_plan / set_* / kill_update / unkill
signatures, result attributes (tree, node_seq, x_seq, u_seq, t_seq, T, plan_reached_goal,
get_state, get_effort), ValueError conventions and return values.

What is different: the per-iteration loop body (planner.py:233-290: sample, cost-to-go
nearest neighbour, LQR-policy steer with feasibility sweep, tree append, goal test) is
executed by the HIP engine in *waves* of up to `wave_size` samples.  In exact mode (the
default) a wave's result is identical to running its samples one after another: every sample is
first evaluated against the tree as it stood at the start of the wave, then the engine
re-checks each sample against the nodes accepted earlier in the same wave and re-steers the
few whose true parent was born inside the wave, iterating to the fix-point
(csrc/engine.hip).  wave_mode='synchronous' drops that validation: all samples of a wave (exactly
`wave_size` of them) see the tree as of the start of the wave -- not the reference's tree any more, but
the one oracle/lqrrt_oracle.c's orc_extend_sync defines, and several times faster.
dynamics / lqr / erf / is_feasible must be the plugin handles
of a native system (lqrrt_amd.systems); arbitrary Python callables raise ValueError.
"""
from __future__ import division

import time

import numpy as np
import scipy.interpolate

from . import _native as nat
from .constraints import Constraints
from .engine import Engine
from .systems import plugin_system
from .tree import Tree


class Planner:
    """
    To initialize, provide...

    dynamics, lqr: the `.dynamics` and `.lqr` handles of an lqrrt_amd.systems object
                   (same call signatures as the reference: xnext = dynamics(x, u, dt),
                   (S, K) = lqr(x, u)).

    constraints: Instance of the Constraints class (feasibility, goal region).

    horizon: The simulation duration in seconds used to extend the tree.

    dt: The simulation timestep in seconds used to extend the tree.

    FPR: Failed Path Retention factor.

    error_tol: The state error array or scalar defining controller convergence.

    erf: the `.erf` handle of the same system object.

    min_time, max_time, max_nodes, goal0, sys_time, printing: as in the reference.

    wave_size: (new, optional) upper bound on the samples evaluated per wave.

    wave_mode: (new, optional) 'exact' (default, the reference's sequential result) or 'synchronous'.

    device: (new, optional) HIP device ordinal.

    """

    def __init__(self, dynamics, lqr, constraints,
                 horizon, dt=0.05, FPR=0,
                 error_tol=0.05, erf=np.subtract,
                 min_time=0.5, max_time=1, max_nodes=1E5,
                 goal0=None, sys_time=time.time, printing=True,
                 wave_size=1024, device=0, wave_mode='exact'):

        self.device = device
        self.wave_size = int(wave_size)
        if wave_mode not in ('exact', 'synchronous'):
            raise ValueError("wave_mode must be 'exact' or 'synchronous'")
        self.wave_mode = wave_mode
        self._engine = None
        self._engine_key = None

        self.set_system(dynamics, lqr, constraints, erf)

        self.set_resolution(horizon, dt, FPR, error_tol)

        self.set_runtime(min_time, max_time, max_nodes, sys_time)

        self.set_goal(goal0)

        self.printing = printing
        self.killed = False
        self.stats = None

#################################################

    def _get_engine(self):
        """(Re)creates the native engine when the system or the capacity changed."""
        capacity = int(self.max_nodes) + self.wave_size + 8
        key = (id(self.system), capacity, self.wave_size, self.device)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(self.system, capacity=capacity, max_wave=self.wave_size, device=self.device)
            self._engine_key = key
        self._engine.sync_geometry()        # the world may have changed since the last plan (new map, new obstacles)
        self._engine.set_wave_mode(self.wave_mode)
        return self._engine

    def update_plan(self, x0, sample_space, goal_bias=0,
                    guide=None, xrand_gen=None, pruning=True,
                    finish_on_goal=False, specific_time=None):
        """
        A new tree is grown from the seed x0 in an attempt to plan a path to the goal
        (planner.py:104-336).  Arguments and return value as in the reference; xrand_gen may
        be None or an integer >= 1 (tries allowed for a feasible random sample).

        Returns True if it finished fully, or False if it was haulted (killed, tree exceeded
        max_nodes, or no goal set).
        """
        # Safety first!
        x0 = np.array(x0, dtype=np.float64)
        if self.goal is None:
            print("No goal has been set yet!")
            self.get_state = lambda t: x0
            self.get_effort = lambda t: np.zeros(self.ncontrols)
            return False

        if specific_time is None:
            min_time = self.min_time
            max_time = self.max_time
        else:
            min_time = specific_time
            max_time = specific_time

        # Default sampler description (planner.py:176-198)
        if xrand_gen is None or type(xrand_gen) is int:
            if goal_bias is None:
                goal_bias = [0] * self.nstates
            elif hasattr(goal_bias, '__contains__'):
                if len(goal_bias) != self.nstates:
                    raise ValueError("Expected goal_bias to be scalar or have same length as state.")
            else:
                goal_bias = [goal_bias] * self.nstates
            tries_limit = xrand_gen if (xrand_gen is not None and xrand_gen > 0) else 10
            sample_space = np.array(sample_space, dtype=np.float64)
            if sample_space.shape != (self.nstates, 2):
                raise ValueError("Expected sample_space to be list of nstates tuples.")
            sampling_centers = np.mean(sample_space, axis=1)
            sampling_spans = np.diff(sample_space).flatten()
        else:
            # A user sampling function (planner.py:213-216).  It is called once per sample, in order, but a
            # whole batch ahead of the wave that consumes it, so it sees the tree as of the batch start
            # rather than of the previous iteration (documented deviation; the default sampler never looks
            # at the tree, so it is unaffected).
            if not hasattr(xrand_gen, '__call__'):
                raise ValueError("Expected xrand_gen to be None, an integer >= 1,  or a function.")

        # Store guide state
        if guide is None:
            self.xguide = np.copy(self.goal)
        else:
            self.xguide = np.array(guide, dtype=np.float64)

        # Reset the tree on the device (planner.py:172)
        eng = self._get_engine()
        if self.hfactor:
            # adaptive horizon: rollouts go to hspan[1] steps (see include/lqrrt_hip.h, lqrrt_resolution.adaptive)
            eng.set_resolution(self.dt, self.FPR, int(self.hspan[1]), self.error_tol, self.goal,
                               self.constraints.goal_buffer, adaptive=True, hspan_min=int(self.hspan[0]),
                               horizon_iters_state=int(self.horizon_iters))
        else:
            eng.set_resolution(self.dt, self.FPR, self.horizon_iters, self.error_tol, self.goal,
                               self.constraints.goal_buffer)
        eng.tree_reset(x0)
        user_sampler = hasattr(xrand_gen, '__call__')
        if not user_sampler:
            eng.set_sampler(sampling_centers, sampling_spans, np.array(goal_bias, dtype=np.float64), tries_limit)
            eng.seed_from_numpy_global()
        self.tree = Tree(eng)

        if self.printing:
            print("\n...planning...")
        self.plan_reached_goal = False
        self.T = np.inf
        time_elapsed = 0
        time_start = self.sys_time()
        best_end = -1
        total = None

        # Planning loop: each native call grows the tree by a few waves and returns at every goal hit
        while True:
            budget = 4 * self.wave_size
            if user_sampler:
                missing = budget - eng.queued_samples()
                if missing > 0:
                    eng.push_samples(np.array([np.array(xrand_gen(self), dtype=np.float64) for _ in range(missing)]))
            st = eng.extend(self.wave_size, max_attempts=budget, node_limit=int(self.max_nodes),
                            pruning=pruning, stop_on_goal=True)
            total = st if total is None else _add_stats(total, st)

            if st.goal_hits:
                self.plan_reached_goal = True
                end, steps, _ = eng.plan_best()
                if end != best_end:                    # a faster plan was found (planner.py:276)
                    best_end = end
                    self.T = steps * self.dt
                    if self.printing:
                        print("Found plan at elapsed time: {} s".format(np.round(time_elapsed, 6)))

            time_elapsed = self.sys_time() - time_start

            if self.killed:
                break

            elif self.plan_reached_goal and time_elapsed >= min_time:
                self._adopt_plan(best_end)
                if finish_on_goal:
                    self._finish_on_goal()
                if self.printing:
                    print("Tree size: {0}\nETA: {1} s".format(self.tree.size, np.round(self.T, 2)))
                self._prepare_interpolators()
                break

            elif time_elapsed >= max_time or self.tree.size > self.max_nodes:
                # Find closest node to guide state (planner.py:311-323)
                Sguide = np.array(self.system.Smatrix(), dtype=np.float64)
                for i, g in enumerate(self.constraints.goal_buffer):
                    if np.isinf(g):
                        Sguide[:, i] = 0
                ids, _ = eng.nn_argmin(self.xguide.reshape(1, -1), S=Sguide, use_ignore=False)
                self._adopt_plan(int(ids[0]))
                if self.printing:
                    print("Didn't reach goal.\nTree size: {0}\nETA: {1} s".format(self.tree.size, np.round(self.T, 2)))
                self._prepare_interpolators()
                break

        if not user_sampler:
            eng.sync_numpy_global()
        if self.hfactor:
            self.horizon_iters = eng.horizon_iters_state()       # planner.py:421,424: the heuristic's state persists
        self.stats = total.as_dict() if total is not None else None

        if self.killed or self.tree.size > self.max_nodes:
            if self.killed and best_end >= 0 and not hasattr(self, "node_seq"):
                self._adopt_plan(best_end)
            if self.printing:
                print("Plan update terminated abruptly!")
            self.killed = False
            return False
        else:
            return True

#################################################

    def _adopt_plan(self, end_node):
        """Climb + trajectory for the chosen end node (planner.py:266-281 / :319-323)."""
        self.node_seq = self.tree.climb(end_node)
        self.x_seq, self.u_seq = self.tree.trajectory(self.node_seq)
        self.T = len(self.x_seq) * self.dt
        self.t_seq = np.arange(len(self.x_seq)) * self.dt

    def _finish_on_goal(self):
        """
        planner.py:294-303: steer from the plan's last node to the exact goal (force_arrive) and, if that
        produced anything, tack it onto the plan and the tree.  The reference stops this rollout on a
        wall-clock timeout of clip(min_time/2, 0.1, inf) seconds (:402-406); here the same budget is turned
        into a step cap at the reference's measured ~0.3 ms per simulated step, which is deterministic.
        """
        budget_s = float(np.clip(self.min_time / 2, 0.1, np.inf))
        max_steps = int(min(max(budget_s / 3e-4, 64), 20000))
        if getattr(self, "force_arrive_max_steps", None):
            max_steps = int(self.force_arrive_max_steps)        # explicit override of the timeout stand-in
        xgoal_seq, ugoal_seq = self._engine.steer_force(self.node_seq[-1], self.goal, max_steps)
        if len(xgoal_seq) == max_steps and self.printing:
            print("(exact goal-convergence timed-out)")
        if len(xgoal_seq) > 0:
            xs, us = [row for row in xgoal_seq], [row for row in ugoal_seq]
            self.tree.add_node(self.node_seq[-1], self.goal, None, xs, us)
            self.node_seq.append(self.tree.size - 1)
            self.x_seq.extend(xs)
            self.u_seq.extend(us)
            self.t_seq = np.arange(len(self.x_seq)) * self.dt

    def _in_goal(self, x):
        """Returns True if some state x is in the goal region (planner.py:442-447)."""
        return all(goal_span[0] < v < goal_span[1] for goal_span, v in zip(self.goal_region, x))

    def _prepare_interpolators(self):
        """Updates the interpolator functions the user calls (planner.py:451-464)."""
        if len(self.x_seq) == 1:
            self.get_state = lambda t: self.x_seq[0]
            self.get_effort = lambda t: np.zeros(self.ncontrols)
        else:
            self.get_state = scipy.interpolate.interp1d(self.t_seq, np.array(self.x_seq), axis=0, assume_sorted=True,
                                                        bounds_error=False, fill_value=self.x_seq[-1][:])
            self.get_effort = scipy.interpolate.interp1d(self.t_seq, np.array(self.u_seq), axis=0, assume_sorted=True,
                                                         bounds_error=False, fill_value=self.u_seq[-1][:])

#################################################

    def set_goal(self, goal):
        """
        Modifies the goal state and region (planner.py:468-487).
        Be sure to update the plan after modifying the goal.
        """
        if goal is None:
            self.goal = None
        else:
            if len(goal) == self.nstates:
                self.goal = np.array(goal, dtype=np.float64)
            else:
                raise ValueError("The goal state must have same dimensionality as state space.")

            goal_region = []
            for i, buff in enumerate(self.constraints.goal_buffer):
                goal_region.append((self.goal[i] - buff, self.goal[i] + buff))

            self.goal_region = goal_region
            self.plan_reached_goal = False

#################################################

    def set_runtime(self, min_time=None, max_time=None, max_nodes=None, sys_time=None):
        """Arguments not given are not modified (planner.py:491-513)."""
        if min_time is not None:
            self.min_time = min_time

        if max_time is not None:
            self.max_time = max_time

        if self.min_time > self.max_time:
            raise ValueError("The min_time must be less than or equal to the max_time.")

        if max_nodes is not None:
            self.max_nodes = max_nodes

        if sys_time is not None:
            if hasattr(sys_time, '__call__'):
                self.sys_time = sys_time
            else:
                raise ValueError("Expected sys_time to be a function.")

#################################################

    def set_resolution(self, horizon=None, dt=None, FPR=None, error_tol=None):
        """Arguments not given are not modified (planner.py:517-553)."""
        if horizon is not None:
            self.horizon = horizon

        if dt is not None:
            self.dt = dt

        if FPR is not None:
            self.FPR = FPR

        if error_tol is not None:
            if np.shape(error_tol) in [(), (self.nstates,)]:
                self.error_tol = np.abs(error_tol).astype(np.float64)
            else:
                raise ValueError("Shape of error_tol must be scalar or length of state.")

        if hasattr(self.horizon, '__contains__'):
            if len(self.horizon) != 2:
                raise ValueError("Expected horizon to be tuple (min, max) or a single scalar.")
            if self.horizon[0] < self.dt:
                raise ValueError("The minimum horizon must be at least as big as dt.")
            if self.horizon[0] >= self.horizon[1]:
                raise ValueError("A horizon range tuple must be given as (min, max) where min < max.")
            self.horizon_iters = 1
            self.hspan = np.divide(self.horizon, self.dt).astype(np.int64)
            self.hfactor = int(2)
        elif self.horizon >= self.dt:
            self.horizon_iters = int(self.horizon / self.dt)
            self.hspan = (self.horizon_iters, self.horizon_iters)
            self.hfactor = 0
        else:
            raise ValueError("The horizon must be at least as big as dt.")

#################################################

    def set_system(self, dynamics=None, lqr=None, constraints=None, erf=None):
        """
        Arguments not given are not modified (planner.py:557-592).
        If dynamics gets modified, so must lqr (and vis versa).  All handles must come from
        the same native system object.
        """
        if dynamics is not None or lqr is not None:
            if hasattr(dynamics, '__call__'):
                system = plugin_system(dynamics, "dynamics")
            else:
                raise ValueError("Expected dynamics to be a function.")
            if hasattr(lqr, '__call__'):
                if plugin_system(lqr, "lqr") is not system:
                    raise ValueError("dynamics and lqr belong to different native systems.")
            else:
                raise ValueError("Expected lqr to be a function.")
            self.dynamics = dynamics
            self.lqr = lqr
            self.system = system

        if constraints is not None:
            if isinstance(constraints, Constraints):
                self.constraints = constraints
                self.nstates = self.constraints.nstates
                self.ncontrols = self.constraints.ncontrols
            else:
                raise ValueError("Expected constraints to be an instance of the Constraints class.")

        if erf is not None:
            if hasattr(erf, '__call__'):
                if erf is np.subtract:
                    if self.system.wrap_dims:
                        raise ValueError("This system has angular states; pass its .erf handle.")
                elif plugin_system(erf, "erf") is not self.system:
                    raise ValueError("erf belongs to a different native system.")
                self.erf = erf
            else:
                raise ValueError("Expected erf to be a function.")

        if getattr(self, "constraints", None) is not None and getattr(self, "system", None) is not None:
            if self.constraints.system is not self.system:
                raise ValueError("constraints.is_feasible belongs to a different native system.")

        self.plan_reached_goal = False

#################################################

    def kill_update(self):
        """Raises a flag that will cause an abrupt termination of the update_plan routine."""
        self.killed = True

    def unkill(self):
        """Lowers the kill_update flag. Do this if you made a mistake."""
        self.killed = False

    def visualize(self, dx, dy):
        """Plots the (dx,dy)-cross-section of the current tree, highlighting the plan."""
        if hasattr(self, 'node_seq'):
            self.tree.visualize(dx, dy, node_seq=self.node_seq)
        else:
            print("There is no plan to visualize!")


def _add_stats(a, b):
    out = nat.ExtendStats()
    for k, _ in nat.ExtendStats._fields_:
        setattr(out, k, getattr(a, k) + getattr(b, k))
    out.tree_size = b.tree_size
    out.stop_reason = b.stop_reason
    out.candidates = b.candidates
    return out
