"""
Planner -- drop-in for the reference's lqrrt.Planner (lqrrt/planner.py) whose extend path runs on an MI355X.

Same as the reference: constructor / update_plan / set_* / kill_update / unkill signatures, result attributes
(tree, node_seq, x_seq, u_seq, t_seq, T, plan_reached_goal, get_state, get_effort), which arguments raise
ValueError and with which message, and update_plan's return value.

Different: the loop body of planner.py:233-290 (sample, cost-to-go nearest neighbour, LQR-policy steer with a
feasibility sweep per step, tree append, goal test) is executed by the HIP engine (csrc/engine.hip) in WAVES of
up to `wave_size` samples.  In 'exact' mode, the default, a wave's outcome is the one the samples would have
produced one after another: all samples are first evaluated against the tree as of the start of the wave, then
each is re-checked against the nodes accepted earlier in the same wave and the few whose true parent was born
inside the wave are steered again, to the fix-point.  'synchronous' mode skips that validation (every sample of
a wave of exactly `wave_size` sees the wave-start tree): not the reference's tree any more but the one
oracle/lqrrt_oracle.c's orc_extend_sync defines, several times faster.

Two kinds of plugins are accepted:
  * the handles of ONE native system object (lqrrt_amd.systems: the reference's demo problems, the ROS behaviours, an
    out-of-tree header): the whole loop body runs on the device as described above;
  * arbitrary Python callables with the reference's signatures (planner.py:35-59, constraints.py:27) -- "callback mode"
    (lqrrt_amd/callback.py): a GPU cannot call back into Python, so sampling, steering and feasibility run on the host in
    the reference's order of events, one iteration at a time, and the device keeps the node table and does
    Planner._costs_to_go + the nearest selection (95 % of the reference's time at 8k nodes).
There is no CPU path in either: without the HIP library and a device, update_plan raises.
"""
from __future__ import division

import os
import time

import numpy as np
import scipy.interpolate

from . import _native as nat
from .constraints import Constraints
from .engine import Engine
from . import callback
from .systems import native_system_of
from .tree import Tree, held_elsewhere


def _callable(f):
    return hasattr(f, '__call__')


class Planner:
    """
    dynamics, lqr, erf: handles of an lqrrt_amd.systems object, called like the reference's plugins
        (xnext = dynamics(x, u, dt);  (S, K) = lqr(x, u);  e = erf(xgoal, x)).
    constraints: a Constraints instance (feasibility handle of the same system, goal buffer).
    horizon: seconds simulated per tree extension, or (min, max) for the adaptive heuristic.
    dt: simulation step [s].   FPR: failed-path retention factor (planner.py:394-395).
    error_tol: state error (scalar or per state) below which a steer counts as converged.
    min_time, max_time, max_nodes, goal0, sys_time, printing: as in the reference (planner.py:61-82).
    wave_size, wave_mode, device: new and optional -- samples per wave (upper bound), 'exact' | 'synchronous',
        HIP device ordinal.
    angle_dims: new and optional, callback mode only -- the states `erf` wraps to (-pi, pi]; checked against a probe of erf
        (ValueError when they disagree).  None: the probe alone decides (lqrrt_amd/callback.py classify_erf).
    """

    def __init__(self, dynamics, lqr, constraints,
                 horizon, dt=0.05, FPR=0,
                 error_tol=0.05, erf=np.subtract,
                 min_time=0.5, max_time=1, max_nodes=1E5,
                 goal0=None, sys_time=time.time, printing=True,
                 wave_size=1024, device=0, wave_mode='exact', angle_dims=None):
        if wave_mode not in ('exact', 'synchronous'):
            raise ValueError("wave_mode must be 'exact' or 'synchronous'")
        self.device, self.wave_size, self.wave_mode = device, int(wave_size), wave_mode
        self._engine = self._engine_key = None
        self._callback_run = None
        self.angle_dims = angle_dims
        self.tree = None
        self.set_system(dynamics, lqr, constraints, erf)
        self.set_resolution(horizon, dt, FPR, error_tol)
        self.set_runtime(min_time, max_time, max_nodes, sys_time)
        self.set_goal(goal0)
        self.printing = printing
        self.killed = False
        self.stats = None
        # A user sampling function (xrand_gen = a callable, planner.py:213-216) is called once per iteration on the tree the previous
        # iteration left, like the reference calls it (planner.py:236): one sample per native call.  False: a function that never looks at
        # planner.tree / plan_reached_goal may be called a batch AHEAD of the waves that consume its samples (same samples, same tree, no
        # host turn per iteration).
        self.xrand_gen_sees_tree = True
        # HBM pools are sized by max_nodes (Engine.footprint(): ~1.7 kB per node for the boats with horizon_iters = 20, i.e. ~170 MB at
        # the reference's default of 1e5 nodes).  A planner built with an explicit max_nodes gets them now, outside any plan's time
        # budget; one left at the default gets them with its first update_plan -- or when the caller asks (warm_up()) -- so that a fleet
        # of planners built with defaults does not pin gigabytes before anyone plans.
        if max_nodes != 1E5:
            self.warm_up()
        else:
            self.warm_up_error = None if nat.available() else RuntimeError("lqrrt_amd: no engine created: %s" % (
                "liblqrrt_hip.so is not built" if not os.path.exists(nat.LIB_PATH) else "no HIP device visible"))

    # ------------------------------------------------------------------------------------------ engine
    def warm_up(self):
        """Creates the device engine (HBM pools sized by max_nodes, geometry upload) ahead of the first update_plan,
        so that none of it is charged to a plan's time budget.  Where there is no GPU or no built library nothing is created --
        constructing and configuring a Planner works anywhere, planning does not -- and the reason is kept in
        `self.warm_up_error` (None when the engine exists), so that a broken install can be diagnosed before the first plan;
        update_plan raises it in full."""
        self.warm_up_error = None
        if self.callback_mode:
            if not nat.available():
                self.warm_up_error = RuntimeError("lqrrt_amd: no engine created: %s" % (
                    "liblqrrt_hip.so is not built" if not os.path.exists(nat.LIB_PATH) else "no HIP device visible"))
            return
        try:
            if nat.available():
                self._get_engine()
            else:
                self.warm_up_error = RuntimeError("lqrrt_amd: no engine created: %s" % (
                    "liblqrrt_hip.so is not built" if not os.path.exists(nat.LIB_PATH) else "no HIP device visible"))
        except (nat.NativeError, RuntimeError, OSError) as ex:
            self._engine = self._engine_key = None
            self.warm_up_error = ex

    def _retire_tree(self):
        """The engine is about to be reused: the previous plan's Tree keeps its contents (copied out of HBM once -- tens of milliseconds
        for a 100k-node tree) IF anybody still holds it (the ROS node does, lqrrt_node.py:477); a tree that only this planner refers to
        is simply dropped, like the reference drops its Tree at planner.py:172."""
        t = self.tree
        if t is None or not getattr(t, "on_device", False):
            return
        del t
        if held_elsewhere(self, "tree"):
            self.tree._detach()
        else:
            self.tree._e, self.tree._generation = None, None        # nobody can read it again
            self.tree = None

    def _get_engine(self):
        """The native engine for the current system / capacity / device (recreated when one of them changed)."""
        capacity = int(self.max_nodes) + self.wave_size + 8
        key = (id(self.system), capacity, self.wave_size, self.device)
        if self._engine is None or self._engine_key != key:
            self._retire_tree()
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(self.system, capacity=capacity, max_wave=self.wave_size, device=self.device)
            self._engine_key = key
        self._engine.sync_geometry()        # the world may have changed since the last plan (new map, new obstacles)
        self._engine.set_wave_mode(self.wave_mode)
        return self._engine

    # ------------------------------------------------------------------------------------------ planning
    def update_plan(self, x0, sample_space, goal_bias=0,
                    guide=None, xrand_gen=None, pruning=True,
                    finish_on_goal=False, specific_time=None):
        """
        Grows a new tree from the seed x0 toward the goal and extracts a plan from it (planner.py:104-336);
        arguments as in the reference.  xrand_gen: None, an integer >= 1 (tries allowed per feasible random sample)
        or a function of the planner returning a sample.

        Returns True if it ran to completion, False if it was halted (killed, tree larger than max_nodes, no goal).
        """
        # the feasibility function may have been swapped on the Constraints object itself since set_system (the ROS node does:
        # lqrrt_node.py:65 planner.constraints.set_feasibility_function(...)): the mode follows the plugins as they are NOW
        self._resolve_mode()
        if self.callback_mode:
            return self._update_plan_callback(x0, sample_space, goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time)
        run = self._plan_begin(x0, sample_space, goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time)
        if run is None:
            return False
        # Each native call grows the tree by a few waves.  The clock and the kill flag are looked at between calls, so a
        # call is sized to what the time budget still allows.
        while True:
            budget = self._plan_budget(run)
            # The call only has to come back AT a goal hit when that hit can end the plan (min_time already elapsed,
            # planner.py:293); before that, hits are bookkept by the engine (lqrrt_plan_best) and reported with the call.
            t_call = time.perf_counter()
            st = run.eng.extend(self.wave_size, max_attempts=budget, node_limit=int(self.max_nodes),
                                pruning=pruning, stop_on_goal=bool(run.time_elapsed >= run.min_time))
            if self._plan_after_call(run, st, time.perf_counter() - t_call):
                break
        return self._plan_end(run)

    def _update_plan_callback(self, x0, sample_space, goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time, resume=None):
        """update_plan for plugins that are host callables: the reference's loop on the host, the nearest-neighbour stage on the
        device (lqrrt_amd/callback.py)."""
        if self.goal is not None:
            callback.require_device()
        if self._erf_angles is False:                               # erf is classified once per erf, at its first use
            self._erf_angles = callback.classify_erf(self.erf, self.nstates, self.angle_dims)
        if self._callback_run is None:
            self._callback_run = callback.CallbackRun(self)
        self._retire_tree()                                         # a tree grown by the native engine before the plugins were swapped
        return self._callback_run.run(x0, sample_space, goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time, resume=resume)

    # The three phases of update_plan, separately callable so that several planners can share native calls (update_plans below):
    # set-up (planner.py:157-231), what follows each native call (:260-311 as far as the host is concerned), wrap-up (:313-336).
    def _plan_begin(self, x0, sample_space, goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time, seed=None):
        x0 = np.array(x0, dtype=np.float64)
        if self.goal is None:
            print("No goal has been set yet!")
            self.get_state = lambda t: x0
            self.get_effort = lambda t: np.zeros(self.ncontrols)
            return None
        run = _PlanRun()
        run.min_time, run.max_time = (self.min_time, self.max_time) if specific_time is None else (specific_time, specific_time)
        run.pruning, run.finish_on_goal, run.xrand_gen = pruning, finish_on_goal, xrand_gen

        run.user_sampler = not (xrand_gen is None or type(xrand_gen) is int)
        if run.user_sampler:
            # planner.py:213-216.  The function is called once per sample, in order: by default once per native call of ONE attempt --
            # the reference's order of events exactly (planner.py:236: sample, extend, goal bookkeeping, next sample), so it may read
            # planner.tree / planner.plan_reached_goal.  `planner.xrand_gen_sees_tree = False` lets a function that does not look at
            # the tree be called a batch ahead of the wave that consumes the samples (it then sees the tree as of the batch start).
            if not _callable(xrand_gen):
                raise ValueError("Expected xrand_gen to be None, an integer >= 1,  or a function.")
        else:
            # description of the default sampler (planner.py:176-198)
            if goal_bias is None:
                bias = np.zeros(self.nstates)
            elif hasattr(goal_bias, '__contains__'):
                if len(goal_bias) != self.nstates:
                    raise ValueError("Expected goal_bias to be scalar or have same length as state.")
                bias = np.array(goal_bias, dtype=np.float64)
            else:
                bias = np.full(self.nstates, goal_bias, dtype=np.float64)
            tries_limit = xrand_gen if (xrand_gen is not None and xrand_gen > 0) else 10
            space = np.array(sample_space, dtype=np.float64)
            if space.shape != (self.nstates, 2):
                raise ValueError("Expected sample_space to be list of nstates tuples.")

        self.xguide = np.copy(self.goal) if guide is None else np.array(guide, dtype=np.float64)

        # the device tree is about to be overwritten: a Tree object from the previous plan keeps its contents
        self._retire_tree()
        eng = run.eng = self._get_engine()
        run.own_stream = seed is not None                           # (update_plans: a sample stream per planner, np.random untouched)
        if run.own_stream and not run.user_sampler:
            # (seeded before anything else: set_resolution rewinds the generator to the first uncommitted candidate of the plan before --
            #  a replay of every draw that plan consumed, milliseconds after a long one -- unless the stream has just been replaced;
            #  update_plan's own path ends with sync_numpy_global, which leaves nothing to replay)
            st = np.random.RandomState(seed).get_state()
            eng.set_mt19937(st[1], st[2])
        if self.hfactor:
            # adaptive horizon: rollouts may run hspan[1] steps (include/lqrrt_hip.h, lqrrt_resolution.adaptive)
            eng.set_resolution(self.dt, self.FPR, int(self.hspan[1]), self.error_tol, self.goal, self.constraints.goal_buffer,
                               adaptive=True, hspan_min=int(self.hspan[0]), horizon_iters_state=int(self.horizon_iters))
        else:
            eng.set_resolution(self.dt, self.FPR, self.horizon_iters, self.error_tol, self.goal, self.constraints.goal_buffer)
        eng.tree_reset(x0)                                          # planner.py:172
        if not run.user_sampler:
            eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), bias, tries_limit)
            if run.own_stream:
                st = np.random.RandomState(seed).get_state()
                eng.set_mt19937(st[1], st[2])
            else:
                eng.seed_from_numpy_global()
        S0 = self.system.Smatrix()
        self.tree = Tree(x0, (S0, np.zeros((self.ncontrols, self.nstates))))
        self.tree._bind(eng, S0)

        if self.printing:
            print("\n...planning...")
        self.plan_reached_goal = False
        self.T = np.inf
        run.time_elapsed = 0
        run.time_start = self.sys_time()
        run.best_end = -1
        run.total = None
        run.rate = None                                             # attempts per second of real time, measured
        run.adopted = False
        return run

    def _plan_budget(self, run):
        """Attempts the next native call may commit for this plan; a user sampling function is called for as many samples."""
        exit_at = run.min_time if self.plan_reached_goal else run.max_time
        budget = self._attempt_budget(run.rate, exit_at - run.time_elapsed)
        if run.user_sampler and self.xrand_gen_sees_tree:
            budget = 1
        if run.user_sampler:
            missing = budget - run.eng.queued_samples()
            if missing > 0:
                run.eng.push_samples(np.array([np.array(run.xrand_gen(self), dtype=np.float64) for _ in range(missing)]))
        return budget

    def _plan_after_call(self, run, st, dt_call, wrap_up=True):
        """Goal bookkeeping, clock, kill flag and exit rules after a native call (planner.py:260-323); True when the plan is over.
        The wrap-up of a finished plan (plan extraction, finish_on_goal, fallback search, interpolators: milliseconds of host work and
        copies out of HBM) follows at once, or -- wrap_up=False, several planners sharing native calls -- when the caller gets to it
        (`_plan_wrap_up`), so that one planner's wrap-up is not charged to the others' clocks."""
        eng = run.eng
        if st.attempts > 0 and dt_call > 0:
            run.rate = st.attempts / dt_call if run.rate is None else 0.5 * run.rate + 0.5 * st.attempts / dt_call
        run.total = st if run.total is None else _add_stats(run.total, st)

        if st.goal_hits:
            self.plan_reached_goal = True
            end, steps, _ = eng.plan_best()
            if end != run.best_end:                                 # a faster plan (planner.py:276)
                run.best_end = end
                self.T = steps * self.dt
                if self.printing:
                    print("Found plan at elapsed time: {} s".format(np.round(run.time_elapsed, 6)))

        run.time_elapsed = self.sys_time() - run.time_start

        run.outcome = None
        if self.killed:
            run.outcome = "killed"
        elif self.plan_reached_goal and run.time_elapsed >= run.min_time:
            run.outcome = "goal"
        elif run.time_elapsed >= run.max_time or self.tree.size > self.max_nodes:
            run.outcome = "fallback"
        if run.outcome is not None and wrap_up:
            self._plan_wrap_up(run)
        return run.outcome is not None

    def _plan_wrap_up(self, run):
        """What the reference does between leaving its loop and returning (planner.py:293-328) for a plan that `_plan_after_call` ended."""
        if run.outcome == "goal":
            self._adopt_plan(run.best_end)
            run.adopted = True
            if run.finish_on_goal:
                self._finish_on_goal()
            if self.printing:
                print("Tree size: {0}\nETA: {1} s".format(self.tree.size, np.round(self.T, 2)))
            self._prepare_interpolators()
        elif run.outcome == "fallback":
            # no goal hit (or not enough time spent): plan to the node nearest the guide state (planner.py:311-323)
            if getattr(self.system, "riccati", False):
                self.system._engine(self.dt)       # the lqr handle of a Riccati system linearises with THIS planner's dt
            Sguide = np.array(self.lqr(self.xguide, np.zeros(self.ncontrols))[0], dtype=np.float64)
            Sguide[:, np.isinf(np.asarray(self.constraints.goal_buffer, dtype=np.float64))] = 0
            ids, _ = run.eng.nn_argmin(self.xguide.reshape(1, -1), S=Sguide, use_ignore=False)
            self._adopt_plan(int(ids[0]))
            run.adopted = True
            if self.printing:
                print("Didn't reach goal.\nTree size: {0}\nETA: {1} s".format(self.tree.size, np.round(self.T, 2)))
            self._prepare_interpolators()

    def _plan_end(self, run):
        eng = run.eng
        if not run.user_sampler and not run.own_stream:
            eng.sync_numpy_global()
        if self.hfactor:
            self.horizon_iters = eng.horizon_iters_state()           # planner.py:421,424: the heuristic's state persists
        self.stats = run.total.as_dict() if run.total is not None else None

        if self.killed or self.tree.size > self.max_nodes:
            # The reference keeps node_seq / x_seq / u_seq / t_seq up to date inside the loop (planner.py:276-281), so
            # after a kill they describe the best plan of THIS tree whenever one was found.
            if self.killed and not run.adopted and run.best_end >= 0:
                self._adopt_plan(run.best_end)
            if self.printing:
                print("Plan update terminated abruptly!")
            self.killed = False
            return False
        return True

    def _attempt_budget(self, rate, remaining):
        """Attempts the next native call may commit: at most 4 waves, at most what fits in about half of the time left
        at the measured throughput, at least a small wave.  Before a rate is known: one wave."""
        cap = 4 * self.wave_size
        if rate is None:
            return min(cap, self.wave_size)
        if not np.isfinite(remaining):
            return cap
        budget = int(min(cap, max(32, 0.5 * rate * max(float(remaining), 0.0))))
        if self.wave_mode == 'synchronous':                        # whole waves only: the wave size defines the result
            budget = max(self.wave_size, budget // self.wave_size * self.wave_size)
        return budget

    def _adopt_plan(self, end_node):
        """The plan that ends in `end_node`: climb to the seed, lay the edges end to end (planner.py:266-281, :319-323)."""
        self.node_seq = self.tree.climb(end_node)
        self.x_seq, self.u_seq = self.tree.trajectory(self.node_seq)
        self.T = len(self.x_seq) * self.dt
        self.t_seq = np.arange(len(self.x_seq)) * self.dt

    def _finish_on_goal(self):
        """
        planner.py:294-303: steer from the plan's last node to the exact goal (force_arrive) and, if that produced
        anything, tack it onto the plan and the tree.  The reference ends this rollout on a wall-clock timeout of
        clip(min_time/2, 0.1, inf) seconds (:402-406); here the same budget becomes a step cap at the reference's
        measured ~0.3 ms per simulated step, which is deterministic.
        """
        budget_s = float(np.clip(self.min_time / 2, 0.1, np.inf))
        max_steps = int(min(max(budget_s / 3e-4, 64), 20000))
        if getattr(self, "force_arrive_max_steps", None):
            max_steps = int(self.force_arrive_max_steps)            # explicit override of the timeout stand-in
        xgoal_seq, ugoal_seq = self._engine.steer_force(self.node_seq[-1], self.goal, max_steps)
        if len(xgoal_seq) == max_steps and self.printing:
            print("(exact goal-convergence timed-out)")
        if len(xgoal_seq) > 0:
            xs, us = list(xgoal_seq), list(ugoal_seq)
            self.tree.add_node(self.node_seq[-1], self.goal, None, xs, us)
            self.node_seq.append(self.tree.size - 1)
            self.x_seq.extend(xs)
            self.u_seq.extend(us)
            self.t_seq = np.arange(len(self.x_seq)) * self.dt

    def _in_goal(self, x):
        """True if x lies strictly inside the goal box (planner.py:442-447)."""
        x = np.asarray(x, dtype=np.float64)
        lo, hi = np.array(self.goal_region, dtype=np.float64).T
        return bool(np.all((lo < x) & (x < hi)))

    def _prepare_interpolators(self):
        """get_state(t) / get_effort(t) over the current plan; held at the last sample beyond its end (planner.py:451-464)."""
        if len(self.x_seq) == 1:
            only, zero = self.x_seq[0], np.zeros(self.ncontrols)
            self.get_state = lambda t: only
            self.get_effort = lambda t: zero
            return
        for name, seq in (("get_state", self.x_seq), ("get_effort", self.u_seq)):
            table = np.array(seq)
            setattr(self, name, scipy.interpolate.interp1d(self.t_seq, table, axis=0, assume_sorted=True,
                                                           bounds_error=False, fill_value=table[-1].copy()))

    # ------------------------------------------------------------------------------------------ setters
    def set_goal(self, goal):
        """New goal state and goal region (planner.py:468-487); update the plan afterwards."""
        if goal is None:
            self.goal = None
            return
        if len(goal) != self.nstates:
            raise ValueError("The goal state must have same dimensionality as state space.")
        self.goal = np.array(goal, dtype=np.float64)
        buff = np.asarray(self.constraints.goal_buffer, dtype=np.float64)
        self.goal_region = list(zip(self.goal - buff, self.goal + buff))
        self.plan_reached_goal = False

    def set_runtime(self, min_time=None, max_time=None, max_nodes=None, sys_time=None):
        """planner.py:491-513; arguments left at None keep their value."""
        if sys_time is not None and not _callable(sys_time):
            raise ValueError("Expected sys_time to be a function.")
        lo = self.min_time if min_time is None else min_time
        hi = self.max_time if max_time is None else max_time
        self.min_time, self.max_time = lo, hi
        if lo > hi:
            raise ValueError("The min_time must be less than or equal to the max_time.")
        if max_nodes is not None:
            self.max_nodes = max_nodes
        if sys_time is not None:
            self.sys_time = sys_time

    def set_resolution(self, horizon=None, dt=None, FPR=None, error_tol=None):
        """planner.py:517-553; arguments left at None keep their value.  horizon may be a (min, max) pair, which
        switches the adaptive-horizon heuristic on (hfactor = 2, starting from one step)."""
        for name, value in (("horizon", horizon), ("dt", dt), ("FPR", FPR)):
            if value is not None:
                setattr(self, name, value)
        if error_tol is not None:
            if np.shape(error_tol) not in [(), (self.nstates,)]:
                raise ValueError("Shape of error_tol must be scalar or length of state.")
            self.error_tol = np.abs(error_tol).astype(np.float64)
        if hasattr(self.horizon, '__contains__'):
            if len(self.horizon) != 2:
                raise ValueError("Expected horizon to be tuple (min, max) or a single scalar.")
            shortest, longest = self.horizon
            if shortest < self.dt:
                raise ValueError("The minimum horizon must be at least as big as dt.")
            if shortest >= longest:
                raise ValueError("A horizon range tuple must be given as (min, max) where min < max.")
            self.hspan = np.divide(self.horizon, self.dt).astype(np.int64)
            self.horizon_iters, self.hfactor = 1, 2
        else:
            if self.horizon < self.dt:
                raise ValueError("The horizon must be at least as big as dt.")
            steps = int(self.horizon / self.dt)
            self.horizon_iters, self.hspan, self.hfactor = steps, (steps, steps), 0

    def set_system(self, dynamics=None, lqr=None, constraints=None, erf=None):
        """
        planner.py:557-592; arguments left at None keep their value, and dynamics and lqr can only be replaced
        together.  All handles must belong to one native system object.
        """
        if dynamics is not None or lqr is not None:
            if not _callable(dynamics):
                raise ValueError("Expected dynamics to be a function.")
            if not _callable(lqr):
                raise ValueError("Expected lqr to be a function.")
            self.dynamics, self.lqr = dynamics, lqr
        if constraints is not None:
            if not isinstance(constraints, Constraints):
                raise ValueError("Expected constraints to be an instance of the Constraints class.")
            self.constraints = constraints
            self.nstates, self.ncontrols = constraints.nstates, constraints.ncontrols
        if erf is not None:
            if not _callable(erf):
                raise ValueError("Expected erf to be a function.")
            self.erf = erf
            self._erf_angles = False                                # (callback mode: not classified yet)
        self._resolve_mode()
        self.plan_reached_goal = False

    def _resolve_mode(self):
        """Native mode when every plugin is a handle of ONE native system object (then the whole loop runs on the device), callback
        mode as soon as one of them is a plain Python callable.  Handles of different native systems, or np.subtract as the erf of
        a native system with angular states, are mistakes and raise."""
        handles = [(getattr(self, "dynamics", None), "dynamics"), (getattr(self, "lqr", None), "lqr")]
        erf = getattr(self, "erf", None)
        cons = getattr(self, "constraints", None)
        if erf is not None and erf is not np.subtract:
            handles.append((erf, "erf"))
        if cons is not None:
            handles.append((cons.is_feasible, "is_feasible"))
        systems = [native_system_of(f, kind) for f, kind in handles if f is not None]
        if any(sysobj is None for sysobj in systems):
            self.callback_mode, self.system = True, None
            return
        if len(set(id(sysobj) for sysobj in systems)) > 1:
            raise ValueError("dynamics, lqr, erf and constraints.is_feasible belong to different native systems.")
        self.callback_mode, self.system = False, (systems[0] if systems else None)
        if self.system is not None and erf is np.subtract and self.system.wrap_dims:
            raise ValueError("This system has angular states; pass its .erf handle.")
        self.plan_reached_goal = False

    def kill_update(self):
        """Asks a running update_plan (another thread's) to stop; it returns False at its next check."""
        self.killed = True

    def unkill(self):
        """Withdraws a kill_update that has not been honoured yet."""
        self.killed = False

    def visualize(self, dx, dy, show=True):
        """Plots the (dx, dy) cross-section of the tree with the current plan highlighted (planner.py:614-624)."""
        if not hasattr(self, "node_seq"):
            print("There is no plan to visualize!")
            return None
        return self.tree.visualize(dx, dy, node_seq=self.node_seq, show=show)


class _PlanRun(object):
    """What update_plan keeps between native calls of one plan."""


def update_plans(jobs):
    """
    Several planners plan at once: `jobs` is a list of dicts, each with the keys `planner`, `x0`, `sample_space` and, optionally,
    update_plan's keyword arguments (goal_bias, guide, xrand_gen, pruning, finish_on_goal, specific_time) plus `seed` and `group`.
    Every planner gets exactly what its own update_plan would give it -- its tree is the one it grows alone from the same sample
    stream -- but the native calls are shared: lqrrt_engine_extend_multi advances all trees of a GROUP in lock step with two kernel
    launches per tick whatever their number (csrc/engine_multi.hpp), which is how one MI355X is filled by planners that each use ~2 %
    of it (a fleet's vehicles, the behaviours of one vehicle, restarts of one query: 16 trees 6 x, 64 trees 10 x the throughput of one).

    Groups: the planners of one device form a group (a job's `group` key splits a device's planners further); every group runs its own
    shared loop on a host thread of its own (the native call releases the GIL), with no exchange between groups -- planners on the 8
    GPUs of a node are 8 independent fleets, which is the one multi-GPU form of this path that scales with the device count
    (DESIGN.md section 8).  Within a group the planners must share the native system type, horizon, max_nodes, wave_size and pruning,
    and run exact waves of an analytic-gain system (ValueError otherwise -- checked for EVERY job before any planner is touched).

    `seed`: the default sampler of that planner draws from np.random.RandomState(seed) and np.random itself is left alone.  Without it
    a planner starts from np.random's current state, as its own update_plan would: unseeded planners therefore grow THE SAME tree from
    the same start (a warning is printed when that happens), and np.random is left where the first unseeded planner's sampler stopped.

    Time: each planner's clock starts when its group's loop does and is read right after every native call, before anyone's wrap-up;
    a call is sized by the smallest remaining budget of the group and the wrap-up of finished plans is deferred until the group's
    loop has ended, so a planner overruns its max_time by at most one shared call (the same bound as a solo update_plan) whatever
    the number of jobs.  A goal hit of ANY planner ends the running call early once some planner of the group is past its min_time
    (the others just continue with the next call).  Returns the list of update_plan's return values.  Not in the reference: its
    Planner plans one tree on one core.
    """
    keys = ("goal_bias", "guide", "xrand_gen", "pruning", "finish_on_goal", "specific_time", "seed", "group")
    jobs = [dict(j) for j in jobs]
    if not jobs:
        return []
    planners = [j.get("planner") for j in jobs]
    if len(set(id(p) for p in planners)) != len(planners):
        raise ValueError("update_plans: a planner appears twice.")
    groups = {}
    for k, (p, j) in enumerate(zip(planners, jobs)):
        unknown = set(j) - set(keys) - {"planner", "x0", "sample_space"}
        if unknown:
            raise ValueError("update_plans: unknown job key(s) %s." % sorted(unknown))
        if not isinstance(p, Planner) or "x0" not in j or "sample_space" not in j:
            raise ValueError("update_plans: every job needs a planner, x0 and sample_space.")
        p._resolve_mode()
        if p.callback_mode:
            raise ValueError("update_plans: planners whose plugins are Python callables plan one by one (update_plan).")
        if p.wave_mode != "exact" or getattr(p.system, "riccati", False):
            raise ValueError("update_plans: exact waves of analytic-gain systems only.")
        # what update_plan itself would refuse, before anything is reset (planner.py:183,196,216)
        xg = j.get("xrand_gen")
        if not (xg is None or type(xg) is int or _callable(xg)):
            raise ValueError("Expected xrand_gen to be None, an integer >= 1,  or a function.")
        if xg is None or type(xg) is int:
            gb = j.get("goal_bias", 0)
            if gb is not None and hasattr(gb, '__contains__') and len(gb) != p.nstates:
                raise ValueError("Expected goal_bias to be scalar or have same length as state.")
            if np.array(j["sample_space"], dtype=np.float64).shape != (p.nstates, 2):
                raise ValueError("Expected sample_space to be list of nstates tuples.")
        groups.setdefault((p.device, j.get("group")), []).append(k)
    for members in groups.values():
        p0, prun0 = planners[members[0]], bool(jobs[members[0]].get("pruning", True))
        if len(members) > 128:
            raise ValueError("update_plans: at most 128 planners per group (split them with the `group` key).")
        for k in members:
            p, j = planners[k], jobs[k]
            same = (type(p.system) is type(p0.system) and tuple(int(v) for v in p.hspan) == tuple(int(v) for v in p0.hspan)
                    and p.hfactor == p0.hfactor and int(p.max_nodes) == int(p0.max_nodes) and p.wave_size == p0.wave_size
                    and bool(j.get("pruning", True)) == prun0)
            if not same:
                raise ValueError("update_plans: the planners of a group must share system type, horizon, max_nodes, wave_size, pruning and device.")
    unseeded = [k for k, j in enumerate(jobs) if j.get("seed") is None and (j.get("xrand_gen") is None or type(j.get("xrand_gen")) is int)]
    if len(unseeded) > 1 and any(getattr(planners[k], "printing", False) for k in unseeded):
        print("update_plans: %d planners use the default sampler without a `seed`: they all start from np.random's current state." % len(unseeded))

    results = [None] * len(jobs)
    runs = {}
    for k, (p, j) in enumerate(zip(planners, jobs)):
        run = p._plan_begin(j["x0"], j["sample_space"], j.get("goal_bias", 0), j.get("guide"), j.get("xrand_gen"),
                            bool(j.get("pruning", True)), j.get("finish_on_goal", False), j.get("specific_time"), seed=j.get("seed"))
        if run is None:
            results[k] = False
        else:
            runs[k] = run

    def group_loop(members):
        mine = [(k, planners[k], runs[k]) for k in members if k in runs]
        if not mine:
            return
        p0, prun0 = mine[0][1], bool(jobs[mine[0][0]].get("pruning", True))
        # every planner's clock starts when the shared loop does: the others' set-up is not part of its time budget
        for _, p, run in mine:
            run.time_start, run.time_elapsed = p.sys_time(), 0
        active = list(mine)
        try:
            while active:
                budget = min(p._plan_budget(run) for _, p, run in active)
                stop = any(run.time_elapsed >= run.min_time for _, _, run in active)
                t_call = time.perf_counter()
                sts = Engine.extend_multi([run.eng for _, _, run in active], p0.wave_size, max_attempts=budget, node_limit=int(p0.max_nodes),
                                          pruning=prun0, stop_on_goal=bool(stop))
                dt_call = time.perf_counter() - t_call
                # clocks and exit rules of ALL planners first (cheap), wrap-ups after the loop
                active = [(k, p, run) for (k, p, run), st in zip(active, sts) if not p._plan_after_call(run, st, dt_call, wrap_up=False)]
        except Exception:
            # a failed native call leaves every engine of the group mid-wave (include/lqrrt_hip.h): nothing of these trees may be read
            for _, p, run in mine:
                p.tree._e, p.tree._snap = None, None
                p._engine_key = None                                 # the next plan builds a fresh engine
            raise
        for _, p, run in mine:
            p._plan_wrap_up(run)

    order = sorted(groups, key=lambda g: min(groups[g]))
    if len(order) == 1:
        group_loop(groups[order[0]])
    else:
        import threading
        errors = []

        def guarded(members):
            try:
                group_loop(members)
            except Exception as ex:                                  # surfaced on the caller's thread below
                errors.append(ex)
        threads = [threading.Thread(target=guarded, args=(groups[g],)) for g in order]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    synced = False
    for k in sorted(runs):
        run = runs[k]
        if not run.user_sampler and not run.own_stream:
            if synced:
                run.own_stream = True                                # (np.random: left where the first unseeded planner's sampler stopped)
            synced = True
        results[k] = planners[k]._plan_end(run)
    return results


def _add_stats(a, b):
    out = nat.ExtendStats()
    for k, _ in nat.ExtendStats._fields_:
        setattr(out, k, getattr(a, k) + getattr(b, k))
    out.tree_size = b.tree_size
    out.stop_reason = b.stop_reason
    out.candidates = b.candidates
    return out
