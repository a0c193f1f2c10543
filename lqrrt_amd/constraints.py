"""
Constraints -- same class surface as the reference's lqrrt/constraints.py:17-61: state and
effort dimensionality, the goal buffer, and the feasibility function: the `.is_feasible`
plugin handle of a native system (lqrrt_amd.systems; the collision sweep then runs inside the
HIP steer kernel, one problem per wavefront) or any Python callable (callback mode).
"""
import numpy as np

from .systems import native_system_of


class Constraints:
    """
    To initialize, provide...

    nstates: The dimensionality of the state space.

    ncontrols: The dimensionality of the effort space.

    goal_buffer: Half-edge lengths of box defining goal region.

    is_feasible: Function is_feasible(x, u) -> bool: the `.is_feasible` handle of an
                 lqrrt_amd.systems object (evaluated on the GPU) or any Python callable.

    """

    def __init__(self, nstates, ncontrols, goal_buffer, is_feasible):
        self.nstates = nstates
        self.ncontrols = ncontrols
        self.set_buffers(goal_buffer)
        self.set_feasibility_function(is_feasible)

    def set_buffers(self, goal_buffer=None):
        """Arguments not given are not modified (constraints.py:39-49)."""
        if goal_buffer is not None:
            if len(goal_buffer) == self.nstates:
                self.goal_buffer = np.abs(goal_buffer).astype(np.float64)
            else:
                raise ValueError("The goal_buffer must have same dimensionality as state.")

    def set_feasibility_function(self, is_feasible):
        """constraints.py:53-61.  The `.is_feasible` handle of a native system (the sweep then runs inside the HIP steer kernel) or
        any Python callable is_feasible(x, u) -> bool (the planner then runs in callback mode, lqrrt_amd/callback.py)."""
        if not hasattr(is_feasible, '__call__'):
            raise ValueError("Expected is_feasible to be a function.")
        system = native_system_of(is_feasible, "is_feasible")
        if system is not None and (system.nstates != self.nstates or system.ncontrols != self.ncontrols):
            raise ValueError("The feasibility plugin is for a %d-state/%d-effort system." % (system.nstates, system.ncontrols))
        self.is_feasible = is_feasible
        self.system = system

    # -- batched evaluation (build-only additions; the reference calls is_feasible once per state) ------
    def feasible_batch(self, X, U=None):
        """is_feasible for every row of X (and U, zeros when omitted) in one device launch (a Python callable: row by row)."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if self.system is None:
            U = np.zeros((len(X), self.ncontrols)) if U is None else np.atleast_2d(np.asarray(U, dtype=np.float64))
            return np.array([bool(self.is_feasible(x, u)) for x, u in zip(X, U)], dtype=bool)
        return self.system._engine().feasible_batch(X, None if U is None else np.atleast_2d(np.asarray(U, dtype=np.float64)))

    def first_infeasible(self, X, U=None):
        """
        Index of the first infeasible row of X, or -1: the plan re-evaluation of the ROS node
        (lqrrt_node.py:806-824 walks the next seconds of the current plan, velocities zeroed by the
        caller, and reacts to the first state that now collides with the map).
        """
        ok = self.feasible_batch(X, U)
        bad = np.nonzero(~np.asarray(ok, dtype=bool))[0]
        return int(bad[0]) if len(bad) else -1
