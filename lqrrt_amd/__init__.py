"""
lqrrt_amd -- MI355X-native expansion engine behind the jnez71/lqRRT Python API.

    import lqrrt_amd as lqrrt
    boat = lqrrt.systems.BoatAdvanced()
    constraints = lqrrt.Constraints(nstates=6, ncontrols=3, goal_buffer=boat.goal_buffer,
                                    is_feasible=boat.is_feasible)
    planner = lqrrt.Planner(boat.dynamics, boat.lqr, constraints, horizon=2, dt=0.1, FPR=0.9,
                            error_tol=boat.error_tol, erf=boat.erf, goal0=boat.goal)
    planner.update_plan(boat.x0, boat.sample_space, goal_bias=boat.goal_bias)

Exports mirror lqrrt/__init__.py:1-2 of the reference (Constraints, Planner) plus Tree and
the native problem plugins (systems), and update_plans (several planners through shared native calls: one GPU, many
trees -- planner.py).  Importing works without a GPU; computing does not.
"""
from .constraints import Constraints
from .planner import Planner, update_plans
from .tree import Tree
from . import systems
from . import dare

__all__ = ["Constraints", "Planner", "Tree", "systems", "dare", "update_plans"]
