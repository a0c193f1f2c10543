"""
`lqrrt` -- the reference's package name (lqrrt/__init__.py:1-2 exports Constraints and Planner), served by the
MI355X build: code written against jnez71/lqRRT keeps its `import lqrrt` line and gets lqrrt_amd's classes.
Tree and the native problem plugins (`systems`) are exported as well.
"""
from lqrrt_amd import Constraints, Planner, Tree, systems, update_plans  # noqa: F401

__all__ = ["Constraints", "Planner", "Tree", "systems", "update_plans"]
