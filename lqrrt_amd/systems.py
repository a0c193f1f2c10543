"""
Native problem plugins.

In the reference every problem-specific piece -- dynamics(x,u,dt), lqr(x,u), erf(xgoal,x),
Constraints.is_feasible(x,u) -- is a Python callable defined in a demo script
(planner.py:35-59, constraints.py:27).  A GPU cannot call Python, so lqrrt_amd ships the
reference's demo problems as device code (csrc/systems.hpp) and this module provides, per
problem, a *system object* that

  * builds the constant tables on the host (NumPy: hull grid, thruster map, drag, gains ...)
    exactly as the demo's definition section does and packs them into the parameter block
    the device code reads (layout documented per class; mirrored in csrc/systems.hpp);
  * exposes `.dynamics`, `.lqr`, `.erf`, `.is_feasible` as *plugin handles*: callables with
    the reference's signatures, evaluated on the GPU through the batched C-ABI operators, so
    user code written against the reference (e.g. the demos' tracking simulation, which calls
    dynamics/erf/lqr after planning) keeps working;
  * is recognised by lqrrt_amd.Planner / Constraints, which then run the whole extend path
    on the device.  Plain Python callables are rejected loudly: there is no CPU path.

  BoatAdvanced      demos/demo_boat_advanced.py:21-238
  BoatIntermediate  demos/demo_boat_intermediate.py:24-221
  BoatNovice        demos/demo_boat_novice.py:21-175
  Car               demos/demo_car.py:26-193
  DoublePendulum    demos/demo_pendulum.py:23-165
  BoatNoviceLqr     demos/demo_boat_novice.py dynamics + the same Riccati lqr, linearised about (x, 0): 6 states, 3 controls
  PendulumLqr       demos/demo_pendulum.py dynamics + the lqr of the API contract (planner.py:39-42): Riccati gains
  DoubleIntegrator  BASELINE.json config 5 (not in the reference)
"""
import ctypes as C
import os

import numpy as np
import numpy.linalg as npl

from . import _native as nat


# --------------------------------------------------------------------------- table builders

def hull_grid(length, width, buffer, spacing):
    """2 x V lattice of body-frame hull points (demo_boat_advanced.py:60-68)."""
    half_l, half_w = (length + buffer) / 2, (width + buffer) / 2
    gx, gy = np.mgrid[slice(-half_l, half_l + spacing, spacing), slice(-half_w, half_w + spacing, spacing)]
    return np.ascontiguousarray(np.vstack((gx.ravel(), gy.ravel())), dtype=np.float64)


def obstacle_grid(seed, goal, x0, clearance, spacing=12, lo=5, hi=60):
    """
    The demos' jittered 6x6 circle field (demo_boat_advanced.py:190-202): centres rounded to cm,
    radius 1; lattice points closer than `clearance` to start or goal become the placeholder
    [-9999,-9999,-9999] (negative radius: can never collide).  RandomState(seed).rand(2) per point
    reproduces `np.random.seed(seed)` followed by the demo's draws.
    """
    rs = np.random.RandomState(seed)
    gx, gy = np.mgrid[slice(lo, hi + spacing, spacing), slice(lo, hi + spacing, spacing)]
    gx, gy = gx.ravel(), gy.ravel()
    goal = np.asarray(goal, dtype=np.float64)
    x0 = np.asarray(x0, dtype=np.float64)
    obs = np.full((gx.size, 3), -9999.0)
    for i in range(gx.size):
        p = np.round([gx[i], gy[i]] + 3 * (rs.rand(2) - 0.5), 2)
        if npl.norm(p - goal[:2]) > clearance and npl.norm(np.array(p - x0[:2])) > clearance:
            obs[i] = (p[0], p[1], 1.0)
    return obs


# --------------------------------------------------------------------------- plugin handles

# Speed (m/s) above which the boats' heading torque is evaluated in its one-atan2 form (csrc/systems.hpp rudder_term); the
# environment variable is a measurement lever (A/B of the threshold on one box), the attribute `torque_vmin` of a system the API.
_TORQUE_VMIN = float(os.environ.get("LQRRT_TORQUE_VMIN", "0.01"))


class _Plugin(object):
    """A callable bound to a native system; `kind` in dynamics|lqr|erf|is_feasible."""

    def __init__(self, system, kind):
        self.system = system
        self.kind = kind
        self.__name__ = kind

    def __call__(self, *args):
        return getattr(self.system, "_eval_" + self.kind)(*args)

    def __repr__(self):
        return "<lqrrt_amd %s plugin of %s>" % (self.kind, type(self.system).__name__)


def plugin_system(fn, kind):
    """Returns the native system behind a plugin handle, or raises the boundary's ValueError."""
    if isinstance(fn, _Plugin) and fn.kind == kind:
        return fn.system
    raise ValueError(
        "Expected %s to be a native plugin (e.g. lqrrt_amd.systems.BoatAdvanced().%s): lqrrt_amd "
        "evaluates the problem on the GPU and cannot call arbitrary Python functions.  A problem of your own is one C++ header "
        "(dynamics / lqr / erf / is_feasible, see examples/user_system/unicycle.hpp) compiled in by tools/build_user_system.py "
        "and wrapped by lqrrt_amd.systems.UserSystem -- INTEGRATION.md section 5." % (kind, kind))


def native_system_of(fn, kind):
    """The native system behind a plugin handle of that kind, or None for any other callable."""
    return fn.system if isinstance(fn, _Plugin) and fn.kind == kind else None


class NativeSystem(object):
    """Base class: owns the parameter block and a small engine used to evaluate the handles."""
    model = None
    nstates = ncontrols = 0
    wrap_dims = ()
    S = None          # dense constant S (n x n) or None = identity

    def __init__(self):
        self.dynamics = _Plugin(self, "dynamics")
        self.lqr = _Plugin(self, "lqr")
        self.erf = _Plugin(self, "erf")
        self.is_feasible = _Plugin(self, "is_feasible")
        self.vps = np.zeros((2, 0))
        self.obs = np.zeros((0, 3))
        self.obs_stride = 3
        self.ogrid = None         # optional occupancy grid: dict(grid=int8 [rows][cols], origin=(x,y), cpm, threshold)
        self.revision = 0         # bumped whenever the world changes; engines re-upload on their next use
        self._ops = None
        self._ops_dt = None

    # -- packing -------------------------------------------------------------------------------
    def params(self):
        raise NotImplementedError

    def desc(self):
        """lqrrt_system_desc (include/lqrrt_hip.h) + the arrays it points to (keep alive)."""
        d = nat.SystemDesc()
        p = np.ascontiguousarray(self.params(), dtype=np.float64)
        if p.size > nat.MAX_PARAMS:
            raise ValueError("too many parameters")
        d.model, d.nstates, d.ncontrols, d.n_params = self.model, self.nstates, self.ncontrols, p.size
        for i, v in enumerate(p):
            d.params[i] = v
        vps = np.ascontiguousarray(self.vps, dtype=np.float64)
        obs = np.ascontiguousarray(self.obs, dtype=np.float64).reshape(-1, self.obs_stride)
        d.n_vertices, d.n_obstacles, d.obs_stride = vps.shape[1], obs.shape[0], self.obs_stride
        d.vps = vps.ctypes.data_as(C.POINTER(C.c_double)) if vps.size else None
        d.obs = obs.ctypes.data_as(C.POINTER(C.c_double)) if obs.size else None
        grid = None
        if self.ogrid is not None:
            grid = np.ascontiguousarray(self.ogrid["grid"], dtype=np.int8)
            d.ogrid = grid.ctypes.data_as(C.POINTER(C.c_int8))
            d.og_rows, d.og_cols = grid.shape
            d.og_origin[0], d.og_origin[1] = (float(v) for v in self.ogrid["origin"])
            d.og_cpm = float(self.ogrid["cpm"])
            d.og_threshold = float(self.ogrid["threshold"])
        return d, (vps, obs, p, grid)

    def set_occupancy_grid(self, grid, origin, resolution=None, cpm=None, threshold=90.0, vps=None):
        """
        Switches the collision model of a planar vehicle to the ROS node's occupancy-grid test
        (demos/lqrrt_ros/nodes/lqrrt_node.py:719-745; grid as in nav_msgs/OccupancyGrid: -1 unknown, 0..100).
        Optionally replaces the hull points (e.g. behaviors/params.py's 0.1 m lattice).  Engines that already
        exist (a Planner's, the one behind the plugin handles) pick the new map up on their next use, which is
        how the node swaps maps between plans (lqrrt_node.py:65, 1120-1140).
        """
        if self.model not in (nat.MODEL_BOAT_ADVANCED, nat.MODEL_BOAT_INTERMEDIATE, nat.MODEL_CAR, nat.MODEL_ROS_BOAT):
            raise ValueError("occupancy-grid feasibility is defined for the hull-sweeping vehicles")
        if (resolution is None) == (cpm is None):
            raise ValueError("give exactly one of resolution (m/cell) or cpm (cells/m)")
        grid = np.asarray(grid)
        if grid.ndim != 2:
            raise ValueError("grid must be 2-D [rows][cols]")
        self.ogrid = dict(grid=grid.astype(np.int8), origin=(float(origin[0]), float(origin[1])),
                          cpm=float(cpm) if cpm is not None else 1.0 / float(resolution), threshold=float(threshold))
        if vps is not None:
            self.vps = np.ascontiguousarray(vps, dtype=np.float64).reshape(2, -1)
        self.revision += 1

    def clear_occupancy_grid(self):
        """Back to the obstacle-table collision model."""
        self.ogrid = None
        self.revision += 1

    def set_obstacles(self, obs):
        """Replaces the obstacle table ([x, y, r] rows; boxes [lo(3), hi(3)] for the double integrator)."""
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.obs_stride)
        self.obs = obs
        self.revision += 1

    def Smatrix(self):
        return np.eye(self.nstates) if self.S is None else np.asarray(self.S, dtype=np.float64)

    # -- host evaluation of the handles through the device operators ----------------------------
    def _engine(self, dt=None):
        from .engine import Engine
        if self._ops is None:
            self._ops = Engine(self, capacity=64, max_wave=64)
        self._ops.sync_geometry()
        if dt is not None and dt != self._ops_dt:
            self._ops.set_resolution(dt=dt, FPR=0.0, horizon_iters=1, error_tol=np.zeros(self.nstates),
                                     goal=None, goal_buffer=None)
            self._ops_dt = dt
        return self._ops

    def _eval_dynamics(self, x, u, dt):
        return self._engine(dt).dynamics_batch(np.atleast_2d(x), np.atleast_2d(u))[0]

    def _eval_lqr(self, x, u):
        K = self._engine().gain_batch(np.atleast_2d(x), np.atleast_2d(np.asarray(u, dtype=np.float64)))[0]
        return (self.Smatrix(), K)

    def _eval_erf(self, xgoal, x):
        return self._engine().erf_batch(np.atleast_2d(xgoal), np.atleast_2d(x))[0]

    def _eval_is_feasible(self, x, u):
        return bool(self._engine().feasible_batch(np.atleast_2d(x), np.atleast_2d(np.asarray(u, dtype=np.float64)))[0])


# --------------------------------------------------------------------------- boats

class _Boat(NativeSystem):
    nstates, ncontrols = 6, 3
    wrap_dims = (2,)

    def _objectives(self, goal_buffer_xy, tol_div, obstacle_seed):
        self.x0 = np.array([0, 0, np.deg2rad(0), 0, 0, 0])
        self.goal = [40, 40, np.deg2rad(90), 0, 0, 0]
        self.goal_buffer = [goal_buffer_xy, goal_buffer_xy, np.inf, np.inf, np.inf, np.inf]
        self.error_tol = np.copy(self.goal_buffer) / tol_div
        self.boat_length = 210 * 0.0254
        self.boat_width = 96 * 0.0254
        self.obs = obstacle_grid(obstacle_seed, self.goal, self.x0, 2 * self.boat_length)


class BoatAdvanced(_Boat):
    """
    4-thruster boat with per-thruster saturation and a planning speed box.
    params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 B[3][4] | 21 invB[4][3] | 33 thrust_max[4] |
            37 rudder | 38 velmax_pos0 | 39 velmax_neg0 | 40 kp[3] | 43 kd[3] |
            46 velmax_pos_plan[3] | 49 velmax_neg_plan[3] | 52 torque_vmin^2
    `torque_vmin` (m/s, default 0.01): above this speed the heading torque of demo_boat_advanced.py:101-108,
    rudder * wrap(atan2(R v) - h), is evaluated as rudder * atan2(v_y, v_x) of the body-frame velocity (the same angle, one
    elementary function instead of three); at or below it -- where the dynamics amplify rounding differences -- and for
    v_x < 0 the reference's own sequence runs.  np.inf: the reference's sequence everywhere (csrc/systems.hpp rudder_term).
    """
    torque_vmin = _TORQUE_VMIN
    model = nat.MODEL_BOAT_ADVANCED
    plan_kwargs = dict(horizon=2, dt=0.1, FPR=0.9)          # demo_boat_advanced.py:245-249

    def __init__(self, obstacle_seed=0, obstacles=None):
        NativeSystem.__init__(self)
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = np.array([2.5, 1, 0.7])
        self.velmax_neg = np.array([-0.8, -1, -0.7])
        self.thrust_max = np.array([220, 220, 220, 220])
        positions = np.array([[-1.9000, 1.0000, -0.0123], [-1.9000, -1.0000, -0.0123],
                              [1.6000, 0.6000, -0.0123], [1.6000, -0.6000, -0.0123]])
        directions = np.array([[0.7071, 0.7071, 0.0000], [0.7071, -0.7071, 0.0000],
                               [0.7071, -0.7071, 0.0000], [0.7071, 0.7071, 0.0000]])
        levers = np.cross(positions, directions)
        self.B = np.concatenate((directions.T, levers.T))[[0, 1, 5]]      # demo_boat_advanced.py:44
        self.invB = npl.pinv(self.B)                                      # :45
        Fx_max = self.B.dot(self.thrust_max * [1, 1, 1, 1])[0]
        Fy_max = self.B.dot(self.thrust_max * [1, -1, -1, 1])[1]
        Mz_max = self.B.dot(self.thrust_max * [-1, 1, -1, 1])[2]
        self.D_pos = np.abs([Fx_max, Fy_max, Mz_max] / self.velmax_pos)
        self.D_neg = np.abs([Fx_max, Fy_max, Mz_max] / self.velmax_neg)
        self._objectives(8, 8, obstacle_seed)
        if obstacles is not None:
            self.obs = np.asarray(obstacles, dtype=np.float64).reshape(-1, 3)
        self.vps = hull_grid(self.boat_length, self.boat_width, 0.25, 1)
        self.magic_rudder = 4000
        self.kp = np.diag([120, 20, 0])
        self.kd = np.diag([120, 20, 0])
        self.velmax_pos_plan = np.array([1.1, 0.4, 0.2])
        self.velmax_neg_plan = np.array([-0.65, -0.4, -0.2])
        self.sample_space = [(self.x0[0], self.goal[0]), (self.x0[1], self.goal[1]), (0, 0),
                             (0.9 * self.velmax_pos_plan[0], self.velmax_pos_plan[0]),
                             (-abs(self.velmax_neg_plan[1]), self.velmax_pos_plan[1]),
                             (-abs(self.velmax_neg_plan[2]), self.velmax_pos_plan[2])]
        self.goal_bias = [0.2, 0.2, 0, 0, 0, 0]

    def params(self):
        return np.concatenate((self.invM, self.D_pos, self.D_neg, self.B.ravel(), self.invB.ravel(),
                               self.thrust_max, [self.magic_rudder, self.velmax_pos[0], self.velmax_neg[0]],
                               np.diag(self.kp), np.diag(self.kd), self.velmax_pos_plan, self.velmax_neg_plan,
                               [self.torque_vmin ** 2]))


class BoatIntermediate(_Boat):
    """
    Wrench-saturated boat with the heading "rudder" and a dense hull grid.
    params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 u_max[3] | 12 rudder | 13 velmax_pos0 |
            14 velmax_neg0 | 15 kp[3] | 18 kd[3] | 21 torque_vmin^2 (see BoatAdvanced)
    """
    torque_vmin = _TORQUE_VMIN
    model = nat.MODEL_BOAT_INTERMEDIATE
    plan_kwargs = dict(horizon=2, dt=0.1, FPR=0.5)          # demo_boat_intermediate.py:228-232

    def __init__(self, obstacle_seed=0, obstacles=None):
        NativeSystem.__init__(self)
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = [1.1, 0.45, 0.2]
        self.velmax_neg = [0.68, 0.45, 0.2]
        thrust_max, thrust_lever = 220, 2.15
        self.u_max = np.array([2 * np.sqrt(2) * thrust_max, 0.2 * np.sqrt(2) * thrust_max,
                               4 * thrust_lever * thrust_max])
        self.D_pos = np.abs(self.u_max / self.velmax_pos)
        self.D_neg = np.abs(self.u_max / self.velmax_neg)
        self._objectives(8, 8, obstacle_seed)
        if obstacles is not None:
            self.obs = np.asarray(obstacles, dtype=np.float64).reshape(-1, 3)
        self.vps = hull_grid(self.boat_length, self.boat_width, 2, 0.5)
        self.rudder = 5000
        self.kp = np.diag([120, 120, 0])
        self.kd = np.diag([120, 120, 0])
        self.sample_space = [(self.x0[0], self.goal[0]), (self.x0[1], self.goal[1]), (0, 0),
                             (0.9 * self.velmax_pos[0], self.velmax_pos[0]),
                             (-self.velmax_neg[1], self.velmax_pos[1]),
                             (-self.velmax_neg[2], self.velmax_pos[2])]
        self.goal_bias = [0.2, 0.2, 0, 0, 0, 0]

    def params(self):
        return np.concatenate((self.invM, self.D_pos, self.D_neg, self.u_max,
                               [self.rudder, self.velmax_pos[0], self.velmax_neg[0]],
                               np.diag(self.kp), np.diag(self.kd), [self.torque_vmin ** 2]))


class BoatNovice(_Boat):
    """
    Holonomic boat, centre-point collision model.
    params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 u_max[3] | 12 kp[3] | 15 kd[3] | 18 boat_length/2
    """
    model = nat.MODEL_BOAT_NOVICE
    plan_kwargs = dict(horizon=2, dt=0.1, FPR=0.5)          # demo_boat_novice.py:182-186

    def __init__(self, obstacle_seed=0, obstacles=None):
        NativeSystem.__init__(self)
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = [1.1, 0.45, 0.2]
        self.velmax_neg = [0.68, 0.45, 0.2]
        thrust_max, thrust_lever = 220, 2.15
        self.u_max = np.array([2 * np.sqrt(2) * thrust_max, 2 * np.sqrt(2) * thrust_max,
                               4 * thrust_lever * thrust_max])
        self.D_pos = np.abs(self.u_max / self.velmax_pos)
        self.D_neg = np.abs(self.u_max / self.velmax_neg)
        self._objectives(6, 2, obstacle_seed)
        if obstacles is not None:
            self.obs = np.asarray(obstacles, dtype=np.float64).reshape(-1, 3)
        self.kp = np.diag([120, 120, 350])
        self.kd = np.diag([120, 120, 100])
        self.sample_space = [(self.x0[0], self.goal[0]), (self.x0[1], self.goal[1]), (-np.pi, np.pi),
                             (0.5 * self.velmax_pos[0], self.velmax_pos[0]),
                             (-self.velmax_neg[1], self.velmax_pos[1]),
                             (-self.velmax_neg[2], self.velmax_pos[2])]
        self.goal_bias = [0.5, 0.5, 0, 0, 0, 0]

    def params(self):
        return np.concatenate((self.invM, self.D_pos, self.D_neg, self.u_max, np.diag(self.kp),
                               np.diag(self.kd), [self.boat_length / 2]))


class RosBoat(_Boat):
    """
    The boat of the reference's ROS package with its three behaviours
    (demos/lqrrt_ros/behaviors/{params,boat,car,escape}.py):
      'boat'   holonomic, optional focus point to stare at, even-downscaling thruster saturation
      'car'    heading along the velocity line, per-thruster clipping, no reversing, S = diag(1,1,1,0,0,0)
      'escape' holonomic, no heading term, even downscaling
    They are meant to be planned with horizon=(0.1, 3) (adaptive-horizon heuristic) and FPR=0, and with the
    occupancy-grid feasibility of the node (set_occupancy_grid); without a map everything is feasible.
    params: 0 invM[3] | 3 D_pos[3] | 6 D_neg[3] | 9 B[3][4] | 21 invB[4][3] | 33 thrust_max[4] | 37 rudder |
            38 rudder mode | 39 focus[2] | 41 saturation mode | 42 no-reverse | 43 kp[3] | 46 kd[3] |
            49 torque_vmin^2 ('car' behaviour: see BoatAdvanced)
    """
    torque_vmin = _TORQUE_VMIN
    model = nat.MODEL_ROS_BOAT
    plan_kwargs = dict(horizon=(0.1, 3), dt=0.1, FPR=0)     # behaviors/params.py:31-33

    def __init__(self, behavior="boat", focus=None):
        NativeSystem.__init__(self)
        if behavior not in ("boat", "car", "escape"):
            raise ValueError("behavior must be 'boat', 'car' or 'escape'")
        self.behavior = behavior
        self.focus = None if focus is None else np.array(focus, dtype=np.float64)
        m, I = 350, 400                                                     # params.py:40-42
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = np.array([1.2, 0.6, 0.22])
        self.velmax_neg = np.array([-0.6, -0.6, -0.22])
        self.thrust_max = np.array([220, 220, 220, 220])
        positions = np.array([[-1.9000, 1.0000, -0.0123], [-1.9000, -1.0000, -0.0123],
                              [1.6000, 0.6000, -0.0123], [1.6000, -0.6000, -0.0123]])
        directions = np.array([[0.7071, 0.7071, 0.0000], [0.7071, -0.7071, 0.0000],
                               [0.7071, -0.7071, 0.0000], [0.7071, 0.7071, 0.0000]])
        levers = np.cross(positions, directions)
        self.B = np.concatenate((directions.T, levers.T))[[0, 1, 5]]      # params.py:66
        self.invB = npl.pinv(self.B)
        Fx_max = self.B.dot(self.thrust_max * [1, 1, 1, 1])[0]
        Fy_max = self.B.dot(self.thrust_max * [1, -1, -1, 1])[1]
        Mz_max = self.B.dot(self.thrust_max * [-1, 1, -1, 1])[2]
        self.D_pos = np.abs([Fx_max, Fy_max, Mz_max] / self.velmax_pos)
        self.D_neg = np.abs([Fx_max, Fy_max, Mz_max] / self.velmax_neg)
        self.boat_length = 210 * 0.0254
        self.boat_width = 96 * 0.0254
        self.vps = hull_grid(self.boat_length, self.boat_width, 0.15, 0.1)  # params.py:83-93 (1512 points)
        self.obs = np.zeros((0, 3))
        real_tol = [0.5, 0.5, np.deg2rad(10), np.inf, np.inf, np.inf]
        free_radius = 6
        if behavior == "boat":
            self.rudder, self.rudder_mode, self.sat_mode, self.no_reverse = 8000, (1 if focus is not None else 0), 0, 0
            self.kp, self.kd = np.diag([250, 250, 2500]), np.diag([5, 5, 0.001])
            self.goal_buffer = [real_tol[0], real_tol[1], real_tol[2], 10, 10, 6]
            self.error_tol = np.copy(self.goal_buffer)
        elif behavior == "car":
            self.rudder, self.rudder_mode, self.sat_mode, self.no_reverse = 6000, 2, 1, 1
            self.kp, self.kd = np.diag([150, 150, 0]), np.diag([150, 5, 0])
            self.S = np.diag([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])                # car.py:65
            self.goal_buffer = [0.5 * free_radius, 0.5 * free_radius, np.inf, np.inf, np.inf, np.inf]
            self.error_tol = np.copy(self.goal_buffer) / 10
        else:
            self.rudder, self.rudder_mode, self.sat_mode, self.no_reverse = 0, 0, 0, 0
            self.kp, self.kd = np.diag([150, 150, 2000]), np.diag([120, 120, 0.01])
            self.goal_buffer = [free_radius, free_radius, np.inf, np.inf, np.inf, np.inf]
            self.error_tol = np.copy(self.goal_buffer)
        self.x0 = np.zeros(6)
        self.goal = [30, 20, np.deg2rad(45), 0, 0, 0]
        self.sample_space = self.gen_ss(self.x0, self.goal)
        self.goal_bias = [0.3, 0.3, 0, 0, 0, 0]

    def gen_ss(self, seed, goal, buff=None):
        """Sample space for a seed and goal state (boat.py:86-96, car.py:84-94, escape.py:67-77)."""
        vp, vn = self.velmax_pos, self.velmax_neg
        if self.behavior == "escape":
            buff = 40 if buff is None else buff
            return [(seed[0] - buff, seed[0] + buff), (seed[1] - buff, seed[1] + buff), (seed[2], seed[2]),
                    (-abs(vn[0]), vp[0]), (-abs(vn[1]), vp[1]), (-abs(vn[2]), vp[2])]
        buff = [10] * 4 if buff is None else buff
        vx = (0.9 * vp[0], vp[0]) if self.behavior == "car" else (-abs(vn[0]), vp[0])
        return [(min([seed[0], goal[0]]) - buff[0], max([seed[0], goal[0]]) + buff[1]),
                (min([seed[1], goal[1]]) - buff[2], max([seed[1], goal[1]]) + buff[3]),
                (-np.pi, np.pi), vx, (-abs(vn[1]), vp[1]), (-abs(vn[2]), vp[2])]

    def params(self):
        focus = self.focus[:2] if self.focus is not None else [0.0, 0.0]
        return np.concatenate((self.invM, self.D_pos, self.D_neg, self.B.ravel(), self.invB.ravel(), self.thrust_max,
                               [self.rudder, self.rudder_mode], focus, [self.sat_mode, self.no_reverse],
                               np.diag(self.kp), np.diag(self.kd), [self.torque_vmin ** 2]))


# --------------------------------------------------------------------------- car

class Car(NativeSystem):
    """
    Nonholonomic car (no sway state).
    params: 0 invM[2] | 2 D[2] | 4 u_lo[2] | 6 u_hi[2] | 8 velmax0 | 9 kp[2] | 11 kd[2]
    """
    model = nat.MODEL_CAR
    nstates, ncontrols = 5, 2
    wrap_dims = (2,)
    plan_kwargs = dict(horizon=5, dt=0.1, FPR=0)            # demo_car.py:200-204 (FPR defaulted)

    def __init__(self, obstacle_seed=0, obstacles=None):
        NativeSystem.__init__(self)
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / I])
        self.velmax = [1.1, 1]
        self.u_max = np.array([650, 1800])
        self.D = np.abs(self.u_max / self.velmax)
        self.vps = hull_grid(6, 3, 2, 0.5)                  # demo_car.py:77-90
        self.kp = np.diag([120, 600])
        self.kd = np.diag([120, 600])
        self.x0 = np.array([0, 0, np.deg2rad(0), 0, 0])
        self.goal = [40, 40, np.deg2rad(90), 0, 0]
        self.goal_buffer = [8, 8, np.inf, np.inf, np.inf]
        self.error_tol = np.copy(self.goal_buffer) / 2
        self.obs = np.array([[20, 20, 5], [10, 30, 2], [40, 10, 3]], dtype=np.float64)  # 'some'
        if obstacles is not None:
            self.obs = np.asarray(obstacles, dtype=np.float64).reshape(-1, 3)
        buff = 40
        self.sample_space = [(self.goal[0] - buff, self.goal[0] + buff),
                             (self.goal[0] - buff, self.goal[1] + buff),       # sic, demo_car.py:186
                             (-np.pi, np.pi), (0.9 * self.velmax[0], self.velmax[0]),
                             (-self.velmax[1], self.velmax[1])]
        self.goal_bias = [0.5, 0.5, 0, 0, 0]

    def params(self):
        u_lo = [-self.u_max[0] / 10, -self.u_max[1]]        # demo_car.py:57
        return np.concatenate((self.invM, self.D, u_lo, self.u_max, [self.velmax[0]],
                               np.diag(self.kp), np.diag(self.kd)))


# --------------------------------------------------------------------------- double pendulum

class DoublePendulum(NativeSystem):
    """
    BASELINE.json config 1 ("pendulum"): the reference's demo is a 4-state double pendulum.
    params: 0 a | 1 b2 | 2 m1 L1^2 | 3 m1 L0 L1 | 4 g(m0+m1)L0 | 5 m1 g L1 | 6 d[2] | 8 b[2] |
            10 c[2] | 12 umax | 13 umax_plan | 14 K[4]
    The demo passes horizon=0, which the reference rejects (planner.py:548-553); this build
    plans it with horizon=0.05 s (50 steps of dt=1 ms).
    """
    model = nat.MODEL_PENDULUM
    nstates, ncontrols = 4, 1
    wrap_dims = (0, 1)
    plan_kwargs = dict(horizon=0.05, dt=0.001, FPR=0.5)

    def __init__(self, obstacle_seed=0):
        NativeSystem.__init__(self)
        self.L = [1, 0.5]
        self.m = [5, 5]
        self.g = 9.81
        self.d = [0.4, 0.4]
        self.b = [0.01, 0.01]
        self.c = [0.1, 0.1]
        self.umax = np.inf
        self.umax_plan = 0.75 * self.umax
        self.K = np.array([[10, 200, 0, 0]], dtype=np.float64)
        self.x0 = np.array([-np.pi / 2, 0, 0, 0])
        self.goal = [np.pi / 2, 0, 0, 0]
        self.goal_buffer = [np.deg2rad(1), np.deg2rad(1), 0.001, 0.001]
        self.error_tol = [np.deg2rad(10), np.deg2rad(10), 0.1, 0.1]
        self.sample_space = [(0, 1.1 * np.pi), (-np.pi / 2, np.pi / 2), (-np.pi / 2, np.pi), (-np.pi, np.pi)]
        self.goal_bias = [0.5, 0.5, 0.5, 0.5]

    def params(self):
        m, L, g = self.m, self.L, self.g
        a = (m[0] + m[1]) * L[0]**2 + m[1] * L[1]**2        # demo_pendulum.py:62 constant part
        b2 = 2 * m[1] * L[0] * L[1]
        return np.concatenate(([a, b2, m[1] * L[1]**2, m[1] * L[0] * L[1],
                                g * (m[0] + m[1]) * L[0], m[1] * g * L[1]],
                               self.d, self.b, self.c, [self.umax, self.umax_plan], self.K.ravel()))


class PendulumLqr(DoublePendulum):
    """
    The double pendulum with the lqr of the reference's API contract (planner.py:39-42, tree.py:44-47): for every
    (x, u) the dynamics are linearised by central differences (step `eps`), S solves the discrete algebraic Riccati
    equation for the weights Q, R and K = (R + B'SB)^-1 B'SA -- what demo_pendulum.py imports
    scipy.linalg.solve_discrete_are for (:19) and never calls.  On the device this is the north-star steer pipeline:
    one problem per wavefront, finite-difference linearise -> doubling DARE -> K-gain forward rollout, K refreshed at
    every recorded step (planner.py:436); the nearest-neighbour cost uses S about the SAMPLE (:344-345), one matrix per
    sample.  params: as DoublePendulum, then 18 Q[4][4] | 34 R | 35 eps.
    """
    model = nat.MODEL_PENDULUM_LQR
    riccati = True

    def __init__(self, obstacle_seed=0, Q=(10.0, 10.0, 1.0, 1.0), R=0.1, eps=1e-6):
        DoublePendulum.__init__(self, obstacle_seed)
        self.Q = np.diag(np.asarray(Q, dtype=np.float64)) if np.ndim(Q) == 1 else np.array(Q, dtype=np.float64)
        self.R = np.array([[float(R)]])
        self.eps = float(eps)

    def params(self):
        return np.concatenate((DoublePendulum.params(self), self.Q.ravel(), self.R.ravel(), [self.eps]))

    def Smatrix(self):
        return None                                   # no constant S: it is a function of the state (lqr handle)

    def _eval_lqr(self, x, u):
        eng = self._engine(self.plan_kwargs["dt"] if self._ops_dt is None else None)
        S, K, _, _, _ = eng.lqr_dare_batch(np.atleast_2d(x), np.atleast_2d(np.asarray(u, dtype=np.float64).reshape(1, -1)),
                                           self.Q, self.R, self.eps)
        return (S[0], K[0])


class BoatNoviceLqr(BoatNovice):
    """
    demo_boat_novice.py's boat (6 states, 3 controls: the metric's dimension) with the lqr of the API contract instead of the
    demo's PD gain: A, B by central differences of the dynamics about (x, 0) -- like every lqr the reference ships, the
    callback ignores its `u` (the thruster clamp has zero slope beyond saturation: no stabilising Riccati solution there) --
    S from the doubling DARE, K = (R + B'SB)^-1 B'SA.  Per recorded rollout step, per new node, S per sample in the
    nearest-neighbour cost.  params: BoatNovice's 0..18, then 19 Q[6][6] | 55 R[3][3] | 64 eps.
    """
    model = nat.MODEL_BOAT_NOVICE_LQR
    riccati = True

    def __init__(self, obstacle_seed=0, obstacles=None, Q=(1.0, 1.0, 10.0, 0.1, 0.1, 0.1), R=(1e-5, 1e-5, 1e-6), eps=1e-6):
        BoatNovice.__init__(self, obstacle_seed, obstacles)
        self.Q = np.diag(np.asarray(Q, dtype=np.float64))
        self.R = np.diag(np.asarray(R, dtype=np.float64))
        self.eps = float(eps)
        # the demo's error_tol (goal_buffer / 2 = 3 m) makes every sample within 3 m of its nearest node converge on its first
        # step, so the tree stops growing after ~800 nodes: goal_buffer / 8 like the other boats
        self.error_tol = list(np.array(self.goal_buffer, dtype=np.float64) / 8)

    def params(self):
        return np.concatenate((BoatNovice.params(self), self.Q.ravel(), self.R.ravel(), [self.eps]))

    def Smatrix(self):
        return None                                   # no constant S: it is a function of the state (lqr handle)

    def _eval_lqr(self, x, u):
        eng = self._engine(self.plan_kwargs["dt"] if self._ops_dt is None else None)
        S, K, _, _, _ = eng.lqr_dare_batch(np.atleast_2d(x), np.zeros((1, self.ncontrols)), self.Q, self.R, self.eps)
        return (S[0], K[0])


# --------------------------------------------------------------------------- synthetic config 5

class DoubleIntegrator(NativeSystem):
    """
    BASELINE.json config 5 (not in the reference): q, qdot in R^6, u in R^6, explicit Euler
    q+ = q + dt qdot, qdot+ = qdot + dt u.  The cost-to-go S and gain K = (R+B'SB)^-1 B'SA solve the
    discrete Riccati equation for Q = R = I (build-added operator; golden = SciPy, see lqrrt_amd.dare).
    Obstacles: axis-aligned boxes [lo3, hi3] on q[0:3].
    params: 0 dt | 1 K[6][12]
    """
    model = nat.MODEL_DOUBLE_INTEGRATOR
    nstates, ncontrols = 12, 6
    wrap_dims = ()

    def __init__(self, n_boxes=1000, seed=0, dt=0.1, horizon=2.0, extent=100.0):
        NativeSystem.__init__(self)
        from .dare import dare_doubling
        dof = 6
        n, m = 12, 6
        self.plan_kwargs = dict(horizon=horizon, dt=dt, FPR=0.5)
        self.dt_model = dt
        self.A = np.eye(n)
        self.A[:dof, dof:] = dt * np.eye(dof)
        self.Bm = np.vstack((np.zeros((dof, dof)), dt * np.eye(dof)))
        self.S, self.K = dare_doubling(self.A, self.Bm, np.eye(n), np.eye(m))
        rs = np.random.RandomState(seed)
        centres = rs.uniform(0, extent, (n_boxes, 3))
        half = rs.uniform(0.1, 0.5, (n_boxes, 3))
        self.obs = np.hstack((centres - half, centres + half))
        self.obs_stride = 6
        self.x0 = np.zeros(n)
        self.goal = np.concatenate((np.full(3, 0.9 * extent), np.zeros(n - 3)))
        gb = np.full(n, np.inf)
        gb[:3] = 0.08 * extent
        self.goal_buffer = gb
        self.error_tol = gb / 8
        vmax = 2.0
        self.sample_space = [(0, extent)] * 3 + [(-1, 1)] * (dof - 3) + [(-vmax, vmax)] * dof
        self.goal_bias = [0.1] * 3 + [0] * (n - 3)

    def params(self):
        return np.concatenate(([self.dt_model], self.K.ravel()))


class UserSystem(NativeSystem):
    """
    Host-side description of an out-of-tree problem: the device code is lq::UserSystem of the header
    the library was built with (tools/build_user_system.py, loaded through LQRRT_LIB); this object carries the numbers --
    parameter block, dimensions, hull / obstacle tables, and the planning set-up the demos keep in module globals.
    """
    model = nat.MODEL_USER

    def __init__(self, nstates, ncontrols, params, wrap_dims=(), x0=None, goal=None, goal_buffer=None, error_tol=None,
                 sample_space=None, goal_bias=None, plan_kwargs=None, vps=None, obs=None, S=None):
        NativeSystem.__init__(self)
        self.nstates, self.ncontrols = int(nstates), int(ncontrols)
        self._params = np.ascontiguousarray(params, dtype=np.float64)
        self.wrap_dims = tuple(wrap_dims)
        self.x0 = np.zeros(self.nstates) if x0 is None else np.array(x0, dtype=np.float64)
        self.goal, self.goal_buffer, self.error_tol = goal, goal_buffer, error_tol
        self.sample_space, self.goal_bias = sample_space, goal_bias
        self.plan_kwargs = dict(plan_kwargs or dict(horizon=2, dt=0.1, FPR=0))
        if vps is not None:
            self.vps = np.ascontiguousarray(vps, dtype=np.float64).reshape(2, -1)
        if obs is not None:
            self.obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        self.S = None if S is None else np.array(S, dtype=np.float64)

    def params(self):
        return self._params


SYSTEMS = {
    "boat_advanced": BoatAdvanced,
    "boat_intermediate": BoatIntermediate,
    "boat_novice": BoatNovice,
    "car": Car,
    "pendulum": DoublePendulum,
    "double_integrator": DoubleIntegrator,
    "pendulum_lqr": PendulumLqr,
    "boat_novice_lqr": BoatNoviceLqr,
    "ros_boat": RosBoat,
}
