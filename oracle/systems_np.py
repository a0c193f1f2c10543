"""
ORACLE (test infrastructure, NOT product code) -- NumPy restatements of the problem plugins
(dynamics / lqr / erf / is_feasible + tables) that the reference ships only inside its demo
scripts.  Each class documents the demo lines it follows; arithmetic keeps the demos'
operation order so that, on the machine that generated tests/golden/, results are bit-equal.

  BoatAdvanced      <- demos/demo_boat_advanced.py:21-238
  BoatIntermediate  <- demos/demo_boat_intermediate.py:24-221
  BoatNovice        <- demos/demo_boat_novice.py:21-175
  Car               <- demos/demo_car.py:26-193
  DoublePendulum    <- demos/demo_pendulum.py:23-165
  DoubleIntegrator  <- NOT in the reference (BASELINE.json config 5); defined by this build,
                       S,K from scipy.linalg.solve_discrete_are.

Every system exposes: nstates, ncontrols, dynamics(x,u,dt), lqr(x,u)->(S,K), erf(xg,x),
is_feasible(x,u), batch_erf(xg,X), x0, goal, goal_buffer, error_tol, sample_space, goal_bias,
and `plan_kwargs` (horizon, dt, FPR from the demo's PLAN section).
"""
import numpy as np
import numpy.linalg as npl


def wrap_err(target_angle, angle):
    """Angle error on the circle, in the demos' exact form (e.g. demo_boat_advanced.py:159-164)."""
    c, s = np.cos(angle), np.sin(angle)
    cg, sg = np.cos(target_angle), np.sin(target_angle)
    return np.arctan2(sg * c - cg * s, cg * c + sg * s)


def noisy_obstacle_grid(seed, goal, x0, clearance, spacing=12, lo=5, hi=60):
    """
    The 'grid' obstacle field shared by the demos (demo_boat_advanced.py:190-202): a 6x6 lattice
    jittered by +-1.5 m, rounded to cm, radius 1; lattice points within `clearance` of start or goal
    stay as the never-hit placeholder [-9999,-9999,-9999].  np.random.seed(seed) reproduces the
    draw order of the demo when nothing else consumed the global stream before it.
    """
    rs = np.random.RandomState(seed)
    gx, gy = np.mgrid[slice(lo, hi + spacing, spacing), slice(lo, hi + spacing, spacing)]
    gx, gy = gx.reshape(gx.size), gy.reshape(gy.size)
    obs = np.full((gx.size, 3), -9999.0)
    goal = np.asarray(goal, dtype=np.float64)
    x0 = np.asarray(x0, dtype=np.float64)
    for i in range(gx.size):
        p = np.round([gx[i], gy[i]] + 3 * (rs.rand(2) - 0.5), 2)
        if npl.norm(p - goal[:2]) > clearance and npl.norm(np.array(p - x0[:2])) > clearance:
            obs[i] = [p[0], p[1], 1.0]
    return obs


def hull_grid(length, width, buffer, spacing):
    """Body-frame vertex lattice, 2xV (demo_boat_advanced.py:60-68; mgrid stop = +spacing)."""
    gx, gy = np.mgrid[slice(-(length + buffer) / 2, (length + buffer) / 2 + spacing, spacing),
                      slice(-(width + buffer) / 2, (width + buffer) / 2 + spacing, spacing)]
    return np.vstack((gx.reshape(gx.size), gy.reshape(gy.size))).astype(np.float64)


def rot3(h):
    c, s = np.cos(h), np.sin(h)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rot2(h):
    c, s = np.cos(h), np.sin(h)
    return np.array([[c, -s], [s, c]])


class _HeadingSystem(object):
    """Shared pieces of the planar vehicles: heading is state 2."""
    wrap_dims = (2,)

    def erf(self, xgoal, x):
        e = xgoal - x
        e[2] = wrap_err(xgoal[2], x[2])
        return e

    def batch_erf(self, xgoal, X):
        E = xgoal - X
        E[:, 2] = wrap_err(xgoal[2], X[:, 2])
        return E

    ogrid = None      # optional dict(grid, origin, cpm, threshold): the ROS node's collision model

    def set_occupancy_grid(self, grid, origin, cpm, threshold, vps=None):
        self.ogrid = dict(grid=np.asarray(grid), origin=np.asarray(origin, dtype=np.float64), cpm=cpm, threshold=threshold)
        if vps is not None:
            self.vps = np.asarray(vps, dtype=np.float64)

    def _grid_feasible(self, x):
        """demos/lqrrt_ros/nodes/lqrrt_node.py:730-745."""
        og = self.ogrid
        points = x[:2] + rot2(x[2]).dot(self.vps).T
        idx = (og["cpm"] * (points - og["origin"])).astype(np.int64)
        try:
            values = og["grid"][idx[:, 1], idx[:, 0]]
        except IndexError:
            return False
        return bool(np.all(values < og["threshold"]))

    def _hull_hits(self, verts):
        for ob in self.obs:
            if np.any(npl.norm(verts - ob[:2], axis=1) <= ob[2]):
                return True
        return False


# --------------------------------------------------------------------------- boats

class BoatAdvanced(_HeadingSystem):
    nstates, ncontrols = 6, 3
    plan_kwargs = dict(horizon=2, dt=0.1, FPR=0.9)          # demo_boat_advanced.py:245-249

    def __init__(self, obstacle_seed=0):
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = np.array([2.5, 1, 0.7])
        self.velmax_neg = np.array([-0.8, -1, -0.7])
        self.thrust_max = np.array([220, 220, 220, 220])
        pos = np.array([[-1.9, 1.0, -0.0123], [-1.9, -1.0, -0.0123],
                        [1.6, 0.6, -0.0123], [1.6, -0.6, -0.0123]])
        dirs = np.array([[0.7071, 0.7071, 0.0], [0.7071, -0.7071, 0.0],
                         [0.7071, -0.7071, 0.0], [0.7071, 0.7071, 0.0]])
        levers = np.cross(pos, dirs)
        self.B = np.concatenate((dirs.T, levers.T))[[0, 1, 5]]   # :44  3x4 wrench map
        self.invB = npl.pinv(self.B)                             # :45
        Fx = self.B.dot(self.thrust_max * [1, 1, 1, 1])[0]
        Fy = self.B.dot(self.thrust_max * [1, -1, -1, 1])[1]
        Mz = self.B.dot(self.thrust_max * [-1, 1, -1, 1])[2]
        self.D_pos = np.abs([Fx, Fy, Mz] / self.velmax_pos)
        self.D_neg = np.abs([Fx, Fy, Mz] / self.velmax_neg)
        self.boat_length = 210 * 0.0254
        self.boat_width = 96 * 0.0254
        self.vps = hull_grid(self.boat_length, self.boat_width, 0.25, 1)
        self.magic_rudder = 4000
        self.kp = np.diag([120, 20, 0])
        self.kd = np.diag([120, 20, 0])
        self.x0 = np.array([0, 0, np.deg2rad(0), 0, 0, 0])
        self.goal = [40, 40, np.deg2rad(90), 0, 0, 0]
        self.goal_buffer = [8, 8, np.inf, np.inf, np.inf, np.inf]
        self.error_tol = np.copy(self.goal_buffer) / 8
        self.obs = noisy_obstacle_grid(obstacle_seed, self.goal, self.x0, 2 * self.boat_length)
        self.velmax_pos_plan = np.array([1.1, 0.4, 0.2])
        self.velmax_neg_plan = np.array([-0.65, -0.4, -0.2])
        self.sample_space = [(self.x0[0], self.goal[0]), (self.x0[1], self.goal[1]), (0, 0),
                             (0.9 * self.velmax_pos_plan[0], self.velmax_pos_plan[0]),
                             (-abs(self.velmax_neg_plan[1]), self.velmax_pos_plan[1]),
                             (-abs(self.velmax_neg_plan[2]), self.velmax_pos_plan[2])]
        self.goal_bias = [0.2, 0.2, 0, 0, 0, 0]

    def dynamics(self, x, u, dt):
        """demo_boat_advanced.py:78-130 (planning=True branch)."""
        R = rot3(x[2])
        D = np.where(x[3:] >= 0, self.D_pos, self.D_neg)
        vw = R[:2, :2].dot(x[3:5])
        u[2] = u[2] + self.magic_rudder * wrap_err(np.arctan2(vw[1], vw[0]), x[2])
        u = self.B.dot(np.clip(self.invB.dot(u), -self.thrust_max, self.thrust_max))
        xdot = np.concatenate((R.dot(x[3:]), self.invM * (u - D * x[3:])))
        xn = x + xdot * dt
        if x[3] > 0:
            xn[5] = np.clip(np.abs(xn[3] / self.velmax_pos[0]), 0, 1) * xn[5]
        elif x[3] < 0:
            xn[5] = np.clip(np.abs(xn[3] / self.velmax_neg[0]), 0, 1) * xn[5]
        if xn[3] < 0:
            xn[3] = 0
        return xn

    def lqr(self, x, u):
        """demo_boat_advanced.py:139-151."""
        return (np.diag([1, 1, 1, 1, 1, 1]), np.hstack((self.kp.dot(rot3(x[2]).T), self.kd)))

    def is_feasible(self, x, u):
        """demo_boat_advanced.py:209-225: planning speed box, then hull-vs-circles."""
        v = x[3:]
        if np.any(v > self.velmax_pos_plan) or np.any(v < self.velmax_neg_plan):
            return False
        if self.ogrid is not None:
            return self._grid_feasible(x)
        verts = x[:2] + rot2(x[2]).dot(self.vps).T
        return not self._hull_hits(verts)


class BoatIntermediate(_HeadingSystem):
    nstates, ncontrols = 6, 3
    plan_kwargs = dict(horizon=2, dt=0.1, FPR=0.5)          # demo_boat_intermediate.py:228-232

    def __init__(self, obstacle_seed=0):
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = [1.1, 0.45, 0.2]
        self.velmax_neg = [0.68, 0.45, 0.2]
        thrust_max, lever = 220, 2.15
        self.u_max = np.array([2 * np.sqrt(2) * thrust_max, 0.2 * np.sqrt(2) * thrust_max,
                               4 * lever * thrust_max])
        self.D_pos = np.abs(self.u_max / self.velmax_pos)
        self.D_neg = np.abs(self.u_max / self.velmax_neg)
        self.boat_length = 210 * 0.0254
        self.boat_width = 96 * 0.0254
        self.vps = hull_grid(self.boat_length, self.boat_width, 2, 0.5)
        self.rudder = 5000
        self.kp = np.diag([120, 120, 0])
        self.kd = np.diag([120, 120, 0])
        self.x0 = np.array([0, 0, np.deg2rad(0), 0, 0, 0])
        self.goal = [40, 40, np.deg2rad(90), 0, 0, 0]
        self.goal_buffer = [8, 8, np.inf, np.inf, np.inf, np.inf]
        self.error_tol = np.copy(self.goal_buffer) / 8
        self.obs = noisy_obstacle_grid(obstacle_seed, self.goal, self.x0, 2 * self.boat_length)
        self.sample_space = [(self.x0[0], self.goal[0]), (self.x0[1], self.goal[1]), (0, 0),
                             (0.9 * self.velmax_pos[0], self.velmax_pos[0]),
                             (-self.velmax_neg[1], self.velmax_pos[1]),
                             (-self.velmax_neg[2], self.velmax_pos[2])]
        self.goal_bias = [0.2, 0.2, 0, 0, 0, 0]

    def _saturate(self, u):
        for i, mag in enumerate(np.abs(u)):
            if mag > self.u_max[i]:
                u[i] = self.u_max[i] * np.sign(u[i])
        return u

    def dynamics(self, x, u, dt):
        """demo_boat_intermediate.py:48-100."""
        R = rot3(x[2])
        D = np.where(x[3:] >= 0, self.D_pos, self.D_neg)
        vw = R[:2, :2].dot(x[3:5])
        u[2] = u[2] + self.rudder * wrap_err(np.arctan2(vw[1], vw[0]), x[2])
        u = self._saturate(u)
        xdot = np.concatenate((R.dot(x[3:]), self.invM * (u - D * x[3:])))
        xn = x + xdot * dt
        if x[3] > 0:
            xn[5] = np.clip(np.abs(xn[3] / self.velmax_pos[0]), 0, 1) * xn[5]
        elif x[3] < 0:
            xn[5] = np.clip(np.abs(xn[3] / self.velmax_neg[0]), 0, 1) * xn[5]
        if xn[3] < 0:
            xn[3] = 0
        return xn

    def lqr(self, x, u):
        return (np.diag([1, 1, 1, 1, 1, 1]), np.hstack((self.kp.dot(rot3(x[2]).T), self.kd)))

    def is_feasible(self, x, u):
        """demo_boat_intermediate.py:198-210."""
        verts = x[:2] + rot2(x[2]).dot(self.vps).T
        return not self._hull_hits(verts)


class BoatNovice(_HeadingSystem):
    nstates, ncontrols = 6, 3
    plan_kwargs = dict(horizon=2, dt=0.1, FPR=0.5)          # demo_boat_novice.py:182-186

    def __init__(self, obstacle_seed=0):
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = [1.1, 0.45, 0.2]
        self.velmax_neg = [0.68, 0.45, 0.2]
        thrust_max, lever = 220, 2.15
        self.u_max = np.array([2 * np.sqrt(2) * thrust_max, 2 * np.sqrt(2) * thrust_max,
                               4 * lever * thrust_max])
        self.D_pos = np.abs(self.u_max / self.velmax_pos)
        self.D_neg = np.abs(self.u_max / self.velmax_neg)
        self.boat_length = 210 * 0.0254
        self.boat_width = 96 * 0.0254
        self.kp = np.diag([120, 120, 350])
        self.kd = np.diag([120, 120, 100])
        self.x0 = np.array([0, 0, np.deg2rad(0), 0, 0, 0])
        self.goal = [40, 40, np.deg2rad(90), 0, 0, 0]
        self.goal_buffer = [6, 6, np.inf, np.inf, np.inf, np.inf]
        self.error_tol = np.copy(self.goal_buffer) / 2
        self.obs = noisy_obstacle_grid(obstacle_seed, self.goal, self.x0, 2 * self.boat_length)
        self.sample_space = [(self.x0[0], self.goal[0]), (self.x0[1], self.goal[1]), (-np.pi, np.pi),
                             (0.5 * self.velmax_pos[0], self.velmax_pos[0]),
                             (-self.velmax_neg[1], self.velmax_pos[1]),
                             (-self.velmax_neg[2], self.velmax_pos[2])]
        self.goal_bias = [0.5, 0.5, 0, 0, 0, 0]

    def dynamics(self, x, u, dt):
        """demo_boat_novice.py:45-75."""
        R = rot3(x[2])
        D = np.where(x[3:] >= 0, self.D_pos, self.D_neg)
        for i, mag in enumerate(np.abs(u)):
            if mag > self.u_max[i]:
                u[i] = self.u_max[i] * np.sign(u[i])
        xdot = np.concatenate((R.dot(x[3:]), self.invM * (u - D * x[3:])))
        return x + xdot * dt

    def lqr(self, x, u):
        return (np.diag([1, 1, 1, 1, 1, 1]), np.hstack((self.kp.dot(rot3(x[2]).T), self.kd)))

    def is_feasible(self, x, u):
        """demo_boat_novice.py:160-164: centre point vs circles inflated by half the boat length."""
        for ob in self.obs:
            if npl.norm(x[:2] - ob[:2]) <= self.boat_length / 2 + ob[2]:
                return False
        return True


class RosBoat(_HeadingSystem):
    """demos/lqrrt_ros/behaviors/{params,boat,car,escape}.py: one boat, three behaviours."""
    nstates, ncontrols = 6, 3
    plan_kwargs = dict(horizon=(0.1, 3), dt=0.1, FPR=0)     # params.py:31-33

    def __init__(self, behavior="boat", focus=None):
        self.behavior = behavior
        self.focus = None if focus is None else np.array(focus, dtype=np.float64)
        m, I = 350, 400
        self.invM = np.array([1 / m, 1 / m, 1 / I])
        self.velmax_pos = np.array([1.2, 0.6, 0.22])
        self.velmax_neg = np.array([-0.6, -0.6, -0.22])
        self.thrust_max = np.array([220, 220, 220, 220])
        pos = np.array([[-1.9, 1.0, -0.0123], [-1.9, -1.0, -0.0123], [1.6, 0.6, -0.0123], [1.6, -0.6, -0.0123]])
        dirs = np.array([[0.7071, 0.7071, 0.0], [0.7071, -0.7071, 0.0], [0.7071, -0.7071, 0.0], [0.7071, 0.7071, 0.0]])
        levers = np.cross(pos, dirs)
        self.B = np.concatenate((dirs.T, levers.T))[[0, 1, 5]]
        self.invB = npl.pinv(self.B)
        Fx = self.B.dot(self.thrust_max * [1, 1, 1, 1])[0]
        Fy = self.B.dot(self.thrust_max * [1, -1, -1, 1])[1]
        Mz = self.B.dot(self.thrust_max * [-1, 1, -1, 1])[2]
        self.D_pos = np.abs([Fx, Fy, Mz] / self.velmax_pos)
        self.D_neg = np.abs([Fx, Fy, Mz] / self.velmax_neg)
        self.vps = hull_grid(210 * 0.0254, 96 * 0.0254, 0.15, 0.1)
        self.obs = np.zeros((0, 3))
        real_tol = [0.5, 0.5, np.deg2rad(10), np.inf, np.inf, np.inf]
        free_radius = 6
        self.Smat = np.diag([1, 1, 1, 1, 1, 1])
        if behavior == "boat":
            self.rudder = 8000
            self.kp, self.kd = np.diag([250, 250, 2500]), np.diag([5, 5, 0.001])
            self.goal_buffer = [real_tol[0], real_tol[1], real_tol[2], 10, 10, 6]
            self.error_tol = np.copy(self.goal_buffer)
        elif behavior == "car":
            self.rudder = 6000
            self.kp, self.kd = np.diag([150, 150, 0]), np.diag([150, 5, 0])
            self.Smat = np.diag([1, 1, 1, 0, 0, 0])
            self.goal_buffer = [0.5 * free_radius, 0.5 * free_radius, np.inf, np.inf, np.inf, np.inf]
            self.error_tol = np.copy(self.goal_buffer) / 10
        else:
            self.kp, self.kd = np.diag([150, 150, 2000]), np.diag([120, 120, 0.01])
            self.goal_buffer = [free_radius, free_radius, np.inf, np.inf, np.inf, np.inf]
            self.error_tol = np.copy(self.goal_buffer)
        self.x0 = np.zeros(6)
        self.goal = [30, 20, np.deg2rad(45), 0, 0, 0]
        self.goal_bias = [0.3, 0.3, 0, 0, 0, 0]
        self.sample_space = self.gen_ss(self.x0, self.goal)

    def gen_ss(self, seed, goal, buff=None):
        vp, vn = self.velmax_pos, self.velmax_neg
        if self.behavior == "escape":                      # escape.py:67-77
            buff = 40 if buff is None else buff
            return [(seed[0] - buff, seed[0] + buff), (seed[1] - buff, seed[1] + buff), (seed[2], seed[2]),
                    (-abs(vn[0]), vp[0]), (-abs(vn[1]), vp[1]), (-abs(vn[2]), vp[2])]
        buff = [10] * 4 if buff is None else buff          # boat.py:86-96 / car.py:84-94
        vx = (0.9 * vp[0], vp[0]) if self.behavior == "car" else (-abs(vn[0]), vp[0])
        return [(min([seed[0], goal[0]]) - buff[0], max([seed[0], goal[0]]) + buff[1]),
                (min([seed[1], goal[1]]) - buff[2], max([seed[1], goal[1]]) + buff[3]),
                (-np.pi, np.pi), vx, (-abs(vn[1]), vp[1]), (-abs(vn[2]), vp[2])]

    def dynamics(self, x, u, dt):
        R = rot3(x[2])
        D = np.where(x[3:] >= 0, self.D_pos, self.D_neg)
        if self.behavior == "boat" and self.focus is not None:          # boat.py:34-42
            vec = self.focus[:2] - x[:2]
            u[2] = self.rudder * wrap_err(np.arctan2(vec[1], vec[0]), x[2])
        elif self.behavior == "car":                                    # car.py:36-43
            vw = R[:2, :2].dot(x[3:5])
            u[2] = self.rudder * wrap_err(np.arctan2(vw[1], vw[0]), x[2])
        if self.behavior == "car":                                      # car.py:46
            u = self.B.dot(np.clip(self.invB.dot(u), -self.thrust_max, self.thrust_max))
        else:                                                           # boat.py:44-48 / escape.py:32-36
            thrusts = self.invB.dot(u)
            ratios = self.thrust_max / np.clip(np.abs(thrusts), 1E-6, np.inf)
            if np.any(ratios < 1):
                u = self.B.dot(np.min(ratios) * thrusts)
        xdot = np.concatenate((R.dot(x[3:]), self.invM * (u - D * x[3:])))
        xn = x + xdot * dt
        if self.behavior == "car" and xn[3] < 0:                        # car.py:54-56
            xn[3] = abs(x[3])
        return xn

    def lqr(self, x, u):
        return (self.Smat, np.hstack((self.kp.dot(rot3(x[2]).T), self.kd)))

    def is_feasible(self, x, u):
        if self.ogrid is None:                                          # lqrrt_node.py:726-727
            return True
        return self._grid_feasible(x)


# --------------------------------------------------------------------------- car

class Car(_HeadingSystem):
    nstates, ncontrols = 5, 2
    plan_kwargs = dict(horizon=5, dt=0.1, FPR=0)            # demo_car.py:200-204 (FPR defaulted)

    def __init__(self, obstacle_seed=0):
        m, I = 500, 500
        self.invM = np.array([1 / m, 1 / I])
        self.velmax = [1.1, 1]
        self.u_max = np.array([650, 1800])
        self.D = np.abs(self.u_max / self.velmax)
        self.vps = hull_grid(6, 3, 2, 0.5)
        self.kp = np.diag([120, 600])
        self.kd = np.diag([120, 600])
        self.x0 = np.array([0, 0, np.deg2rad(0), 0, 0])
        self.goal = [40, 40, np.deg2rad(90), 0, 0]
        self.goal_buffer = [8, 8, np.inf, np.inf, np.inf]
        self.error_tol = np.copy(self.goal_buffer) / 2
        self.obs = np.array([[20, 20, 5], [10, 30, 2], [40, 10, 3]], dtype=np.float64)  # 'some', :145-149
        buff = 40
        self.sample_space = [(self.goal[0] - buff, self.goal[0] + buff),
                             (self.goal[0] - buff, self.goal[1] + buff),
                             (-np.pi, np.pi), (0.9 * self.velmax[0], self.velmax[0]),
                             (-self.velmax[1], self.velmax[1])]
        self.goal_bias = [0.5, 0.5, 0, 0, 0]

    def dynamics(self, x, u, dt):
        """demo_car.py:46-72."""
        vwx = np.cos(x[2]) * x[3]
        vwy = np.sin(x[2]) * x[3]
        u = np.clip(u, [-self.u_max[0] / 10, -self.u_max[1]], self.u_max)
        xdot = np.concatenate(([vwx, vwy, x[4]], self.invM * (u - self.D * x[3:])))
        xn = x + xdot * dt
        if xn[3] < 0:
            xn[3] = 0
        xn[4] = np.clip(np.abs(xn[3] / self.velmax[0]), 0, 1) * xn[4]
        return xn

    def lqr(self, x, u):
        """demo_car.py:98-113: rows 0 and 2 of R(h)'."""
        w2b = np.array([[np.cos(x[2]), np.sin(x[2]), 0], [0, 0, 1]])
        return (np.diag([1, 1, 1, 1, 1]), np.hstack((self.kp.dot(w2b), self.kd)))

    def is_feasible(self, x, u):
        """demo_car.py:168-180 -- note the extra vertex: the vstack appends x[:2], so it lands at 2*p."""
        verts = x[:2] + np.vstack((rot2(x[2]).dot(self.vps).T, x[:2]))
        return not self._hull_hits(verts)


# --------------------------------------------------------------------------- double pendulum

class DoublePendulum(object):
    nstates, ncontrols = 4, 1
    wrap_dims = (0, 1)
    # demo_pendulum.py:172-176 passes horizon=0 (< dt) which the reference rejects
    # (planner.py:548-553); this build runs the problem with horizon=0.05 -> 50 steps.
    plan_kwargs = dict(horizon=0.05, dt=0.001, FPR=0.5)

    def __init__(self, obstacle_seed=0):
        self.L = [1, 0.5]
        self.m = [5, 5]
        self.g = 9.81
        self.d = [0.4, 0.4]
        self.b = [0.01, 0.01]
        self.c = [0.1, 0.1]
        self.umax = np.inf
        self.x0 = np.array([-np.pi / 2, 0, 0, 0])
        self.goal = [np.pi / 2, 0, 0, 0]
        self.goal_buffer = [np.deg2rad(1), np.deg2rad(1), 0.001, 0.001]
        self.error_tol = [np.deg2rad(10), np.deg2rad(10), 0.1, 0.1]
        self.umax_plan = 0.75 * self.umax
        self.sample_space = [(0, 1.1 * np.pi), (-np.pi / 2, np.pi / 2), (-np.pi / 2, np.pi), (-np.pi, np.pi)]
        self.goal_bias = [0.5, 0.5, 0.5, 0.5]
        self.obs = np.zeros((0, 3))

    def dynamics(self, q, u, dt):
        """demo_pendulum.py:54-100: manipulator equation, explicit Euler."""
        m, L, g, d, b, c = self.m, self.L, self.g, self.d, self.b, self.c
        M = np.zeros((2, 2))
        M[0, 0] = (m[0] + m[1]) * L[0]**2 + m[1] * L[1]**2 + 2 * m[1] * L[0] * L[1] * np.cos(q[1])
        M[0, 1] = m[1] * L[1]**2 + m[1] * L[0] * L[1] * np.cos(q[1])
        M[1, 0] = M[0, 1]
        M[1, 1] = m[1] * L[1]**2
        V = np.array([-m[1] * L[0] * L[1] * (2 * q[2] * q[3] + q[3]**2) * np.sin(q[1]),
                      m[1] * L[0] * L[1] * q[2]**2 * np.sin(q[1])])
        G = np.array([g * (m[0] + m[1]) * L[0] * np.cos(q[0]) + m[1] * g * L[1] * np.cos(q[0] + q[1]),
                      m[1] * g * L[1] * np.cos(q[0] + q[1])])
        Dj = np.array([d[0] * q[2], d[1] * q[3]])
        F = np.array([b[0] * np.tanh(c[0] * q[2]), b[1] * np.tanh(c[1] * q[3])])
        u = np.clip(u, -self.umax, self.umax)
        u = np.concatenate((u, [0]))
        return q + (np.concatenate((q[2:], npl.inv(M).dot(u - V - G - Dj - F))) * dt)

    def lqr(self, x, u):
        """demo_pendulum.py:119-126: constant gains."""
        return (np.diag([1, 1, 1, 1]), np.array([[10, 200, 0, 0]]))

    def erf(self, qgoal, q):
        """demo_pendulum.py:130-142: both joint angles wrap."""
        e = qgoal - q
        for i in (0, 1):
            e[i] = wrap_err(qgoal[i], q[i])
        return e

    def batch_erf(self, qgoal, Q):
        E = qgoal - Q
        for i in (0, 1):
            E[:, i] = wrap_err(qgoal[i], Q[:, i])
        return E

    def is_feasible(self, x, u):
        """demo_pendulum.py:154-157 (umax=inf -> never infeasible)."""
        if abs(u) > self.umax_plan:
            return False
        return True


class PendulumLqr(DoublePendulum):
    """
    The double pendulum with the lqr the reference's API contract describes (planner.py:39-42, tree.py:44-47: "S solves
    the local Riccati equation, K the feedback gain") -- what demo_pendulum.py imports scipy.linalg.solve_discrete_are
    for (:19) without calling it: A, B by central differences of `dynamics` about (x, u), S = solve_discrete_are(A, B, Q, R),
    K = (R + B'SB)^-1 B'SA.  Written the way a user of the reference would write the callback (NumPy + SciPy); the
    reference Planner driven by it generates tests/golden/traj_pendulum_lqr_*.npz.
    """

    def __init__(self, obstacle_seed=0, Q=(10.0, 10.0, 1.0, 1.0), R=0.1, eps=1e-6):
        DoublePendulum.__init__(self, obstacle_seed)
        self.Q = np.diag(np.asarray(Q, dtype=np.float64))
        self.R = np.array([[float(R)]])
        self.eps = float(eps)

    def linearize(self, x, u):
        n, m, dt, eps = self.nstates, self.ncontrols, self.plan_kwargs["dt"], self.eps
        x = np.array(x, dtype=np.float64)
        u = np.atleast_1d(np.array(u, dtype=np.float64))
        A, B = np.zeros((n, n)), np.zeros((n, m))
        for j in range(n):
            d = np.zeros(n)
            d[j] = eps
            A[:, j] = (self.dynamics(x + d, np.copy(u), dt) - self.dynamics(x - d, np.copy(u), dt)) / (2 * eps)
        for j in range(m):
            d = np.zeros(m)
            d[j] = eps
            B[:, j] = (self.dynamics(np.copy(x), u + d, dt) - self.dynamics(np.copy(x), u - d, dt)) / (2 * eps)
        return A, B

    def lqr(self, x, u):
        import scipy.linalg
        A, B = self.linearize(x, u)
        S = scipy.linalg.solve_discrete_are(A, B, self.Q, self.R)
        K = npl.solve(self.R + B.T.dot(S).dot(B), B.T.dot(S).dot(A))
        return (S, K)


class BoatNoviceLqr(BoatNovice):
    """
    demo_boat_novice.py's boat with a Riccati lqr callback written the way a user of the reference would (NumPy + SciPy):
    A, B by central differences of `dynamics` about (x, 0) (the callback ignores u, like every lqr the reference ships),
    S = solve_discrete_are(A, B, Q, R), K = (R + B'SB)^-1 B'SA.  The reference Planner driven by it generates
    tests/golden/traj_boat_novice_lqr_*.npz.
    """

    def __init__(self, obstacle_seed=0, Q=(1.0, 1.0, 10.0, 0.1, 0.1, 0.1), R=(1e-5, 1e-5, 1e-6), eps=1e-6):
        BoatNovice.__init__(self, obstacle_seed)
        self.Q = np.diag(np.asarray(Q, dtype=np.float64))
        self.R = np.diag(np.asarray(R, dtype=np.float64))
        self.eps = float(eps)
        # the demo's error_tol (goal_buffer / 2 = 3 m) makes every sample within 3 m of its nearest node converge on its first
        # step, so the tree stops growing after ~800 nodes: goal_buffer / 8 like the other boats
        self.error_tol = list(np.array(self.goal_buffer, dtype=np.float64) / 8)

    def linearize(self, x, u=None):
        n, m, dt, eps = self.nstates, self.ncontrols, self.plan_kwargs["dt"], self.eps
        x = np.array(x, dtype=np.float64)
        u = np.zeros(m)
        A, B = np.zeros((n, n)), np.zeros((n, m))
        for j in range(n):
            d = np.zeros(n)
            d[j] = eps
            A[:, j] = (self.dynamics(x + d, np.copy(u), dt) - self.dynamics(x - d, np.copy(u), dt)) / (2 * eps)
        for j in range(m):
            d = np.zeros(m)
            d[j] = eps
            B[:, j] = (self.dynamics(np.copy(x), u + d, dt) - self.dynamics(np.copy(x), u - d, dt)) / (2 * eps)
        return A, B

    def lqr(self, x, u):
        import scipy.linalg
        A, B = self.linearize(x)
        S = scipy.linalg.solve_discrete_are(A, B, self.Q, self.R)
        K = npl.solve(self.R + B.T.dot(S).dot(B), B.T.dot(S).dot(A))
        return (S, K)


# --------------------------------------------------------------------------- synthetic config 5

class DoubleIntegrator(object):
    """
    BASELINE.json config 5 (not in the reference): q in R^d, qdot in R^d, u in R^d, explicit
    Euler  q+ = q + qdot*dt, qdot+ = qdot + u*dt ; S,K from the discrete Riccati equation with
    Q=R=I (scipy.linalg.solve_discrete_are); axis-aligned boxes on q[0:3] are obstacles.
    """
    wrap_dims = ()

    def __init__(self, dof=6, n_boxes=1000, seed=0, dt=0.1, horizon=2.0, extent=100.0):
        import scipy.linalg
        self.dof = dof
        self.nstates, self.ncontrols = 2 * dof, dof
        self.plan_kwargs = dict(horizon=horizon, dt=dt, FPR=0.5)
        n, m = self.nstates, self.ncontrols
        self.A = np.eye(n)
        self.A[:dof, dof:] = dt * np.eye(dof)
        self.Bm = np.vstack((np.zeros((dof, dof)), dt * np.eye(dof)))
        Q, Rm = np.eye(n), np.eye(m)
        self.S = scipy.linalg.solve_discrete_are(self.A, self.Bm, Q, Rm)
        self.K = npl.solve(Rm + self.Bm.T.dot(self.S).dot(self.Bm), self.Bm.T.dot(self.S).dot(self.A))
        rs = np.random.RandomState(seed)
        centres = rs.uniform(0, extent, (n_boxes, 3))
        half = rs.uniform(0.1, 0.5, (n_boxes, 3))
        self.box_lo, self.box_hi = centres - half, centres + half
        self.x0 = np.zeros(n)
        self.goal = np.concatenate((np.full(3, 0.9 * extent), np.zeros(n - 3)))
        gb = np.full(n, np.inf)
        gb[:3] = 0.08 * extent
        self.goal_buffer = gb
        self.error_tol = gb / 8
        vmax = 2.0
        self.sample_space = [(0, extent)] * 3 + [(-1, 1)] * (dof - 3) + [(-vmax, vmax)] * dof
        self.goal_bias = [0.1] * 3 + [0] * (n - 3)
        keep = np.all(np.abs((centres - self.x0[:3])) > 1.0, axis=1) | True
        assert keep.all()

    def dynamics(self, x, u, dt):
        return self.A.dot(x) + self.Bm.dot(u)

    def lqr(self, x, u):
        return (self.S, self.K)

    def erf(self, xgoal, x):
        return xgoal - x

    def batch_erf(self, xgoal, X):
        return xgoal - X

    def is_feasible(self, x, u):
        p = x[:3]
        return not bool(np.any(np.all((p >= self.box_lo) & (p <= self.box_hi), axis=1)))


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


def make_oracle_planner(system, max_nodes, fake_clock=True, vectorised_nn=True, **overrides):
    """RefPlanner wired to a system object with the demo's PLAN kwargs (SURVEY.md 8c recipe)."""
    from lqrrt_oracle import RefConstraints, RefPlanner
    cons = RefConstraints(system.nstates, system.ncontrols, system.goal_buffer, system.is_feasible)
    kw = dict(system.plan_kwargs)
    kw.update(error_tol=system.error_tol, erf=system.erf, min_time=0, max_time=1,
              max_nodes=max_nodes, goal0=system.goal, printing=False)
    if fake_clock:
        kw["sys_time"] = lambda: 0.0
    kw.update(overrides)
    p = RefPlanner(system.dynamics, system.lqr, cons, **kw)
    if vectorised_nn:
        p.batch_erf = system.batch_erf
    return p
