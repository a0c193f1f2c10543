"""
Discrete algebraic Riccati equation for the LQR contract of the reference
(planner.py:39-42, tree.py:44-47: "S solves the local Riccati equation, K the feedback gain").

No shipped demo of the reference actually solves a Riccati equation (their `lqr` returns a
constant S and an analytic PD gain), so this operator is an addition of the build; its golden
is scipy.linalg.solve_discrete_are (tests/test_dare_gpu.py::test_dare_batch_vs_scipy on the device,
tests/test_pendulum_lqr.py for the systems whose lqr is a Riccati solution).

`dare_doubling` is the structure-preserving doubling algorithm (Chu, Fan, Lin, Wang 2004):
quadratically convergent (~9 iterations to 1e-14 for the double integrator, versus >100 plain
Riccati sweeps) and built only from n x n products and one n x n solve per iteration, which
is what the device kernel mirrors.  This host version is used once per system for constant
(linear time-invariant) problems, at set-up time, exactly like the demos' pinv(B).
"""
import numpy as np
import numpy.linalg as npl


def dare_doubling(A, B, Q, R, tol=1e-14, max_iter=64):
    """Returns (S, K): S = A'S(I+GS)^-1 A + Q with G = B R^-1 B', K = (R + B'SB)^-1 B'SA."""
    A = np.array(A, dtype=np.float64)
    B = np.array(B, dtype=np.float64)
    n = A.shape[0]
    Ak = A.copy()
    Gk = B.dot(npl.solve(np.array(R, dtype=np.float64), B.T))
    Hk = np.array(Q, dtype=np.float64)
    eye = np.eye(n)
    for _ in range(max_iter):
        W = eye + Gk.dot(Hk)
        WA = npl.solve(W, Ak)                 # (I + G H)^-1 A
        WG = npl.solve(W, Gk)                 # (I + G H)^-1 G
        H_next = Hk + Ak.T.dot(Hk).dot(WA)
        G_next = Gk + Ak.dot(WG).dot(Ak.T)
        A_next = Ak.dot(WA)
        done = npl.norm(H_next - Hk, 1) <= tol * max(1.0, npl.norm(H_next, 1))
        Ak, Gk, Hk = A_next, G_next, H_next
        if done:
            break
    S = 0.5 * (Hk + Hk.T)
    K = npl.solve(np.array(R, dtype=np.float64) + B.T.dot(S).dot(B), B.T.dot(S).dot(A))
    return S, K


def linearize(dynamics, x, u, dt, eps=1e-6):
    """Central finite-difference Jacobians of xnext = dynamics(x,u,dt) about (x,u): A (n x n), B (n x m)."""
    x = np.array(x, dtype=np.float64)
    u = np.array(u, dtype=np.float64)
    n, m = len(x), len(u)
    A = np.zeros((n, n))
    Bm = np.zeros((n, m))
    for j in range(n):
        d = np.zeros(n)
        d[j] = eps
        A[:, j] = (dynamics(x + d, np.copy(u), dt) - dynamics(x - d, np.copy(u), dt)) / (2 * eps)
    for j in range(m):
        d = np.zeros(m)
        d[j] = eps
        Bm[:, j] = (dynamics(np.copy(x), u + d, dt) - dynamics(np.copy(x), u - d, dt)) / (2 * eps)
    return A, Bm
