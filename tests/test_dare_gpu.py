"""
GPU: lqrrt_lqr_dare_batch (finite-difference linearisation + doubling DARE on the device) against
SciPy.  The reference has no Riccati solver, so the golden for this build-added
operator is scipy.linalg.solve_discrete_are on Jacobians obtained from the ORACLE's dynamics with the
same central-difference step.  Tolerances: Jacobians 2e-7 (difference quotients), S and K 1e-8
relative to scipy on the device's own Jacobians (1e-6 for the 1 ms double pendulum, |S| ~ 1e9).
"""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


def _fd(dyn, x, u, dt, eps):
    n, m = len(x), len(u)
    A, B = np.zeros((n, n)), np.zeros((n, m))
    for j in range(n):
        d = np.zeros(n); d[j] = eps
        A[:, j] = (dyn(x + d, np.copy(u), dt) - dyn(x - d, np.copy(u), dt)) / (2 * eps)
    for j in range(m):
        d = np.zeros(m); d[j] = eps
        B[:, j] = (dyn(np.copy(x), u + d, dt) - dyn(np.copy(x), u - d, dt)) / (2 * eps)
    return A, B


@pytest.mark.parametrize("name", ["boat_novice", "car", "double_integrator", "pendulum"])
def test_dare_batch_vs_scipy(name):
    import lqrrt_amd
    from systems_np import SYSTEMS
    s = lqrrt_amd.systems.SYSTEMS[name]() if name == "double_integrator" else lqrrt_amd.systems.SYSTEMS[name](0)
    rs = SYSTEMS[name]() if name == "double_integrator" else SYSTEMS[name](0)
    dt = s.plan_kwargs["dt"]
    eng = s._engine(dt)
    n, m = s.nstates, s.ncontrols
    rng = np.random.RandomState(0)
    Bn = 24
    if name == "pendulum":
        x = rng.uniform(-1, 1, (Bn, n))
        u = rng.uniform(-5, 5, (Bn, m))
    elif name == "double_integrator":
        x = rng.uniform(0, 50, (Bn, n))
        u = rng.uniform(-1, 1, (Bn, m))
    else:
        x = np.zeros((Bn, n))
        x[:, :2] = rng.uniform(0, 40, (Bn, 2))
        x[:, 2] = rng.uniform(-3, 3, Bn)
        x[:, 3] = rng.uniform(0.3, 1.0, Bn)                    # moving forward: heading stays controllable
        x[:, 4:] = rng.uniform(-0.1, 0.1, (Bn, n - 4))
        u = rng.uniform(-50, 50, (Bn, m))                       # inside the actuator limits (smooth region)
    Q, R = np.eye(n), np.eye(m) * (1e-4 if name in ("boat_novice", "car") else 1.0)
    eps = 1e-6
    S, K, A, B, it = eng.lqr_dare_batch(x, u, Q, R, eps=eps)
    assert it.max() <= 40
    for i in range(Bn):
        A_ref, B_ref = _fd(rs.dynamics, x[i], u[i], dt, eps)
        np.testing.assert_allclose(A[i], A_ref, rtol=0, atol=2e-7)
        np.testing.assert_allclose(B[i], B_ref, rtol=0, atol=2e-7)
        S_ref = scipy.linalg.solve_discrete_are(A[i], B[i], Q, R)
        K_ref = np.linalg.solve(R + B[i].T @ S_ref @ B[i], B[i].T @ S_ref @ A[i])
        scale = np.abs(S_ref).max()
        rtol = 1e-6 if name == "pendulum" else 1e-8          # dt = 1 ms makes the pendulum's DARE ill-conditioned (|S| ~ 1e9)
        assert np.abs(S[i] - S_ref).max() <= rtol * scale
        assert np.abs(K[i] - K_ref).max() <= rtol * max(1.0, np.abs(K_ref).max())
        np.testing.assert_allclose(S[i], S[i].T, rtol=0, atol=1e-12 * scale)
        # Riccati residual of the device solution
        res = A[i].T @ S[i] @ A[i] - S[i] - (A[i].T @ S[i] @ B[i]) @ K[i] + Q
        assert np.abs(res).max() <= (1e-5 if name == "pendulum" else 1e-7) * scale
    if name == "double_integrator":
        # host doubling on the EXACT A, B (lqrrt_amd/dare.py) agrees up to the difference-quotient noise
        # of the device Jacobians (|x| ~ 50, eps = 1e-6 -> ~5e-9 per entry)
        np.testing.assert_allclose(S[0], s.S, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(K[0], s.K, rtol=1e-6, atol=1e-6)
