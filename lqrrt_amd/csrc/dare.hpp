// LQR by finite-difference linearisation + discrete algebraic Riccati equation, on the device.
//
// The reference's API contract says lqr(x,u) returns "S solving the local Riccati equation and K the
// associated feedback gain" (planner.py:39-42, tree.py:44-47) but none of its demos computes one
// (constant S, analytic PD gain).  This operator supplies the general case for the compiled-in
// dynamics: A = df/dx, B = df/du by central differences of S::step about (x,u), then the
// structure-preserving doubling algorithm (Chu, Fan, Lin, Wang 2004)
//     A+ = A (I+GH)^-1 A,  G+ = G + A (I+GH)^-1 G A',  H+ = H + A' H (I+GH)^-1 A,   H -> S
// and K = (R + B'SB)^-1 B'SA.  Golden: scipy.linalg.solve_discrete_are (tests/test_dare_gpu.py).
//
// One problem per wavefront; all matrices (n <= 12) live in LDS; the 64 lanes split matrix
// elements.  ~9 doubling iterations reach 1e-14 where plain Riccati sweeps need >100.
#pragma once
#include "systems.hpp"

namespace lq {

// C[r x c] = A[r x k] * B[k x c]   (ta/tb: use the transpose of the stored operand)
__device__ __forceinline__ void mm(double* C, const double* A, const double* B, int r, int k, int c, bool ta, bool tb, int lane) {
    for (int idx = lane; idx < r * c; idx += 64) {
        const int i = idx / c, j = idx % c;
        double acc = 0.0;
        for (int p = 0; p < k; ++p) {
            const double a = ta ? A[p * r + i] : A[i * k + p];
            const double b = tb ? B[j * k + p] : B[p * c + j];
            acc += a * b;
        }
        C[idx] = acc;
    }
    __syncthreads();
}

// Solves W X = RHS in place (Gauss-Jordan, partial pivoting); W is n x n, RHS is n x q; both destroyed/overwritten.
__device__ __forceinline__ void solve_inplace(double* W, double* RHS, int n, int q, int lane) {
    for (int p = 0; p < n; ++p) {
        // pivot search over rows >= p (lanes over rows, then a butterfly)
        double best = -1.0;
        int brow = p;
        for (int r = p + lane; r < n; r += 64) {
            const double v = fabs(W[r * n + p]);
            if (v > best) { best = v; brow = r; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int orow = __shfl_xor(brow, off);
            if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
        }
        if (brow != p) {
            for (int j = lane; j < n + q; j += 64) {
                double* a = j < n ? &W[p * n + j] : &RHS[p * q + (j - n)];
                double* b = j < n ? &W[brow * n + j] : &RHS[brow * q + (j - n)];
                const double t = *a; *a = *b; *b = t;
            }
        }
        __syncthreads();
        const double piv = W[p * n + p];
        __syncthreads();
        for (int j = lane; j < n + q; j += 64) {
            if (j < n) W[p * n + j] = W[p * n + j] / piv; else RHS[p * q + (j - n)] = RHS[p * q + (j - n)] / piv;
        }
        __syncthreads();
        for (int idx = lane; idx < n * (n + q); idx += 64) {
            const int r = idx / (n + q), j = idx % (n + q);
            if (r == p) continue;
            const double f = W[r * n + p];
            if (j == p) continue;                       // column p is zeroed after the sweep
            if (j < n) W[r * n + j] -= f * W[p * n + j]; else RHS[r * q + (j - n)] -= f * RHS[p * q + (j - n)];
        }
        __syncthreads();
        for (int r = lane; r < n; r += 64) if (r != p) W[r * n + p] = 0.0;
        __syncthreads();
    }
}

// LDS work space of one problem (one wavefront)
template <int n, int m>
struct DareLds {
    double A[n * n], Bm[n * m], Ak[n * n], G[n * n], Hm[n * n], W[n * n], T1[n * n], T2[n * n], T3[n * n];
    double Rm[m * m], X[m * n], Y[m * n], Z[m * m];
    double red[2];
};

// lqr(x, u) of the API contract for one (x0, u0), computed by the calling wavefront (all 64 lanes must call; x0 / u0 are
// wave-uniform per-thread arrays): A, B by central differences of S::step, S by doubling, K = (R + B'SB)^-1 B'SA.
// P: model parameters; Qd (n x n), Rd (m x m): weights (any address space).  Results are left in the work space:
// L.T1 = S (symmetrised), L.Y = K (m x n), L.A / L.Bm = the linearisation.  Returns the doubling iterations used.
// Every sum runs in a fixed order inside ONE lane (mm, solve_inplace), so the result does not depend on the lane
// count and oracle/lqrrt_oracle.c restates it sequentially bit for bit.
template <class S>
__device__ __forceinline__ int dare_lqr(const double* P, const double* x0, const double* u0, const double* Qd, const double* Rd,
                                        double dt, double eps, int max_iter, double tol, DareLds<S::N, S::M>& L, int lane) {
    constexpr int n = S::N, m = S::M;
    double *A = L.A, *Bm = L.Bm, *Ak = L.Ak, *G = L.G, *Hm = L.Hm, *W = L.W, *T1 = L.T1, *T2 = L.T2, *T3 = L.T3;
    double *Rm = L.Rm, *X = L.X, *Y = L.Y, *Z = L.Z, *red = L.red;
    __syncthreads();                                             // the previous user of the work space is done
    // ---- central differences: lane j < n perturbs state j, lanes n..n+m-1 perturb effort j-n
    if (lane < n + m) {
        double xp[n], xm[n], tr[2 * S::NW + 1], uc[m];
        double xa[n], ua[m];
        for (int sgn = 0; sgn < 2; ++sgn) {
            for (int d = 0; d < n; ++d) xa[d] = x0[d];
            for (int j = 0; j < m; ++j) ua[j] = u0[j];
            const double h = sgn == 0 ? eps : -eps;
            if (lane < n) { for (int d = 0; d < n; ++d) if (d == lane) xa[d] += h; }
            else { for (int j = 0; j < m; ++j) if (j == lane - n) ua[j] += h; }
            trig_of<S>(xa, tr);
            for (int j = 0; j < m; ++j) uc[j] = ua[j];
            S::step(P, xa, tr, uc, dt, sgn == 0 ? xp : xm);
        }
        for (int d = 0; d < n; ++d) {
            const double v = (xp[d] - xm[d]) / (2.0 * eps);
            if (lane < n) A[d * n + lane] = v; else Bm[d * m + (lane - n)] = v;
        }
    }
    for (int i = lane; i < m * m; i += 64) Rm[i] = Rd[i];
    for (int i = lane; i < n * n; i += 64) Hm[i] = Qd[i];
    __syncthreads();
    for (int i = lane; i < n * n; i += 64) Ak[i] = A[i];
    // ---- G0 = B R^-1 B'
    for (int i = lane; i < m * n; i += 64) X[i] = Bm[(i % n) * m + (i / n)];     // X = B' (m x n)
    for (int i = lane; i < m * m; i += 64) Z[i] = Rm[i];
    __syncthreads();
    solve_inplace(Z, X, m, n, lane);                                           // X = R^-1 B'
    mm(G, Bm, X, n, m, n, false, false, lane);
    // ---- doubling
    int it = 0;
    for (; it < max_iter; ++it) {
        mm(W, G, Hm, n, n, n, false, false, lane);                             // W = G H
        for (int i = lane; i < n; i += 64) W[i * n + i] += 1.0;                 // W = I + G H
        for (int i = lane; i < n * n; i += 64) { T1[i] = Ak[i]; T2[i] = G[i]; }
        __syncthreads();
        // one elimination for both right-hand sides: [T1 | T2] <- W^-1 [A | G]
        for (int i = lane; i < n * n; i += 64) T3[i] = W[i];
        __syncthreads();
        solve_inplace(W, T1, n, n, lane);                                       // T1 = W^-1 A
        solve_inplace(T3, T2, n, n, lane);                                      // T2 = W^-1 G
        mm(W, Hm, T1, n, n, n, false, false, lane);                            // W = H W^-1 A
        mm(T3, Ak, W, n, n, n, true, false, lane);                             // T3 = A' H W^-1 A
        double dmax = 0.0, hmax = 0.0;
        for (int i = lane; i < n * n; i += 64) {
            const double hn = Hm[i] + T3[i];
            dmax = fmax(dmax, fabs(T3[i])); hmax = fmax(hmax, fabs(hn));
            Hm[i] = hn;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { dmax = fmax(dmax, __shfl_xor(dmax, off)); hmax = fmax(hmax, __shfl_xor(hmax, off)); }
        if (lane == 0) { red[0] = dmax; red[1] = hmax; }
        mm(W, Ak, T2, n, n, n, false, false, lane);                            // W = A W^-1 G
        mm(T3, W, Ak, n, n, n, false, true, lane);                             // T3 = A W^-1 G A'
        for (int i = lane; i < n * n; i += 64) G[i] += T3[i];
        mm(W, Ak, T1, n, n, n, false, false, lane);                            // W = A W^-1 A
        for (int i = lane; i < n * n; i += 64) Ak[i] = W[i];
        __syncthreads();
        if (red[0] <= tol * fmax(1.0, red[1])) { ++it; break; }
    }
    // ---- symmetrise, K = (R + B'SB)^-1 B'SA
    for (int i = lane; i < n * n; i += 64) T1[i] = 0.5 * (Hm[i] + Hm[(i % n) * n + (i / n)]);
    __syncthreads();
    mm(X, Bm, T1, m, n, n, true, false, lane);                                 // X = B' S     (m x n)
    mm(Z, X, Bm, m, n, m, false, false, lane);                                 // Z = B' S B   (m x m)
    for (int i = lane; i < m * m; i += 64) Z[i] += Rm[i];
    mm(Y, X, A, m, n, n, false, false, lane);                                  // Y = B' S A   (m x n)
    solve_inplace(Z, Y, m, n, lane);                                           // Y = K
    __syncthreads();
    return it;
}

template <class S>
__global__ __launch_bounds__(64) void k_lqr_dare(Params P, const double* __restrict__ xs, const double* __restrict__ us, int B,
                                                 const double* __restrict__ Qd, const double* __restrict__ Rd, double dt, double eps,
                                                 int max_iter, double tol, double* __restrict__ S_out, double* __restrict__ K_out,
                                                 double* __restrict__ A_out, double* __restrict__ B_out, int* __restrict__ iters_out) {
    constexpr int n = S::N, m = S::M;
    __shared__ DareLds<n, m> L;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    double x0[n], u0[m];
    for (int d = 0; d < n; ++d) x0[d] = xs[(size_t)b * n + d];
    for (int j = 0; j < m; ++j) u0[j] = us ? us[(size_t)b * m + j] : 0.0;
    const int it = dare_lqr<S>(P.p, x0, u0, Qd, Rd, dt, eps, max_iter, tol, L, lane);
    if (S_out) for (int i = lane; i < n * n; i += 64) S_out[(size_t)b * n * n + i] = L.T1[i];
    if (K_out) for (int i = lane; i < m * n; i += 64) K_out[(size_t)b * m * n + i] = L.Y[i];
    if (A_out) for (int i = lane; i < n * n; i += 64) A_out[(size_t)b * n * n + i] = L.A[i];
    if (B_out) for (int i = lane; i < n * m; i += 64) B_out[(size_t)b * n * m + i] = L.Bm[i];
    if (iters_out && lane == 0) iters_out[b] = it;
}

}  // namespace lq
