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
template <int NT = 64>
__device__ __forceinline__ void mm(double* C, const double* A, const double* B, int r, int k, int c, bool ta, bool tb, int lane) {
    for (int idx = lane; idx < r * c; idx += NT) {
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

// Solves W X = RHS in place (Gauss-Jordan, partial pivoting: largest |entry| of the column, lowest row on ties); W is n x n,
// RHS is n x q, row-major; W is destroyed, RHS becomes X.  One wavefront; the sums of an entry run in one lane in a fixed order,
// so every right-hand-side column gets the same bits whether it is solved alone or next to others.
//   per pivot: every lane scans the column itself (n broadcast LDS reads instead of a 6-stage butterfly), lanes j < n + q
//   divide the pivot row once, and one pass over the n (n + q) entries scales and eliminates (reads, barrier, writes).
__device__ __forceinline__ double readlane_f64(double v, int l) {        // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Gauss-Jordan elimination of [W | RHS] (n x (n + q), up to 128 entries) held in registers: entry e = r C + j in lane e % 64,
// slot e / 64.  A pivot costs no LDS traffic and no barrier: the column scan and the pivot are v_readlane's (uniform positions),
// the pivot-row entry and the row's multiplier of each lane two cross-lane gathers.  Same operations on the same values as the
// LDS form in solve_inplace, entry by entry.  On return the RHS columns hold W^-1 RHS.
template <int n, int q>
__device__ __forceinline__ void gj_eliminate(double (&v)[(n * (n + q) + 63) / 64], int lane) {
    constexpr int C = n + q, E = n * C, IT = (E + 63) / 64;
    static_assert(IT <= 2, "register form: at most 128 entries");
    int rr[IT], jj[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) { rr[k] = (lane + 64 * k) / C; jj[k] = (lane + 64 * k) % C; }
    auto elem = [&](int e) __attribute__((always_inline)) -> double {      // entry e, e wave-uniform
        double a = readlane_f64(v[0], e & 63);
        if constexpr (IT > 1) { const double b = readlane_f64(v[1], e & 63); a = e >= 64 ? b : a; }
        return a;
    };
    auto gather = [&](int e) __attribute__((always_inline)) -> double {    // entry e, e per lane
        double a = __shfl(v[0], e & 63);
        if constexpr (IT > 1) { const double b = __shfl(v[1], e & 63); a = e >= 64 ? b : a; }
        return a;
    };
#pragma unroll 1
    for (int p = 0; p < n; ++p) {
        double best = -1.0;
        int brow = p;
#pragma unroll
        for (int r = 0; r < n; ++r) {
            if (r < p) continue;
            const double a = fabs(elem(r * C + p));
            if (a > best) { best = a; brow = r; }
        }
        brow = __builtin_amdgcn_readfirstlane(brow);
        if (brow != p) {
            double t[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int src = rr[k] == p ? brow * C + jj[k] : (rr[k] == brow ? p * C + jj[k] : lane + 64 * k);
                t[k] = gather(src);
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) v[k] = t[k];
        }
        const double piv = elem(p * C + p);
        double nv[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const double y = gather(p * C + jj[k]) / piv;                    // the scaled pivot row, column of this lane
            const double f = gather(rr[k] * C + p);
            double x = v[k];
            x -= f * y;
            nv[k] = rr[k] == p ? y : x;
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) v[k] = nv[k];
    }
}

template <int n, int q, int NT = 64>
__device__ __forceinline__ void solve_inplace(double* W, double* RHS, int tid) {
    constexpr int C = n + q, E = n * C, IT = (E + 63) / 64;
    static_assert(C <= 64, "one lane per column of [W | RHS]");
    static_assert(NT == 64 || IT <= 2, "larger workgroups: register form only (their first wavefront solves, the others wait)");
    const int lane = tid & 63;
    if constexpr (IT <= 2) {
        if (tid < 64) {
            double v[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int idx = lane + 64 * k, r = idx / C, j = idx % C;
                v[k] = idx < E ? (j < n ? W[r * n + j] : RHS[r * q + (j - n)]) : 0.0;
            }
            gj_eliminate<n, q>(v, lane);
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int idx = lane + 64 * k, r = idx / C, j = idx % C;
                if (idx < E && j >= n) RHS[r * q + (j - n)] = v[k];
            }
        }
        __syncthreads();
        return;
    }
#pragma unroll 1
    for (int p = 0; p < n; ++p) {
        double best = -1.0;
        int brow = p;
#pragma unroll
        for (int r = 0; r < n; ++r) {
            if (r < p) continue;
            const double v = fabs(W[r * n + p]);
            if (v > best) { best = v; brow = r; }
        }
        brow = __builtin_amdgcn_readfirstlane(brow);
        if (brow != p) {
            if (lane < C) {
                double* a = lane < n ? &W[p * n + lane] : &RHS[p * q + (lane - n)];
                double* b = lane < n ? &W[brow * n + lane] : &RHS[brow * q + (lane - n)];
                const double t = *a; *a = *b; *b = t;
            }
            __syncthreads();
        }
        const double piv = W[p * n + p];
        double yl = 0.0;
        if (lane < C) yl = (lane < n ? W[p * n + lane] : RHS[p * q + (lane - n)]) / piv;       // the scaled pivot row
        double nv[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int idx = lane + 64 * k;
            const int r = idx / C, j = idx % C;
            const double y = __shfl(yl, j);
            nv[k] = y;
            if (idx < E && r != p) {
                const double f = W[r * n + p];
                double x = j < n ? W[r * n + j] : RHS[r * q + (j - n)];
                x -= f * y;
                nv[k] = x;
            }
        }
        __syncthreads();                                          // every read of this sweep is done
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int idx = lane + 64 * k;
            const int r = idx / C, j = idx % C;
            if (idx < E) { if (j < n) W[r * n + j] = nv[k]; else RHS[r * q + (j - n)] = nv[k]; }
        }
        __syncthreads();
    }
}

// LDS work space of one problem (one wavefront)
template <int n, int m>
struct DareLds {
    double A[n * n], Bm[n * m], Ak[n * n], G[n * n], Hm[n * n], W[n * n], T1[n * n], T2[n * n], T3[n * n];
    double Rm[m * m], X[m * n], Y[m * n], Z[m * m];
    double AG[n * 2 * n];                                         // [A_k | G]: both right-hand sides of one elimination
    double red[2];
    double redw[8];                                               // four wavefronts: their maxima of a doubling iteration
};

// lqr(x, u) of the API contract for one (x0, u0), computed by the calling workgroup of NT = 64 or 256 threads (all of them must
// call; x0 / u0 are uniform per-thread arrays; `tid` = thread index in the workgroup): A, B by central differences of S::step, S by
// doubling, K = (R + B'SB)^-1 B'SA.
// P: model parameters; Qd (n x n), Rd (m x m): weights (any address space).  Results are left in the work space:
// L.T1 = S (symmetrised), L.Y = K (m x n), L.A / L.Bm = the linearisation.  Returns the doubling iterations used.
// Every sum runs in a fixed order inside ONE lane (mm, solve_inplace), so the result does not depend on the lane
// count and oracle/lqrrt_oracle.c restates it sequentially (two eliminations there, one here: same bits per column).
// NT = 256 (round 4; the rollouts of Riccati systems, k_steer<S, DENSE, 4>): four wavefronts on four SIMDs.  A lone wavefront pays
// ~6 cycles per instruction whatever it does, and one doubling iteration at n = 6 is ~700 instructions, two thirds of them the
// Gauss-Jordan elimination of [I + G H | A_k | G] (108 entries = two register slots per lane).  With four wavefronts every matrix
// pass has one entry per lane, and the elimination is done FOUR times side by side, each wavefront with [I + G H | n/2 of the 2n
// right-hand-side columns] in ONE register slot: the pivoting only looks at I + G H, so all four take the same pivots, and a
// right-hand-side column gets the same bits whoever solves it (gj_eliminate).  Same entries, same sums, same order: the bits of
// the 64-thread form (which lqrrt_lqr_dare_batch and the per-sample S table keep using), asserted on the GPU
// (tests/test_dare_gpu.py).
template <class S, int NT = 64>
__device__ __forceinline__ int dare_lqr(const double* P, const double* x0, const double* u0, const double* Qd, const double* Rd,
                                        double dt, double eps, int max_iter, double tol, DareLds<S::N, S::M>& L, int tid) {
    static_assert(NT == 64 || NT == 256, "one or four wavefronts");
    const int lane = tid & 63;
    constexpr int n = S::N, m = S::M;
    double *A = L.A, *Bm = L.Bm, *Ak = L.Ak, *G = L.G, *Hm = L.Hm, *W = L.W, *T1 = L.T1, *T2 = L.T2, *T3 = L.T3;
    double *Rm = L.Rm, *X = L.X, *Y = L.Y, *Z = L.Z, *red = L.red, *AG = L.AG;
    __syncthreads();                                             // the previous user of the work space is done
    // ---- central differences: lane j < n perturbs state j, lanes n..n+m-1 perturb effort j-n; lanes 32.. take the minus side
    static_assert(n + m <= 32, "plus and minus sides of the difference quotients share the wavefront");
    if (tid < 64) {
        const int col = lane & 31;
        const bool minus = lane >= 32;
        double xo[n];
        if (col < n + m) {
            double tr[2 * S::NW + 1], uc[m], xa[n], ua[m];
            for (int d = 0; d < n; ++d) xa[d] = x0[d];
            for (int j = 0; j < m; ++j) ua[j] = u0[j];
            const double h = minus ? -eps : eps;
            if (col < n) { for (int d = 0; d < n; ++d) if (d == col) xa[d] += h; }
            else { for (int j = 0; j < m; ++j) if (j == col - n) ua[j] += h; }
            trig_of<S>(xa, tr);
            for (int j = 0; j < m; ++j) uc[j] = ua[j];
            S::step(P, xa, tr, uc, dt, xo);
        } else {
            for (int d = 0; d < n; ++d) xo[d] = 0.0;
        }
        for (int d = 0; d < n; ++d) {
            const double xm = __shfl_xor(xo[d], 32);
            const double v = (xo[d] - xm) / (2.0 * eps);
            if (!minus && col < n + m) { if (col < n) A[d * n + col] = v; else Bm[d * m + (col - n)] = v; }
        }
    }
    for (int i = tid; i < m * m; i += NT) Rm[i] = Rd[i];
    for (int i = tid; i < n * n; i += NT) Hm[i] = Qd[i];
    __syncthreads();
    for (int i = tid; i < n * n; i += NT) Ak[i] = A[i];
    // ---- G0 = B R^-1 B'
    for (int i = tid; i < m * n; i += NT) X[i] = Bm[(i % n) * m + (i / n)];     // X = B' (m x n)
    for (int i = tid; i < m * m; i += NT) Z[i] = Rm[i];
    __syncthreads();
    solve_inplace<m, n, NT>(Z, X, tid);                                        // X = R^-1 B'
    mm<NT>(G, Bm, X, n, m, n, false, false, tid);
    // ---- doubling
    int it = 0;
    if constexpr (n * 3 * n <= 128) {
        // Four passes and four barriers per iteration: (1) W = I + G H; (2) [W | A_k | G] into registers, eliminated there
        // (gj_eliminate), T1 = W^-1 A_k and T2 = W^-1 G out; (3) the three products that only need T1 / T2 side by side; (4) the two
        // that update H and G, accumulated in place.  A_k and its successor swap buffers instead of being copied.  Every entry is
        // the sum the plain sequence below forms, term by term.
        auto dot = [&](const double* Am, const double* Bq, int i, int j, bool ta, bool tb) __attribute__((always_inline)) -> double {
            double acc = 0.0;
#pragma unroll
            for (int pp = 0; pp < n; ++pp) {
                const double a = ta ? Am[pp * n + i] : Am[i * n + pp];
                const double b = tb ? Bq[j * n + pp] : Bq[pp * n + j];
                acc += a * b;
            }
            return acc;
        };
        double* Akc = Ak;                                  // A_k of this iteration
        double* Akn = AG;                                  // ... of the next one
        double* const W1 = T3;                             // H W^-1 A
        double* const W2 = AG + n * n;                     // A W^-1 G
        constexpr int C3 = 3 * n, E3 = n * C3, IT3 = (E3 + 63) / 64;
        constexpr int QW = n / 2, CW = n + QW, EW = n * CW;          // NT = 256: columns of [W | its share of A_k | G] per wavefront
        static_assert(NT == 64 || (n % 2 == 0 && EW <= 64), "four wavefronts: n/2 right-hand-side columns each, one register slot");
        for (; it < max_iter; ++it) {
            for (int e = tid; e < n * n; e += NT) {
                const int i = e / n, j = e % n;
                double acc = dot(G, Hm, i, j, false, false);
                if (i == j) acc += 1.0;
                W[e] = acc;                                                         // W = I + G H
            }
            __syncthreads();
            if constexpr (NT == 64) {
                double v[IT3];
#pragma unroll
                for (int k = 0; k < IT3; ++k) {
                    const int idx = lane + 64 * k, r = idx / C3, j = idx % C3;
                    v[k] = idx < E3 ? (j < n ? W[r * n + j] : (j < 2 * n ? Akc[r * n + (j - n)] : G[r * n + (j - 2 * n)])) : 0.0;
                }
                gj_eliminate<n, 2 * n>(v, lane);                                    // [T1 | T2] = W^-1 [A | G]
#pragma unroll
                for (int k = 0; k < IT3; ++k) {
                    const int idx = lane + 64 * k, r = idx / C3, j = idx % C3;
                    if (idx < E3 && j >= n) { if (j < 2 * n) T1[r * n + (j - n)] = v[k]; else T2[r * n + (j - 2 * n)] = v[k]; }
                }
            } else {
                // wavefront w solves columns [w QW, (w+1) QW) of [A_k | G]: w = 0, 1 give T1, w = 2, 3 give T2
                const int w = tid >> 6, r = lane / CW, j = lane % CW;
                const int col = w * QW + (j - n);                                   // column of [A_k | G] (j >= n)
                double v[1];
                v[0] = lane < EW ? (j < n ? W[r * n + j] : (col < n ? Akc[r * n + col] : G[r * n + (col - n)])) : 0.0;
                gj_eliminate<n, QW>(v, lane);
                if (lane < EW && j >= n) { if (col < n) T1[r * n + col] = v[0]; else T2[r * n + (col - n)] = v[0]; }
            }
            __syncthreads();
            for (int idx = tid; idx < 3 * n * n; idx += NT) {
                const int w = idx / (n * n), e = idx % (n * n), i = e / n, j = e % n;
                const double acc = dot(w == 0 ? Hm : Akc, w == 1 ? T2 : T1, i, j, false, false);
                (w == 0 ? W1 : (w == 1 ? W2 : Akn))[e] = acc;                       // H W^-1 A | A W^-1 G | A W^-1 A
            }
            __syncthreads();
            double dmax = 0.0, hmax = 0.0;
            for (int idx = tid; idx < 2 * n * n; idx += NT) {
                const int w = idx / (n * n), e = idx % (n * n), i = e / n, j = e % n;
                if (w == 0) {
                    const double t3 = dot(Akc, W1, i, j, true, false);              // A' H W^-1 A
                    const double hn = Hm[e] + t3;
                    dmax = fmax(dmax, fabs(t3)); hmax = fmax(hmax, fabs(hn));
                    Hm[e] = hn;
                } else {
                    const double t3 = dot(W2, Akc, i, j, false, true);              // A W^-1 G A'
                    G[e] += t3;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { dmax = fmax(dmax, __shfl_xor(dmax, off)); hmax = fmax(hmax, __shfl_xor(hmax, off)); }
            if constexpr (NT > 64) {                                                // (a maximum does not care about the order)
                if (lane == 0) { L.redw[2 * (tid >> 6)] = dmax; L.redw[2 * (tid >> 6) + 1] = hmax; }
            }
            __syncthreads();                                                        // H, G (and the wavefronts' maxima) are there
            if constexpr (NT > 64) {                                                // (redw is next written three barriers from here)
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) { dmax = fmax(dmax, L.redw[2 * w]); hmax = fmax(hmax, L.redw[2 * w + 1]); }
            }
            double* tsw = Akc; Akc = Akn; Akn = tsw;
            if (dmax <= tol * fmax(1.0, hmax)) { ++it; break; }
        }
    } else {
    static_assert(NT == 64 || n * 3 * n <= 128, "the four-wavefront form exists for the register elimination only");
    for (; it < max_iter; ++it) {
        mm(W, G, Hm, n, n, n, false, false, lane);                             // W = G H
        for (int i = lane; i < n; i += 64) W[i * n + i] += 1.0;                 // W = I + G H
        for (int i = lane; i < n * n; i += 64) { const int r = i / n, c = i % n; AG[r * 2 * n + c] = Ak[i]; AG[r * 2 * n + n + c] = G[i]; }
        __syncthreads();
        solve_inplace<n, 2 * n>(W, AG, lane);                                   // one elimination: [T1 | T2] = W^-1 [A | G]
        for (int i = lane; i < n * n; i += 64) { const int r = i / n, c = i % n; T1[i] = AG[r * 2 * n + c]; T2[i] = AG[r * 2 * n + n + c]; }
        __syncthreads();
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
    }
    // ---- symmetrise, K = (R + B'SB)^-1 B'SA
    for (int i = tid; i < n * n; i += NT) T1[i] = 0.5 * (Hm[i] + Hm[(i % n) * n + (i / n)]);
    __syncthreads();
    mm<NT>(X, Bm, T1, m, n, n, true, false, tid);                              // X = B' S     (m x n)
    for (int idx = tid; idx < m * m + m * n; idx += NT) {                      // Z = R + B' S B (m x m) and Y = B' S A (m x n) side by side
        if (idx < m * m) {
            const int i = idx / m, j = idx % m;
            double acc = 0.0;
            for (int pp = 0; pp < n; ++pp) acc += X[i * n + pp] * Bm[pp * m + j];
            Z[idx] = acc + Rm[idx];
        } else {
            const int e = idx - m * m, i = e / n, j = e % n;
            double acc = 0.0;
            for (int pp = 0; pp < n; ++pp) acc += X[i * n + pp] * A[pp * n + j];
            Y[e] = acc;
        }
    }
    __syncthreads();
    solve_inplace<m, n, NT>(Z, Y, tid);                                        // Y = K
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
