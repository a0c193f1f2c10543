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
#include "measure.hpp"
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

// Gauss-Jordan elimination of [W | RHS] (W n x n, RHS n x q; partial pivoting: largest |entry| of the column, lowest row on ties)
// with one COLUMN per lane: lane j < n + q holds c[r] = entry (r, j), r < n, in registers.  A pivot costs n pairs of v_readlane
// (column p, wave-uniform from then on: the scan for the pivot row, the pivot and every row's multiplier are scalar values), one
// division and n - 1 multiply-subtracts per lane -- no LDS traffic, no cross-lane gather, no barrier, and the row exchange is a
// uniform branch around register moves.  (Round 5.  The forms this replaces kept one ENTRY per lane: every pivot then paid two
// ds_bpermute gathers per register slot plus the exchange's, ~230 instructions per pivot at n = 6 with two slots against ~70 here,
// and a lone wavefront pays ~6 cycles per instruction whatever it does.)  The sums of an entry run in one lane in a fixed order and
// a right-hand-side column never looks at another one, so a column gets the same bits whether it is solved alone or next to others
// -- and the same bits as the entry-per-lane forms gave: same pivots, same quotient, same multiply, same subtraction
// (tools/dare_ab.sh compares two builds bit for bit; oracle/lqrrt_oracle.c restates the elimination sequentially).
// On return the RHS columns hold W^-1 RHS (the W columns hold the identity).
__device__ __forceinline__ double readlane_f64(double v, int l) {        // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Maximum over the 64 lanes, the same value in every lane on return.  Data-parallel-primitive moves inside the VALU (four shifts
// within a row of 16, two row broadcasts, one v_readlane pair) instead of six butterfly stages through the LDS crossbar
// (ds_bpermute: ~100 cycles of latency each for a lone wavefront).  Lanes without a source keep their own value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const int hi = __double2hiint(v), lo = __double2loint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_move_f64<0x111, 0xf>(v));                    // row_shr:1
    v = fmax(v, dpp_move_f64<0x112, 0xf>(v));                    // row_shr:2
    v = fmax(v, dpp_move_f64<0x114, 0xf>(v));                    // row_shr:4
    v = fmax(v, dpp_move_f64<0x118, 0xf>(v));                    // row_shr:8: lane 15 of a row holds the row's maximum
    v = fmax(v, dpp_move_f64<0x142, 0xa>(v));                    // row_bcast:15 into rows 1 and 3
    v = fmax(v, dpp_move_f64<0x143, 0xc>(v));                    // row_bcast:31 into rows 2 and 3: lane 63 holds the maximum
    return readlane_f64(v, 63);
}

template <int n, int q>
__device__ __forceinline__ void gj_columns(double (&c)[n]) {
    static_assert(n + q <= 64, "one lane per column of [W | RHS]");
#pragma unroll
    for (int p = 0; p < n; ++p) {
        double f[n];                                             // column p: the same values in every lane
#pragma unroll
        for (int r = 0; r < n; ++r) f[r] = readlane_f64(c[r], p);
        if (p + 1 < n) {                                         // (p is a compile-time value: the loop is unrolled)
            // Does row p stay?  It does unless a row below holds a strictly larger |entry| (lowest row on ties).  Every lane asks
            // its own column in its own registers (n - p - 1 v_max and one compare); lane p's answer is the one that counts.  Only
            // when it says no -- rare -- is the row looked for and exchanged: a real (uniform) branch, the empty asm keeps the
            // compiler from turning the exchange into 4 (n - p - 1) selects on every pivot.  (A NaN in row p answers no and takes
            // the scan, which then does what it always did.)
            double rest = -1.0;
#pragma unroll
            for (int r = p + 1; r < n; ++r) rest = fmax(rest, fabs(c[r]));
            const bool stays = fabs(c[p]) >= rest;
            if (!((__ballot(stays) >> p) & 1ull)) {
                asm volatile("; row exchange");
                double best = -1.0;
                int brow_v = p;
#pragma unroll
                for (int r = p; r < n; ++r) {
                    const double a = fabs(c[r]);
                    if (a > best) { best = a; brow_v = r; }
                }
                const int brow = __builtin_amdgcn_readlane(brow_v, p);
#pragma unroll
                for (int r = p + 1; r < n; ++r) {
                    if (brow == r) {
                        const double tc = c[p]; c[p] = c[r]; c[r] = tc;
                        const double tf = f[p]; f[p] = f[r]; f[r] = tf;
                    }
                }
            }
        }
        const double y = c[p] / f[p];                            // the scaled pivot row, column of this lane
#pragma unroll
        for (int r = 0; r < n; ++r) {
            if (r == p) continue;
            double x = c[r];
            x -= f[r] * y;
            c[r] = x;
        }
        c[p] = y;
    }
}

// Solves W X = RHS in place; W is n x n, RHS is n x q, row-major, both in LDS; W is left as it was, RHS becomes X.  The first
// wavefront of the workgroup eliminates (gj_columns), the others wait at the barrier.
template <int n, int q, int NT = 64>
__device__ __forceinline__ void solve_inplace(double* W, double* RHS, int tid) {
    constexpr int C = n + q;
    static_assert(C <= 64, "one lane per column of [W | RHS]");
    if (tid < 64) {
        double c[n];
#pragma unroll
        for (int r = 0; r < n; ++r) c[r] = tid < n ? W[r * n + tid] : (tid < C ? RHS[r * q + (tid - n)] : 0.0);
        gj_columns<n, q>(c);
        if (tid >= n && tid < C) {
#pragma unroll
            for (int r = 0; r < n; ++r) RHS[r * q + (tid - n)] = c[r];
        }
    }
    __syncthreads();
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
// ~6 cycles per instruction whatever it does; with four wavefronts every matrix pass of a doubling iteration has one entry per lane.
// The elimination of [I + G H | A_k | G] is the first wavefront's alone (gj_columns: one column per lane, 3 n lanes; round 5 -- until
// then it was done four times side by side with one ENTRY per lane, ~2/3 of an iteration's instructions).  Same entries, same sums,
// same order in both sizes: lqrrt_lqr_dare_batch and the per-sample S table (64 threads) and the rollouts (256) produce the same
// bits, asserted on the GPU (tests/test_dare_gpu.py).
template <class S, int NT = 64>
__device__ __forceinline__ int dare_lqr(const double* P, const double* x0, const double* u0, const double* Qd, const double* Rd,
                                        double dt, double eps, int max_iter, double tol, DareLds<S::N, S::M>& L, int tid) {
    static_assert(NT == 64 || NT == 256, "one or four wavefronts");
    const int lane = tid & 63;
    constexpr int n = S::N, m = S::M;
    double *A = L.A, *Bm = L.Bm, *Ak = L.Ak, *G = L.G, *Hm = L.Hm, *W = L.W, *T1 = L.T1, *T2 = L.T2, *T3 = L.T3;
    double *Rm = L.Rm, *X = L.X, *Y = L.Y, *Z = L.Z, *red = L.red, *AG = L.AG;
    __syncthreads();                                             // the previous user of the work space is done
    DARE_TS(ts0);
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
    DARE_TS(ts1); DARE_ACC(0, ts0, ts1);
    for (int i = tid; i < n * n; i += NT) Ak[i] = A[i];
    // ---- G0 = B R^-1 B'
    for (int i = tid; i < m * n; i += NT) X[i] = Bm[(i % n) * m + (i / n)];     // X = B' (m x n)
    for (int i = tid; i < m * m; i += NT) Z[i] = Rm[i];
    __syncthreads();
    solve_inplace<m, n, NT>(Z, X, tid);                                        // X = R^-1 B'
    mm<NT>(G, Bm, X, n, m, n, false, false, tid);
    DARE_TS(ts2); DARE_ACC(1, ts1, ts2);
    // ---- doubling
    int it = 0;
    if constexpr (n * 3 * n <= 128) {
        // Four passes and four barriers per iteration: (1) W = I + G H; (2) [W | A_k | G] into registers, eliminated there
        // (gj_columns), T1 = W^-1 A_k and T2 = W^-1 G out; (3) the three products that only need T1 / T2 side by side; (4) the two
        // that update H and G, accumulated in place.  A_k and its successor swap buffers instead of being copied.  Every entry is
        // the sum the plain sequence below forms, term by term.  Pass (1) of iteration k + 1 runs beside the reduction of iteration
        // k's convergence test (round 5): the test's verdict is read behind the barrier that pass (1) needs anyway.
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
        static_assert(n * n <= 64, "an n x n pass fits one wavefront");
        constexpr int HW = NT > 64 ? 1 : 0;                // the wavefront that owns the entries of H in pass (4) and reduces the test's maxima
        auto pass_W = [&]() __attribute__((always_inline)) {
            if (tid < n * n) {
                const int i = tid / n, j = tid % n;
                double acc = dot(G, Hm, i, j, false, false);
                if (i == j) acc += 1.0;
                W[tid] = acc;                                                       // W = I + G H
            }
        };
        DARE_TS(ta0);
        pass_W();
        __syncthreads();
        DARE_TS(tb0); DARE_ACC(2, ta0, tb0);
        while (it < max_iter) {
            DARE_TS(tb);
            if (tid < 64) {
                // lane j < 3 n holds column j of [W | A_k | G]; the workgroup's other wavefronts (NT = 256) wait at the barrier
                const double* src = lane < n ? W + lane : (lane < 2 * n ? Akc + (lane - n) : (lane < 3 * n ? G + (lane - 2 * n) : W));
                double c[n];
#pragma unroll
                for (int r = 0; r < n; ++r) c[r] = src[r * n];                      // (lanes >= 3 n: column 0 once more, never stored)
                gj_columns<n, 2 * n>(c);                                            // [T1 | T2] = W^-1 [A | G]
                if (lane >= n && lane < 3 * n) {
                    double* dst = lane < 2 * n ? T1 + (lane - n) : T2 + (lane - 2 * n);
#pragma unroll
                    for (int r = 0; r < n; ++r) dst[r * n] = c[r];
                }
            }
            __syncthreads();
            DARE_TS(tc); DARE_ACC(3, tb, tc);
            for (int idx = tid; idx < 3 * n * n; idx += NT) {
                const int w = idx / (n * n), e = idx % (n * n), i = e / n, j = e % n;
                const double acc = dot(w == 0 ? Hm : Akc, w == 1 ? T2 : T1, i, j, false, false);
                (w == 0 ? W1 : (w == 1 ? W2 : Akn))[e] = acc;                       // H W^-1 A | A W^-1 G | A W^-1 A
            }
            __syncthreads();
            DARE_TS(td); DARE_ACC(4, tc, td);
            // (4) G in the first wavefront; H in wavefront HW (the first one again when there is only one), which keeps |increment| and
            // |entry| of its lane for the test
            double dmax = 0.0, hmax = 0.0;
            if (tid < n * n) {
                const int i = tid / n, j = tid % n;
                const double t3 = dot(W2, Akc, i, j, false, true);                  // A W^-1 G A'
                G[tid] += t3;
            }
            const int eh = tid - 64 * HW;
            if (eh >= 0 && eh < n * n) {
                const int i = eh / n, j = eh % n;
                const double t3 = dot(Akc, W1, i, j, true, false);                  // A' H W^-1 A
                const double hn = Hm[eh] + t3;
                dmax = fabs(t3); hmax = fabs(hn);
                Hm[eh] = hn;
            }
            __syncthreads();                                                        // H, G are there
            DARE_TS(te); DARE_ACC(5, td, te);
            // The next iteration's W beside the reduction of the test's maxima (two wavefronts when there are four; W is wasted when the
            // test then says stop -- it is a scratch matrix)
            pass_W();
            if ((tid >> 6) == HW) {
                dmax = wave_max(dmax); hmax = wave_max(hmax);                       // (a maximum does not care about the order)
                if (lane == 0) { L.redw[0] = dmax; L.redw[1] = hmax; }
            }
            __syncthreads();                                                        // W and the maxima are there
            dmax = L.redw[0]; hmax = L.redw[1];                                     // (redw is next written four barriers from here)
            double* tsw = Akc; Akc = Akn; Akn = tsw;
            ++it;
            DARE_TS(tf); DARE_ACC(2, te, tf); DARE_ACC(8, 0ull, 1ull);
            if (dmax <= tol * fmax(1.0, hmax)) break;
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
    DARE_TS(ts3);
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
    DARE_TS(ts4); DARE_ACC(6, ts3, ts4); DARE_ACC(7, 0ull, 1ull);
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
