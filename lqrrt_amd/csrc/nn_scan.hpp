// nn_scan.hpp -- the cost-to-go nearest-neighbour kernels: k_nn_scan (tree and in-wave scans), k_nn_reduce, candidate pack / unpack of tree-sharded waves, k_costs.
// Fragment of kernels.hpp (included there, in order, inside namespace lq).
#pragma once

// ------------------------------------------------------------------------------------------
// NN scan.  grid = (ceil(W/64), n_chunks), block = 64 (one wavefront).  Lane = sample; the loop over the chunk's
// nodes is WAVE-UNIFORM, so a node's data (state, trig, eligibility) is the same for all 64 lanes: it is fetched
// by the scalar unit (s_load through the scalar cache, straight from the node table in L2) into SGPRs and used as
// the scalar operand of the per-lane fp64 arithmetic.  Nothing is staged in LDS and no vector-memory or LDS
// instruction sits in the inner loop -- round 1 read every node with five ds_read_b128 broadcasts per wavefront and
// was bound by the LDS pipe, not by the VALU.  Ineligible nodes (ignore bit / empty in-wave record) are skipped by a
// scalar branch before any arithmetic.
// TRI: only nodes with index < sample index are eligible (in-wave pass; a separate instantiation so that profiles
// tell it apart from the tree scan).
// Output: partial minima over the eligible nodes at [chunk * ps_c + sample * ps_t]: the tree scan writes
// sample-major (ps_c = 1) so that the reduce reads a sample's partials contiguously; the in-wave scan writes
// chunk-major (ps_t = 1), the order k_decide wants.
// xtrig: cos/sin of the samples' angular coordinates [W][2*NW] if the caller has them (the engine computes them once
// per sample batch), else null and they are computed here.
// WPB = 4 (round 4, two-level reduction; LQRRT_NN_WG4): four wavefronts per workgroup scan four consecutive chunks and reduce
// their minima through LDS, so a sample gets ONE partial per four chunks: a quarter of the scattered 12-byte stores (each of
// them a 64-byte transaction: 58 % of the scan's physical traffic, profiles/r03_nn_traffic.json) and a quarter of the partials
// the steer prologue has to read back.  Chunks ascend in node id, the combination keeps the (cost, id) order.
// The body of a scan launch for workgroup `b` of a gx x gy grid (linear id, x fastest): k_nn_scan (one engine's launch) and
// k_nn_scan_multi (one launch whose grid spans several engines, lqrrt_engine_extend_multi) both run it.  pt_n: entries of `pt` to apply.
template <class S, int DENSE, bool TRI, bool PATCH, int WPB>
__device__ __forceinline__ void nn_scan_body(const NodeView& nv, const double* __restrict__ xs, const double* __restrict__ xtrig,
                                             const int W, const double* __restrict__ Sd, const int chunk,
                                             Part* __restrict__ part, int* __restrict__ tri_idx,
                                             const int ps_c, const int ps_t, const IgnPatch& pt, const int pt_n,
                                             const int b, const int gx, const int gy) {
    static_assert(WPB == 1 || (!PATCH && !TRI), "the four-wavefront form exists for the plain tree scan");
    const int lane = threadIdx.x & 63;
    // patch entry k lives in lane k (and k + 16, ...): one vector load each, issued with the launch's first loads -- the
    // argument block is not in any cache yet, and a lookup that went back to it per tile cost the launch ~2 us
    int pt_idx = -1;
    unsigned long long pt_val = 0;
    if constexpr (PATCH) {
        if (pt_n > 0) {
            pt_idx = pt.idx[lane & 15]; pt_val = pt.val[lane & 15];
            if (b == 0 && lane < pt_n && nv.ignore) const_cast<unsigned long long*>(nv.ignore)[pt_idx] = pt_val;
        }
    }
    // XCD-aware tile mapping: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each
    // with its own L2.  Re-index so that XCD k owns a contiguous band of node chunks (for every sample
    // group): each L2 then holds 1/8 of the node table instead of all of it.  Speed only; any mapping is
    // correct because every (group, chunk) pair is still visited exactly once.  (A launch that spans several engines starts every
    // engine's range at a multiple of 8 workgroups, so b & 7 is the XCD there too.)
    int bx = b % gx, by = b / gx;
    {
        const int nb = gx * gy;
        if ((nb & 7) == 0) {
            const int v = (b & 7) * (nb >> 3) + (b >> 3);
            bx = v % gx;
            by = v / gx;
        }
    }
    const int t = bx * 64 + lane;
    const int ts = t < W ? t : W - 1;
    // The chunk index must be visibly wave-uniform: the node loop below is fed by the scalar unit only if `base` lives in an SGPR.
    // Round 4 wrote `by * WPB + (threadIdx.x >> 6)` for every WPB; the compiler does not fold the shift for WPB == 1, the loop
    // index became a vector value, every node fetch a vector load, and the kernels grew from 117 (tree scan) / 96 (in-wave scan) to
    // 155 / 174 VGPRs, i.e. from 4 / 5 to 3 / 2 wavefronts per SIMD: W = 1024 x 10k nodes 13 -> 27 us (profiles/r05_nn_regression.txt;
    // tests/test_abi_cpu.py pins the register counts of these instantiations now).
    int wchunk = by;
    if constexpr (WPB > 1) wchunk = by * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = nv.first + wchunk * chunk;
    int i1 = i0 + chunk;
    if (i1 > nv.first + nv.count) i1 = nv.first + nv.count;
    if constexpr (TRI) {
        const int tmax = bx * 64 + 63;
        if (i1 > tmax) i1 = tmax;
    }
    // does a patched ignore word cover nodes of this workgroup's chunk at all?  (a hit's path: a few words, mostly the newest
    // nodes -- nearly every workgroup skips the patch lookup below)
    bool patched = false;
    if constexpr (PATCH) {
        const int w0 = i0 >> 6, w1 = (i1 - 1) >> 6;
        if (pt_n > 0) patched = __any(lane < pt_n && pt_idx >= w0 && pt_idx <= w1) != 0;
    }
    double xg[S::N], gtrig[2 * S::NW + 1];
#pragma unroll
    for (int d = 0; d < S::N; ++d) xg[d] = xs[(size_t)ts * S::N + d];
    if (xtrig) {
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) gtrig[j] = xtrig[(size_t)ts * (2 * S::NW) + j];
    } else {
        trig_of<S>(xg, gtrig);
    }
    // Angle errors.  mode 2: every sample of this wavefront has the sampler's fixed angular coordinates and the tree
    // carries the nodes' errors w.r.t. them (NodeView::werr): the error is one more scalar load per node.  mode 1: the
    // wavefront's samples share their angular coordinates (any value): lane j computes the error of node j of a
    // 64-node tile once and the node loop pulls it out of that lane with v_readlane (no LDS: LDS and scalar loads
    // share one completion counter, so waiting for an LDS word would also wait for the prefetched scalar loads).
    // mode 0: one atan2 per (sample, node) pair.
    int mode = 0;
    if constexpr (S::NW > 0) {
        bool same = true, fixed = nv.werr != nullptr;
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) {
            same = same && (gtrig[j] == __shfl(gtrig[j], 0));
            fixed = fixed && (gtrig[j] == nv.wtrig[j]);
        }
        mode = __all(fixed) ? 2 : (__all(same) ? 1 : 0);
    }

    // cost-to-go matrix about the SAMPLE (planner.py:344-345): a constant of the system, or one matrix per sample
    double Sl[DENSE == S_PERSAMPLE ? S::N * S::N : 1];
    const double* Suse = Sd;
    if constexpr (DENSE == S_PERSAMPLE) {
#pragma unroll
        for (int q = 0; q < S::N * S::N; ++q) Sl[q] = Sd[(size_t)ts * (S::N * S::N) + q];
        Suse = Sl;
    }
    constexpr int QC = DENSE == S_PERSAMPLE ? S_DENSE : DENSE;

    double best = INFINITY;
    int bidx = -1;
    // (one copy of the loop per mode, chosen once per wavefront: the cheap modes must not carry the atan2 in their body)
    auto scan = [&](auto mode_c) {
    constexpr int MODE = S::NW > 0 ? decltype(mode_c)::value : 0;
    constexpr int NT = S::N + (MODE == 0 ? 2 * S::NW : (MODE == 2 ? S::NW : 0));   // doubles fetched per node
    for (int base = i0; base < i1; base += 64) {
        const int cnt = (i1 - base) < 64 ? (i1 - base) : 64;
        // eligibility of the tile's nodes as one wave-uniform 64-bit mask (lane j looks at node base + j)
        bool el = false;
        if (lane < cnt) {
            const long long i = base + lane;
            if constexpr (TRI) el = nv.len[i * nv.sn] > 0.0;
            else if (nv.ignore) {
                const int wi = (int)(i >> 6);
                unsigned long long w = nv.ignore[wi];
                if (PATCH && patched) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int ik = __builtin_amdgcn_readlane(pt_idx, k);
                        const unsigned long long vk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(pt_val >> 32), k) << 32) |
                                                      (unsigned)__builtin_amdgcn_readlane((int)pt_val, k);
                        w = (k < pt_n && ik == wi) ? vk : w;
                    }
                }
                el = ((w >> (i & 63)) & 1ull) == 0;
            } else el = true;
        }
        const unsigned long long m = __ballot(el);
        if (m == 0) continue;
        double werr[S::NW > 0 ? S::NW : 1];
        if constexpr (MODE == 1) {
            const long long i = base + (lane < cnt ? lane : 0);  // coalesced on the SoA tree
#pragma unroll
            for (int k = 0; k < S::NW; ++k)
                werr[k] = wrap_err_c(gtrig[2 * k], gtrig[2 * k + 1], nv.trig[i * nv.tn + (2 * k) * nv.td],
                                   nv.trig[i * nv.tn + (2 * k + 1) * nv.td]);
        }
        // Nodes are fetched four at a time: an aligned quad of node slots is one 32-byte scalar load per component on
        // the SoA tree (the in-wave records are AoS and take four 8-byte loads).  The load latency is hidden by the other
        // wavefronts of the SIMD -- the launch is cut into enough workgroups for several of them -- rather than by
        // software pipelining inside this one: a second quad in flight needs more SGPRs than the wavefront has, and
        // scalar-ALU instructions share its issue bandwidth with the fp64 ones, so the loop keeps them to a handful per
        // node (no mask tests at all when the whole tile is eligible).
        struct Quad { double v[4][NT + 1]; };
        auto fetch = [&](int j0, Quad& q) {                      // slots j0 .. j0 + 3 of the tile
            const long long i = base + j0;
            if constexpr (!TRI) {
                // SoA, node index fastest: the quad is contiguous (reading up to three slots past the chunk is harmless:
                // the tables are padded to a multiple of 64 nodes and those slots are never visited)
                auto quad = [&](const double* p, int c) {
                    const double4 w = *reinterpret_cast<const double4*>(p);
                    q.v[0][c] = w.x; q.v[1][c] = w.y; q.v[2][c] = w.z; q.v[3][c] = w.w;
                };
#pragma unroll
                for (int d = 0; d < S::N; ++d) quad(nv.x + i + d * nv.sd, d);
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 2 * S::NW; ++k) quad(nv.trig + i + k * nv.td, S::N + k);
                } else if constexpr (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < S::NW; ++k) quad(nv.werr + i + k * nv.wk, S::N + k);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int d = 0; d < S::N; ++d) q.v[r][d] = nv.x[(i + r) * nv.sn + d * nv.sd];
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int k = 0; k < 2 * S::NW; ++k) q.v[r][S::N + k] = nv.trig[(i + r) * nv.tn + k * nv.td];
                    }
                }
            }
        };
        auto visit = [&](const double* nd, int jj) {            // one (sample, node) pair per lane
            double e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) e[d] = xg[d] - nd[d];
#pragma unroll
            for (int k = 0; k < S::NW; ++k) {
                if constexpr (MODE == 2) e[S::wd(k)] = nd[S::N + k];
                else if constexpr (MODE == 1)
                    e[S::wd(k)] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(werr[k]), jj),
                                                   __builtin_amdgcn_readlane(__double2loint(werr[k]), jj));
                else e[S::wd(k)] = wrap_err_c(gtrig[2 * k], gtrig[2 * k + 1], nd[S::N + 2 * k], nd[S::N + 2 * k + 1]);
            }
            const double c = quad_cost<S, QC>(e, Suse);
            const int i = base + jj;
            const bool ok = (TRI ? (i < t) : true) && c < best;   // strict: the older node keeps an exactly equal cost
            bidx = ok ? i : bidx;
            best = ok ? c : best;
        };
        Quad Q;
        if (m == (cnt == 64 ? ~0ull : (1ull << cnt) - 1ull) && (cnt & 3) == 0) {
#pragma unroll 1
            for (int j0 = 0; j0 < cnt; j0 += 4) {                // every node of the tile eligible: no mask tests
                fetch(j0, Q);
#pragma unroll
                for (int r = 0; r < 4; ++r) visit(Q.v[r], j0 + r);
            }
            continue;
        }
        unsigned long long qm = (m | (m >> 1) | (m >> 2) | (m >> 3)) & 0x1111111111111111ull;   // bit 4g: quad g has an eligible node
        while (qm) {
            const int j0 = __builtin_ctzll(qm);
            qm &= qm - 1;
            fetch(j0, Q);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((m >> (j0 + r)) & 1ull) visit(Q.v[r], j0 + r);
        }
    }
    };
    if (mode == 2) scan(std::integral_constant<int, 2>{});
    else if (mode == 1) scan(std::integral_constant<int, 1>{});
    else scan(std::integral_constant<int, 0>{});
    if constexpr (WPB > 1) {
        __shared__ double rc[WPB][64];
        __shared__ int ri[WPB][64];
        const int wv = threadIdx.x >> 6;
        rc[wv][lane] = best; ri[wv][lane] = bidx;
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int w = 1; w < WPB; ++w) {                          // ascending chunks: strict '<' keeps the lowest id among equal costs
            const double oc = rc[w][lane];
            const int oi = ri[w][lane];
            if (oi >= 0 && (bidx < 0 || oc < best)) { best = oc; bidx = oi; }
        }
    }
    if (t < W) {
        const size_t o = (size_t)by * ps_c + (size_t)t * ps_t;
        // Tree scan: ONE 16-byte store per partial (Part).  In-wave scan (TRI; consumed by k_decide, chunk-major): cost and id in two
        // arrays behind two __restrict__ pointers as before -- with a single pointer the compiler's alias analysis changes the whole
        // kernel's register allocation: 86 -> 126 VGPRs, 5 -> 4 wavefronts per SIMD for the boats (tests/test_abi_cpu.py pins it).
        if constexpr (TRI) { reinterpret_cast<double*>(part)[o] = best; tri_idx[o] = bidx; }
        else { *reinterpret_cast<int4*>(part + o) = make_int4(__double2loint(best), __double2hiint(best), bidx, 0); }
    }
}

// (waves_per_eu: the plain tree scan of a four-state system sits at 79-81 VGPRs, on the edge between 6 and 5 wavefronts per SIMD; it is
//  told to stay at 6 -- the scan's speed is its occupancy; costs it nothing: no scratch -- while every other instantiation keeps the
//  allocation the compiler finds)
template <class S, int DENSE, bool TRI, bool PATCH = false, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(S::N <= 4 && WPB == 1 && !PATCH && !TRI && DENSE == S_IDENT ? 6 : 1))) void k_nn_scan(NodeView nv, const double* __restrict__ xs, const double* __restrict__ xtrig,
                                                int W, const double* __restrict__ Sd, int chunk,
                                                Part* __restrict__ part, int* __restrict__ tri_idx,
                                                int ps_c, int ps_t, IgnPatch pt) {
    nn_scan_body<S, DENSE, TRI, PATCH, WPB>(nv, xs, xtrig, W, Sd, chunk, part, tri_idx, ps_c, ps_t, pt, pt.n,
                                            (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y);
}

// cos/sin of the angular coordinates of a batch of samples, [B][2*NW]: computed once per sample batch so that
// neither the scan nor the steer pays a sincos per (sample, launch)
template <class S>
__global__ void k_sample_trig(const double* __restrict__ xs, int B, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if constexpr (S::NW > 0) {
        double x[S::N], tr[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) x[d] = xs[(size_t)b * S::N + d];
        trig_of<S>(x, tr);
#pragma unroll
        for (int j = 0; j < 2 * S::NW; ++j) out[(size_t)b * (2 * S::NW) + j] = tr[j];
    }
}

// Lexicographic (cost, id) minimum over the chunk partials: one wavefront per sample, lanes span
// the chunks, then a butterfly over the 64 lanes.  Ordering by (cost, node id) keeps the lowest
// node id among exactly equal costs (stable-argsort order, planner.py:240; chunks are ascending in
// id, so comparing ids is the same as comparing chunk order).  When every node is ignored the
// overall best is returned (planner.py:241,245 fallback).
__device__ __forceinline__ void lexmin_wave(double& c, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(c, off);
        const int oi = __shfl_xor(i, off);
        const bool take = (oi >= 0) && (i < 0 || oc < c || (oc == c && oi < i));
        if (take) { c = oc; i = oi; }
    }
}

template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_nn_reduce(const Part* __restrict__ part,
                                                  int W, int n_chunks, NodeView nv, const double* __restrict__ xs,
                                                  const double* __restrict__ Sd, long long s_stride,
                                                  int* __restrict__ out_id, double* __restrict__ out_cost,
                                                  double* __restrict__ rec, int R, int off_cost, int off_parent,
                                                  int* __restrict__ par_done, unsigned char* __restrict__ changed,
                                                  unsigned char* __restrict__ stale) {
    const int t = blockIdx.x;
    if (t >= W) return;
    const int lane = threadIdx.x;
    double b = INFINITY;
    int bi = -1;
    const Part* pp = part + (size_t)t * n_chunks;             // sample-major partials: coalesced
    for (int c = lane; c < n_chunks; c += 64) {               // ascending per lane, strict '<'
        const Part pm = pp[c];
        const double v = pm.c;
        const int vi = pm.i;
        if (vi >= 0 && (bi < 0 || v < b)) { b = v; bi = vi; }
    }
    lexmin_wave(b, bi);
    // Every node ignored (planner.py:241,245): the reference falls back to the overall nearest.  Rare (a tree
    // that is nothing but goal paths), so it is not worth a second set of partials in the scan: this
    // wavefront rescans the table without the mask, lanes striding over the nodes.
    const bool fallback = bi < 0 && nv.ignore != nullptr;
    if (fallback) {
        double xg[S::N], gtrig[2 * S::NW + 1];
#pragma unroll
        for (int d = 0; d < S::N; ++d) xg[d] = xs[(size_t)t * S::N + d];
        trig_of<S>(xg, gtrig);
        for (int i = lane; i < nv.count; i += 64) {
            double x[S::N], trig[2 * S::NW + 1], e[S::N];
#pragma unroll
            for (int d = 0; d < S::N; ++d) x[d] = nv.x[(long long)i * nv.sn + d * nv.sd];
#pragma unroll
            for (int j = 0; j < 2 * S::NW; ++j) trig[j] = nv.trig[(long long)i * nv.tn + j * nv.td];
            erf_cached<S>(xg, gtrig, x, trig, e);
            const double c = quad_cost<S, DENSE>(e, Sd + (size_t)t * s_stride);
            if (bi < 0 || c < b) { b = c; bi = i; }
        }
        lexmin_wave(b, bi);
    }
    if (lane == 0) {
        if (out_id) out_id[t] = bi;
        if (out_cost) out_cost[t] = b;
        if (rec) {
            // A fallback parent only stands if nothing else exists: any (never ignored) node born earlier
            // in the same wave must beat it regardless of cost, so the record carries +inf as its cost.
            rec[(size_t)t * R + off_cost] = fallback ? INFINITY : b;
            rec[(size_t)t * R + off_parent] = (double)bi;
        }
        if (par_done) { par_done[t] = bi; changed[t] = 0; stale[t] = 0; }   // wave bookkeeping starts here
    }
}

// Tree-sharded waves: (cost, id) candidate of every sample from one rank's node range, as W pairs of doubles (the
// all-gather payload), and back into the partial-minima layout the steer prologue reduces ([sample][part]).
__global__ void k_best_pack(const double* __restrict__ cost, const int* __restrict__ id, int W, double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < W) { out[2 * t] = cost[t]; out[2 * t + 1] = (double)id[t]; }
}
__global__ void k_best_unpack(const double* __restrict__ in, int W, int parts, Part* __restrict__ part) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= W * parts) return;
    const int t = q / parts, p = q - t * parts;
    const double* src = in + ((size_t)p * W + t) * 2;
    Part pm;
    pm.c = src[0]; pm.i = (int)src[1]; pm.pad = 0;
    part[q] = pm;
}

// Full cost vector of one sample (planner.py:340-350); thread per node.
template <class S, int DENSE>
__global__ void k_costs(NodeView nv, const double* __restrict__ xq, const double* __restrict__ Sd,
                        double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv.count) return;
    double xg[S::N], gtrig[2 * S::NW + 1], x[S::N], trig[2 * S::NW + 1], e[S::N];
#pragma unroll
    for (int d = 0; d < S::N; ++d) { xg[d] = xq[d]; x[d] = nv.x[(long long)i * nv.sn + d * nv.sd]; }
    trig_of<S>(xg, gtrig);
#pragma unroll
    for (int j = 0; j < 2 * S::NW; ++j) trig[j] = nv.trig[(long long)i * nv.tn + j * nv.td];
    erf_cached<S>(xg, gtrig, x, trig, e);
    out[i] = quad_cost<S, DENSE>(e, Sd);
}

