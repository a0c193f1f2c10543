// multi.hpp -- launches whose grid spans several engines (lqrrt_engine_extend_multi).
// Fragment of kernels.hpp (included there, in order, inside namespace lq).
#pragma once

// ------------------------------------------------------------------------------------------
// Launches whose grid spans SEVERAL engines (lqrrt_engine_extend_multi, round 5).  One planner is a chain of dependent launches
// that leaves ~98 % of the chip idle; n independent planners advance in lock step, one launch of each kind per "tick": the scans
// of the engines that begin a wave in ONE k_nn_scan_multi launch, and every engine's steer launch of that tick -- a speculative
// launch, a fused repair round or its append -- in ONE k_steer_multi launch.  A workgroup finds its engine from the prefix
// table of workgroup counts in the kernel arguments, takes what does not change between launches (buffers, model constants,
// geometry, resolution) from that engine's device-resident EngineProto and what does (tree size, wave size, sample window,
// round number ...) from the arguments, and runs the same body as the one-engine kernels: the trees are bit-identical to those
// the engines grow one by one (tests/test_multi_gpu.py).
constexpr int MULTI_MAX = 32;            // engines per launch
constexpr int MULTI_PATCHES = 4;         // goal hits per tick whose ignore words ride in the arguments (the others are uploaded)
struct EngineProto { Params P; Geo g; Res r; TreeView tv; double* rec; RecLayout L; SteerFuse f; RoundArgs ra; };
struct ProtoTable { const EngineProto* p[MULTI_MAX]; };
struct ScanDyn { const double* xs; const double* xtrig; int W, N, chunk, n_chunks, gx, patch; };
struct ScanMultiArgs { int n, pad; int block0[MULTI_MAX + 2]; ScanDyn d[MULTI_MAX]; IgnPatch patch[MULTI_PATCHES]; };
struct SteerDyn { const double* xs; const double* xtrig; long long max_commit, room; int mode, count, n_chunks, N, W, round, base, seq; };
struct SteerMultiArgs { int n, pad; int block0[MULTI_MAX + 2]; SteerDyn d[MULTI_MAX]; };
enum { MULTI_IDLE = 0, MULTI_SPECULATE = 1, MULTI_ROUND = 2 };

// engine of workgroup `blk`: block0 is ascending, block0[n] the grid size; a handful of scalar compares
__device__ __forceinline__ int multi_engine_of(const int* block0, int n, int blk) {
    int e = 0;
    for (int i = 1; i < n; ++i) e = (blk >= block0[i]) ? i : e;
    return e;
}

template <class S, int DENSE>
__global__ __launch_bounds__(64) void k_nn_scan_multi(ProtoTable pt, ScanMultiArgs a) {
    const int e = multi_engine_of(a.block0, a.n, (int)blockIdx.x);
    const ScanDyn& d = a.d[e];
    const int b = (int)blockIdx.x - a.block0[e];
    if (b >= d.gx * d.n_chunks) return;                         // (every engine's range is padded to a multiple of 8 workgroups)
    const EngineProto& p = *pt.p[e];
    NodeView nv = p.f.nv;
    nv.count = d.N;
    const int slot = d.patch;
    nn_scan_body<S, DENSE, false, true, 1>(nv, d.xs, d.xtrig, d.W, p.f.Sd, d.chunk, const_cast<Part*>(p.f.part), nullptr, 1, d.n_chunks,
                                           a.patch[slot < 0 ? 0 : slot], slot < 0 ? 0 : a.patch[slot].n, b, d.gx, d.n_chunks);
}

template <class S, int DENSE, int NWF>
__global__ __launch_bounds__(64 * NWF) void k_steer_multi(ProtoTable pt, SteerMultiArgs a) {
    const int e = multi_engine_of(a.block0, a.n, (int)blockIdx.x);
    const SteerDyn& d = a.d[e];
    const int bid = (int)blockIdx.x - a.block0[e];
    if (bid >= d.count) return;
    const EngineProto& p = *pt.p[e];
    {
        // as k_steer touches its argument block: the prototype's ~2.5 KB are read lazily by scalar loads on the critical path
        const volatile int* ka = (const volatile int*)&p;
        constexpr int LINES = (int)(sizeof(EngineProto) / 64);
        if ((int)(threadIdx.x & 63) < LINES) (void)ka[(threadIdx.x & 63) * 16];
    }
    SteerFuse f = p.f;
    f.W = d.W; f.xtrig = d.xtrig;
    const int* par = nullptr;
    int rd_on = 0;
    if (d.mode == MULTI_SPECULATE) {
        f.n_chunks = d.n_chunks; f.nv.count = d.N;
        par = f.par_out;
    } else {
        f.n_chunks = 0; f.M = nullptr;
        rd_on = 1;
    }
    steer_body<S, DENSE, NWF, false>(p.P, p.g, p.r, p.tv, p.rec, p.L, d.xs, nullptr, 0, par, nullptr, f, p.ra, rd_on, d.round, d.W, d.base, d.seq, d.max_commit, d.room, bid);
}
