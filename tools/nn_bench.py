#!/usr/bin/env python
"""NN-scan microbenchmark (GPU box): the demo_boat_advanced tree grown to --nodes, then lqrrt_nn_argmin over W
samples for several W, the scan kernel timed with the HIP events attached to its dispatch (lqrrt_profile_*).
Prints one JSON line per W: launch time, sample-node pairs/s, algorithmic GB/s (W*N*(8n+1) bytes), fp64 VALU rate
(16 flop per pair: 5 sub + 6 mul + 5 add, identity S) against the non-FMA vector peak.
  python tools/nn_bench.py [--nodes 10000] [--system boat_advanced|double_integrator] [--reps 200]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--system", default="boat_advanced")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--waves", default="64,128,256,512,1024")
    a = ap.parse_args()
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    if a.system == "double_integrator":
        s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=100000, seed=0)
    else:
        s = lqrrt_amd.systems.SYSTEMS[a.system](0)
    eng = Engine(s, capacity=a.nodes + 2048, max_wave=1024)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(1).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    eng.extend(1024, until_size=a.nodes)
    N, n = eng.size, s.nstates
    rng = np.random.RandomState(5)
    import torch
    for W in [int(w) for w in a.waves.split(",")]:
        q = space[:, 0] + (space[:, 1] - space[:, 0]) * rng.random_sample((W, n))
        dq = torch.from_numpy(np.ascontiguousarray(q)).cuda()
        ids = torch.empty(W, dtype=torch.int32, device="cuda")
        cost = torch.empty(W, dtype=torch.float64, device="cuda")
        from lqrrt_amd import _native as nat
        call = lambda: nat.check(nat.lib().lqrrt_nn_argmin(eng.h, dq.data_ptr(), W, None, 1, ids.data_ptr(), cost.data_ptr(), eng._stream()))
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        eng.profile_enable(True, steer=False)
        for _ in range(a.reps):
            call()
        torch.cuda.synchronize()
        p = eng.profile_read()
        eng.profile_enable(False)
        us = 1e3 * p["nn_ms"] / p["nn_launches"]
        pairs = W * N
        flop = 16.0 if a.system != "double_integrator" else 0.0
        print(json.dumps(dict(system=a.system, W=W, N=N, launch_us=round(us, 2), pairs_per_s=pairs / (us * 1e-6),
                              algorithmic_GBps=pairs * (8 * n + 1) / (us * 1e-6) / 1e9,
                              fp64_TFLOPs=pairs * flop / (us * 1e-6) / 1e12)))


if __name__ == "__main__":
    main()
