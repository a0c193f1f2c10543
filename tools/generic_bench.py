#!/usr/bin/env python
"""Callback mode's nearest-neighbour stage (csrc/generic.hpp) as a function of the tree size (GPU box).

For N nodes of n states (nw of them angular): (a) the latency of the synchronous host-form query lqrrt_nn_argmin_host -- what one
iteration of the callback planner pays -- on the host clock; (b) the duration of the scan + reduce launches of the device form
(lqrrt_nn_argmin, W = 1) between two events on the launch stream, and the bandwidth that is of one pass over the table:
N * (8 n + 16 nw) bytes + N / 8 ignore bytes -- the HBM-shaped scan of SURVEY 8(d).  Small tables are launch-latency-bound; the
large ones show what the kernel does against the 8 TB/s roof.
  python tools/generic_bench.py [--n 6] [--angles 2] [--sizes 1000,10000,100000,1000000,4000000,16000000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=6)
    ap.add_argument("--angles", default="2")
    ap.add_argument("--sizes", default="1000,10000,100000,1000000,4000000,16000000")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    import torch
    from lqrrt_amd import _native as nat
    from lqrrt_amd.engine import NodeTable
    angles = tuple(int(v) for v in a.angles.split(",")) if a.angles else ()
    rng = np.random.RandomState(3)
    for N in [int(v) for v in a.sizes.split(",")]:
        t = NodeTable(a.n, 1, angles, capacity=N + 64)
        nodes = rng.uniform(-5, 5, (N, a.n))
        pid = np.maximum(np.arange(N) - 1, -1).astype(np.int32)
        t.load(nodes, pid)
        q = rng.uniform(-5, 5, a.n)
        for dense in (False, True):
            S = None
            if dense:
                A = rng.uniform(-1, 1, (a.n, a.n))
                S = A.dot(A.T) + a.n * np.eye(a.n)
            got = t.nearest(q, S)[0]
            t0 = time.perf_counter()
            for _ in range(200):
                t.nearest(q, S)
            host_us = 1e6 * (time.perf_counter() - t0) / 200
            dq = torch.from_numpy(q.reshape(1, -1).copy()).cuda()
            dS = torch.from_numpy(np.ascontiguousarray(S)).cuda() if dense else None
            ids = torch.empty(1, dtype=torch.int32, device="cuda")
            cost = torch.empty(1, dtype=torch.float64, device="cuda")
            st = nat.current_stream(0)
            call = lambda: nat.check(nat.lib().lqrrt_nn_argmin(t.h, dq.data_ptr(), 1, dS.data_ptr() if dense else None, 1,
                                                               ids.data_ptr(), cost.data_ptr(), st))
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(a.reps):
                call()
            ev1.record()
            torch.cuda.synchronize()
            dev_us = 1e3 * ev0.elapsed_time(ev1) / a.reps
            want = None                                              # (checked AFTER the timing: seconds of NumPy leave the GPU idle)
            if N <= 1000000:
                e = q - nodes
                for d in angles:
                    e[:, d] = np.arctan2(np.sin(e[:, d]), np.cos(e[:, d]))
                c = np.sum(e.dot(np.eye(a.n) if S is None else S) * e, axis=1)
                want = int(np.argmin(c))
            nbytes = N * (8 * a.n + 16 * len(angles)) + N / 8
            print(json.dumps(dict(N=N, n=a.n, angular=len(angles), S="dense" if dense else "identity", host_query_us=round(host_us, 2),
                                  device_scan_plus_reduce_us=round(dev_us, 2), table_bytes=int(nbytes), GBps=round(nbytes / dev_us / 1e3, 1),
                                  frac_of_8TBps=round(nbytes / dev_us / 1e3 / 8000.0, 4), agrees_with_numpy=(None if want is None else bool(got == want)))))
        t.close()


if __name__ == "__main__":
    main()
