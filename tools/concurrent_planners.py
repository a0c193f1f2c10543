#!/usr/bin/env python
"""One tree is latency-bound (a few hundred wavefronts in flight on a 1024-SIMD device).  How far does ONE MI355X go with
several independent planners at once?  Each thread owns an engine and a HIP stream (ctypes releases the GIL inside
the native calls; engine handles share no state) and runs bench.py's windowed loop on its own tree.

    python tools/concurrent_planners.py [n_planners ...]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def worker(idx, seed, steps, barrier, out):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        boat, eng = bench.build_problem("cfg4", 10000, 1024, 0, seed=seed)
        eng.extend(1024, until_size=9500)
        eng.tree_mark()
        barrier.wait()
        t0 = time.perf_counter()
        done = 0
        for _ in range(steps):
            if eng.size > 10100:
                eng.tree_rewind()
            done += eng.extend(1024, max_attempts=1024).attempts
        stream.synchronize()
        out[idx] = (done, time.perf_counter() - t0)


def run(n, steps=150):
    out = [None] * n
    barrier = threading.Barrier(n)
    th = [threading.Thread(target=worker, args=(i, 1 + i, steps, barrier, out)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    total = sum(o[0] for o in out)
    span = max(o[1] for o in out)
    print("planners %2d: aggregate %9.0f attempts/s  (per planner %8.0f, %d attempts each, wall %.2f s incl. tree growth)" % (
        n, total / span, total / span / n, out[0][0], time.perf_counter() - t0))


if __name__ == "__main__":
    for n in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]):
        run(n)
