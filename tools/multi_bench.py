#!/usr/bin/env python
"""Aggregate throughput of n independent planners on ONE GPU through lqrrt_engine_extend_multi (GPU box).
Every engine: demo_boat_advanced, its own sample seed, grown to the 10k-node window of the headline metric (9,500 .. 10,500 nodes,
rewound like bench.py's loop); then `--steps` calls of `--per-call` attempts per engine, timed.  Prints one JSON line per n.
  python tools/multi_bench.py [--trees 1,2,4,8,16,32] [--steps 6] [--per-call 8192]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(seed, nodes, wave):
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.SYSTEMS["boat_advanced"](0)
    eng = Engine(s, capacity=nodes + 2 * wave + 1024, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(seed).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    return eng


def run(n, steps, per_call, nodes=10000, wave=256, threads=1):
    """threads > 1: the engines are split into that many groups, each advanced by its own extend_multi loop in a host thread of its
    own on a stream of its own (ctypes releases the GIL inside the native call): the groups' launches overlap on the GPU."""
    import threading
    import torch
    from lqrrt_amd.engine import Engine
    lo, hi = int(0.95 * nodes), int(1.05 * nodes)
    engs = [build(1 + k, hi + 64, wave) for k in range(n)]
    t0 = time.perf_counter()
    Engine.extend_multi(engs, wave, until_size=lo)
    torch.cuda.synchronize()
    t_grow = time.perf_counter() - t0
    for e in engs:
        e.tree_mark()
        e.set_rewind_above(hi - wave)                   # every engine keeps its own window inside the native call
    groups = [engs[g::threads] for g in range(threads)]
    streams = [torch.cuda.Stream() for _ in range(threads)]

    def group_step(g):
        grp = groups[g]
        with torch.cuda.stream(streams[g]):
            done = 0
            left = per_call
            while left > 0:
                want = left
                sts = Engine.extend_multi(grp, wave, max_attempts=want)
                done += sum(s.attempts for s in sts)
                left -= want
            streams[g].synchronize()
        return done

    def step():
        if threads == 1:
            return group_step(0)
        res = [0] * threads
        th = [threading.Thread(target=lambda g=g: res.__setitem__(g, group_step(g))) for g in range(threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return sum(res)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0
    for _ in range(steps):
        total += step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = [e.counters() for e in engs]
    out = dict(trees=n, host_threads=threads, attempts_per_s=total / dt, per_tree=total / dt / n, growth_s=t_grow, seconds=dt, attempts=total,
               mean_wave=float(np.mean([x.attempts / max(1, x.waves) for x in c])),
               rounds_per_1024=float(np.mean([1024.0 * x.fix_rounds / max(1, x.attempts) for x in c])))
    for e in engs:
        e.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", default="1,2,4,8,16,32")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--per-call", type=int, default=8192)
    ap.add_argument("--threads", default="1", help="host threads (groups of engines, a stream each), comma list")
    a = ap.parse_args()
    for n in [int(x) for x in a.trees.split(",")]:
        for t in [int(x) for x in a.threads.split(",")]:
            if t <= n:
                print(json.dumps(run(n, a.steps, a.per_call, threads=t)), flush=True)


if __name__ == "__main__":
    main()
