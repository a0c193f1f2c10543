#!/usr/bin/env python
"""Randomised engine-vs-sequential-oracle comparison (bit for bit) over systems, seeds, wave sizes, sampler
tries, pruning and horizon modes.

    python tools/fuzz_parity.py [cases] [seed] [only_case]      (FUZZ_WAVE=n overrides the wave size)

Found so far: a wave cut at a goal hit whose hit later vanished left samples beyond the old cut with records
computed from an in-wave parent's previous end state (k_decide now remembers them as stale)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

NAMES = ["boat_advanced", "boat_intermediate", "boat_novice", "car", "double_integrator", "ros_boat", "pendulum"]
# FUZZ_USER=<oracle build of the user header>: fuzz the out-of-tree example instead (examples/user_system/unicycle.hpp; the engine
# with it compiled in comes through LQRRT_LIB, the oracle's callbacks through tools/build_user_system.py --oracle)
USER_ORACLE = os.environ.get("FUZZ_USER")
if USER_ORACLE:
    NAMES = ["user"]
    sys.path.insert(0, os.path.join(ROOT, "examples", "user_system"))


def draw_case(rng, rng2, names=None, rng3=None):
    """One random configuration; consumes a fixed pattern of draws so that case k is reproducible (rng2 is a second
    stream for dimensions added later, so that earlier case numbers keep their meaning)."""
    import lqrrt_amd
    names = names or NAMES
    name = names[rng.randint(len(names))]
    if name == "user":
        import coracle, plan_unicycle
        coracle.use_user_model(USER_ORACLE)
        s = plan_unicycle.make_system(int(rng.randint(4)))
    elif name == "double_integrator":
        s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=int(rng.choice([50, 2000, 20000])), seed=int(rng.randint(5)))
    elif name == "ros_boat":
        s = lqrrt_amd.systems.RosBoat(str(rng.choice(["boat", "car", "escape"])), focus=[12.0, -3.0] if rng.rand() < 0.3 else None)
    else:
        s = lqrrt_amd.systems.SYSTEMS[name](int(rng.randint(4)))
    c = dict(name=name, system=s)
    c["nodes"] = int(rng.choice([40, 150, 400, 900]))
    c["wave"] = int(rng.choice([1, 2, 3, 7, 16, 64, 100, 256, 512, 1024]))
    c["seed"] = int(rng.randint(10 ** 6))
    c["tries"] = int(rng.choice([1, 2, 10]))
    c["pruning"] = bool(rng.rand() < 0.7)
    c["stop_goal"] = bool(rng.rand() < 0.2)
    kw = s.plan_kwargs
    c["adaptive"] = bool(hasattr(kw["horizon"], "__len__") or (rng.rand() < 0.25))
    c["horizon"] = kw["horizon"] if hasattr(kw["horizon"], "__len__") else ((0.1, 1.5) if c["adaptive"] else kw["horizon"])
    c["world"] = int(rng2.choice([1, 1, 1, 2, 3, 8]))      # >1: sample-sharded waves, ranks emulated on one GPU
    if c["world"] > 1:
        c["stop_goal"] = False
    c["sync"] = bool(rng2.rand() < 0.25)                 # synchronous wave mode vs orc_extend_sync
    c["ogrid"] = False
    if name in ("boat_advanced", "boat_intermediate", "ros_boat") and rng2.rand() < 0.35:
        # a synthetic occupancy map over the sample space (lqrrt_node.py:792-860 feasibility model)
        space = np.array(s.sample_space, dtype=np.float64)
        cpm = float(rng2.choice([1.0, 2.5, 5.0]))
        origin = (space[0, 0] - 8.0, space[1, 0] - 8.0)
        cols = int((space[0, 1] - space[0, 0] + 16.0) * cpm)
        rows = int((space[1, 1] - space[1, 0] + 16.0) * cpm)
        grid = rng2.randint(0, 60, size=(rows, cols)).astype(np.int8)
        for _ in range(int(rng2.randint(3, 25))):
            r0, c0 = int(rng2.randint(rows)), int(rng2.randint(cols))
            grid[r0:r0 + int(rng2.randint(1, 6 * cpm + 2)), c0:c0 + int(rng2.randint(1, 6 * cpm + 2))] = int(rng2.randint(80, 101))
        for px, py in ((s.x0[0], s.x0[1]), (s.goal[0], s.goal[1])):
            cc, rr = int(cpm * (px - origin[0])), int(cpm * (py - origin[1]))
            k = int(7 * cpm)
            grid[max(rr - k, 0):rr + k, max(cc - k, 0):cc + k] = 0
        s.set_occupancy_grid(grid, origin, cpm=cpm, threshold=float(rng2.choice([50.0, 90.0])))
        c["ogrid"] = True
    # round-2 dimensions on a third stream: the Riccati pendulum (per-step DARE gains, S per sample) in a share of the
    # cases, and tree-sharded instead of sample-sharded waves for half of the multi-rank ones
    c["shard"] = "sample"
    if rng3 is not None:
        if rng3.rand() < 0.08:
            c["name"], c["system"] = "pendulum_lqr", lqrrt_amd.systems.PendulumLqr(0)
            c["nodes"] = min(c["nodes"], 150)
            c["adaptive"], c["horizon"], c["ogrid"] = False, c["system"].plan_kwargs["horizon"], False
        if rng3.rand() < 0.5 and c["world"] > 1 and not c["sync"]:
            c["shard"] = "tree"
    return c


def run_case(c, verbose=False):
    """Grow the same tree with the HIP engine (waves) and the sequential C oracle; True when bit-identical."""
    import coracle
    from lqrrt_amd.engine import Engine
    s, nodes, wave, seed, tries = c["system"], c["nodes"], c["wave"], c["seed"], c["tries"]
    kw = s.plan_kwargs
    budget = 30 * nodes
    def make_engine():
        eng = Engine(s, capacity=nodes + wave + 8, max_wave=wave)
        if c["adaptive"]:
            hspan = np.divide(c["horizon"], kw["dt"]).astype(np.int64)
            eng.set_resolution(kw["dt"], kw["FPR"], int(hspan[1]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer),
                               adaptive=True, hspan_min=int(hspan[0]), horizon_iters_state=1)
        else:
            eng.set_resolution(kw["dt"], kw["FPR"], int(c["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
        space = np.array(s.sample_space, dtype=np.float64)
        eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), tries)
        st = np.random.RandomState(seed).get_state()
        eng.set_mt19937(st[1], st[2])
        eng.tree_reset(s.x0)
        if c.get("sync"):
            eng.set_wave_mode("synchronous")
        return eng

    world = c.get("world", 1)
    others = []
    if world == 1:
        eng = make_engine()
        stats = eng.extend(wave, max_attempts=budget, node_limit=nodes, pruning=c["pruning"], stop_on_goal=c["stop_goal"])
    else:
        import torch
        from lqrrt_amd.parallel import node_range, records_tensor, shard_bounds
        ranks = [make_engine() for _ in range(world)]
        recs = [records_tensor(e) for e in ranks]
        bufs = [torch.empty((world, wave, 2), dtype=torch.float64, device="cuda") for _ in range(world)]
        attempts = 0
        while ranks[0].size <= nodes and attempts < budget and c.get("shard") == "tree":
            # tree-sharded: every rank scans its node range for all samples, candidates exchanged, everybody steers
            W = min(ranks[0].wave_suggest(wave), budget - attempts)
            views = [b.view(-1)[: world * W * 2].view(world, W, 2) for b in bufs]
            for r, e in enumerate(ranks):
                lo, hi = node_range(e.size, r, world)
                e.wave_scan_nodes(W, lo, hi, views[r][r].data_ptr())
            torch.cuda.synchronize()
            for r in range(world):
                for q in range(world):
                    if q != r:
                        views[q][r].copy_(views[r][r])
            torch.cuda.synchronize()
            for r, e in enumerate(ranks):
                e.wave_steer_candidates(W, world, views[r].data_ptr())
            sts = [e.wave_commit(W, budget - attempts, nodes, c["pruning"]) for e in ranks]
            attempts += sts[0].attempts
        while ranks[0].size <= nodes and attempts < budget:
            W = min(wave, budget - attempts) if c.get("sync") else ranks[0].wave_suggest(wave)
            bounds = [shard_bounds(W, r, world) for r in range(world)]
            for r, e in enumerate(ranks):
                e.wave_speculate(W, bounds[r][1], bounds[r][2])
            torch.cuda.synchronize()
            for r in range(world):
                lo, hi = bounds[r][1], bounds[r][2]
                for q in range(world):
                    if q != r and hi > lo:
                        recs[q][lo:hi].copy_(recs[r][lo:hi])
            torch.cuda.synchronize()
            sts = [e.wave_commit(W, budget - attempts, nodes, c["pruning"]) for e in ranks]
            attempts += sts[0].attempts
        eng, others = ranks[0], ranks[1:]
        stats = eng.counters()
    o = coracle.make(s, nodes + wave + 8, seed=seed, tries=tries, horizon=c["horizon"])
    if c.get("sync"):
        o.extend_sync(wave, max_iters=budget, max_nodes=nodes, pruning=c["pruning"], stop_on_goal=c["stop_goal"])
    else:
        o.extend(max_iters=budget, max_nodes=nodes, pruning=c["pruning"], stop_on_goal=c["stop_goal"])
    ok = (eng.size == o.size and stats.attempts == o.iterations and stats.candidates == o.candidates
          and np.array_equal(eng.parents(), o.parents()) and np.array_equal(eng.states(), o.states())
          and np.array_equal(eng.edge_lengths(), o.edge_lengths()) and np.array_equal(eng.ignored(), o.ignored())
          and eng.plan_best()[0] == o.best()[0] and (not c["adaptive"] or eng.horizon_iters_state() == o.horizon_iters))
    if verbose:
        n = min(eng.size, o.size)
        pe, po, se, so = eng.parents()[:n], o.parents()[:n], eng.states()[:n], o.states()[:n]
        d = np.nonzero((pe != po) | np.any(se != so, axis=1))[0]
        print("sizes", eng.size, o.size, "attempts", stats.attempts, o.iterations, "first differing nodes:", d[:5])
        if len(d):
            i = d[0]
            print("engine parent", pe[i], "state", se[i]); print("oracle parent", po[i], "state", so[i])
            print("state diff", se[i] - so[i], "edge len", eng.edge_lengths()[i], o.edge_lengths()[i])
            xe, ue = eng.edge(int(i)); xo, uo = o.edge(int(i))
            m = min(len(xe), len(xo))
            bad_steps = np.nonzero(np.any(xe[:m] != xo[:m], axis=1) | np.any(ue[:m] != uo[:m], axis=1))[0]
            print("first differing edge step", bad_steps[:3], "of", m)
            if len(bad_steps):
                j = bad_steps[0]
                print(" x eng", [float.hex(float(v)) for v in xe[j]]); print(" x orc", [float.hex(float(v)) for v in xo[j]])
                print(" u eng", [float.hex(float(v)) for v in ue[j]]); print(" u orc", [float.hex(float(v)) for v in uo[j]])
                if j > 0:
                    print(" prev x", [float.hex(float(v)) for v in xe[j - 1]], "same", np.array_equal(xe[j - 1], xo[j - 1]))
                print(" parent state", [float.hex(float(v)) for v in se[pe[i]]], "K equal", np.array_equal(eng.gains()[pe[i]], o.gains()[pe[i]]))
    for e in others:
        ok = ok and e.size == eng.size and np.array_equal(e.parents(), eng.parents()) and np.array_equal(e.states(), eng.states())
        e.close()
    eng.close()
    return ok


def describe(c):
    return " ".join("%s=%s" % (k, c[k]) for k in ("name", "nodes", "wave", "seed", "tries", "pruning", "stop_goal", "adaptive", "world", "shard", "ogrid", "sync")) \
        + (" behavior=%s" % c["system"].behavior if hasattr(c["system"], "behavior") else "")


def run(cases, seed, only=-1, wave_override=None, names=None):
    rng, rng2, rng3 = np.random.RandomState(seed), np.random.RandomState(seed + 7919), np.random.RandomState(seed + 104729)
    bad = []
    for k in range(cases):
        c = draw_case(rng, rng2, names, rng3)
        if os.environ.get("FUZZ_RICCATI"):
            # every case on a Riccati system (per-step doubling DARE inside the rollouts, S about every sample): the pendulum and
            # the 6-state boat in turn; small trees, the sequential oracle solves a Riccati equation per recorded step too
            import lqrrt_amd
            nm = ("pendulum_lqr", "boat_novice_lqr")[k & 1]
            c["name"], c["system"] = nm, lqrrt_amd.systems.SYSTEMS[nm](int(k >> 1) & 3)
            c["nodes"] = min(c["nodes"], 150)
            c["adaptive"], c["horizon"], c["ogrid"] = False, c["system"].plan_kwargs["horizon"], False
        if wave_override:
            c["wave"] = int(wave_override)
        if only >= 0 and k != only:
            continue
        if not run_case(c, verbose=only >= 0):
            bad.append((k, describe(c)))
    return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    t0 = time.time()
    bad = run(cases, seed, only, os.environ.get("FUZZ_WAVE"))
    for k, d in bad:
        print("MISMATCH case", k, d)
    print("cases %d mismatches %d in %.1f s%s" % (cases, len(bad), time.time() - t0, " (user system)" if USER_ORACLE else ""))
    sys.exit(1 if bad else 0)
