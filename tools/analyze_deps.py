"""How many sequential rollout slots does exact mode NEED?  (design study for the round-3 scheduler, CPU only)

Runs the sequential C oracle on the headline workload (demo_boat_advanced, 9.5k-node tree), records every attempt
(sample, nearest node, edge length, goal hit) and derives, for every attempt t, what it truly depends on:
  * the attempt that created its parent node, if that is recent (an in-wave conflict), and
  * every recent goal hit whose ignored path changed its choice (planner.py:239-247,270).
From that dependency DAG it prices three schedules in "slots" (one slot = one steer launch = one full rollout):
  A. waves cut at the first goal hit (round 2's loop): 1 speculative slot + max dependency depth per wave
  B. waves of W samples that are NOT cut at goal hits (a hit only invalidates the samples it affects)
  C. a sliding window of W samples: every launch is a round for all samples in flight; the settled prefix commits,
     new samples enter behind it.
Spurious re-steers (a sample that follows a record which later changes) are not modelled: these are lower bounds, and
schedule A's bound is printed next to the measured 28.1 repair rounds per 1024 attempts to show how tight it is.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import coracle
import lqrrt_amd
SYSTEMS = lqrrt_amd.systems.SYSTEMS


def main():
    grow_to = int(sys.argv[1]) if len(sys.argv) > 1 else 9500
    n_att = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    sysd = SYSTEMS["boat_advanced"](0)
    o = coracle.make(sysd, grow_to + n_att + 64, seed=1)
    o.enable_trace(400000)
    t0 = time.time()
    o.extend(max_nodes=grow_to - 1)
    k0 = o.iterations
    o.extend(max_iters=n_att)
    k1 = o.iterations
    print("grown to %d nodes in %d attempts, then %d more attempts -> %d nodes (%.1f s)" % (grow_to, k0, k1 - k0, o.size, time.time() - t0))
    near, ln = o.trace()
    xs = o.trace_samples()
    states = o.states()
    pid = o.parents()
    goal, buf = np.asarray(sysd.goal, float), np.abs(np.asarray(sysd.goal_buffer, float))
    lo, hi = goal - buf, goal + buf
    # node id created by attempt t (or -1), creator attempt of every node
    acc = ln > 0
    node_of = np.where(acc, np.cumsum(acc), -1)          # root is node 0, first accepted attempt creates node 1
    creator = np.full(o.size, -1, dtype=np.int64)
    creator[node_of[acc]] = np.nonzero(acc)[0]
    in_goal = np.all((lo < states) & (states < hi), axis=1)
    is_hit = np.zeros(k1, bool)
    is_hit[acc] = in_goal[node_of[acc]]
    # attempt at which every node became ignored (inf = never)
    ign_time = np.full(o.size, np.inf)
    for t in np.nonzero(is_hit)[0]:
        v = node_of[t]
        while v != -1:
            if ign_time[v] > t:
                ign_time[v] = t
            v = pid[v]
    assert np.array_equal(np.isfinite(ign_time), o.ignored()), "ignore reconstruction differs from the oracle"
    n_before = np.concatenate(([1], 1 + np.cumsum(acc)))[:k1]      # tree size seen by attempt t

    def nn(t, n_nodes, mask_ign):
        e = xs[t] - states[:n_nodes]
        d = xs[t, 2] - states[:n_nodes, 2]
        e[:, 2] = np.arctan2(np.sin(d), np.cos(d))
        c = np.einsum("ij,ij->i", e, e)
        c[mask_ign[:n_nodes]] = np.inf
        return int(np.argmin(c))

    # sanity: the reconstruction reproduces the oracle's decisions
    bad = 0
    for t in range(k0, k0 + 300):
        if nn(t, n_before[t], ign_time < t) != near[t]:
            bad += 1
    print("reconstructed nearest differs from the oracle in %d of 300 attempts (ties / libm)" % bad)

    WINDOW = 1024                                        # dependencies are collected w.r.t. a snapshot this far back
    T = np.arange(k0, k1)
    dep_parent = {}                                      # t -> creator attempt of its parent (any age)
    dep_hits = {}                                        # t -> list of hit attempts (within WINDOW) that changed its choice
    t1 = time.time()
    for t in T:
        p = near[t]
        dep_parent[t] = creator[p]
        a = max(0, t - WINDOW)
        hits = []
        applied = ign_time < a
        q = nn(t, n_before[t], applied)
        guard = 0
        while q != p and np.isfinite(ign_time[q]) and ign_time[q] < t and guard < 64:
            h = int(ign_time[q])
            hits.append(h)
            applied = applied | (ign_time == h)
            q = nn(t, n_before[t], applied)
            guard += 1
        dep_hits[t] = hits
    print("dependencies of %d attempts in %.1f s; attempts with a hit dependency: %d (%.1f per 1024), goal hits: %d (%.1f per 1024)"
          % (len(T), time.time() - t1, sum(1 for t in T if dep_hits[t]), 1024.0 * sum(1 for t in T if dep_hits[t]) / len(T),
             int(is_hit[k0:k1].sum()), 1024.0 * is_hit[k0:k1].sum() / len(T)))
    print("accepted: %.1f %%; attempts whose parent was created within the last 64 / 256 / 1024 attempts: %.1f / %.1f / %.1f %%"
          % (100.0 * acc[k0:k1].mean(), *[100.0 * np.mean([t - dep_parent[t] <= w and dep_parent[t] >= 0 for t in T]) for w in (64, 256, 1024)]))

    def wave_depths(a, b, cut_hits, hit_lag=1):
        """dependency depth of every attempt of the wave [a, b) against the snapshot at a.  hit_lag: rounds between the round that
        makes a goal hit final and the round in which the samples it affects re-select (1: the device applies it itself; 2: the HOST
        applies it -- it hears of the hit when that round closes, by which time the next round is already enqueued, so the changed
        ignore words ride with the round after)"""
        depth = {}
        for t in range(a, b):
            d = 0
            c = dep_parent[t]
            if c >= a:
                d = max(d, depth[c] + 1)
            if not cut_hits:
                for h in dep_hits[t]:
                    if h >= a:
                        d = max(d, depth[h] + hit_lag)
            depth[t] = d
        return depth

    print("\nslots per 1024 attempts (lower bounds: true dependencies only)")
    print("A. waves cut at the first goal hit:")
    for W in (64, 96, 128, 192, 256):
        a, waves, rounds = k0, 0, 0
        while a < k1:
            b = min(a + W, k1)
            hit = np.nonzero(is_hit[a:b])[0]
            if len(hit):
                b = a + hit[0] + 1
            dp = wave_depths(a, b, True)
            rounds += max(dp.values())
            waves += 1
            a = b
        s = 1024.0 / len(T)
        print("   W <= %4d: %5.1f waves, %5.1f repair rounds -> %5.1f full slots (+ %4.1f confirm + append launches)"
              % (W, waves * s, rounds * s, (waves + rounds) * s, waves * s))
    print("B. waves not cut at goal hits:")
    for W in (128, 256, 512, 1024):
        a, waves, rounds = k0, 0, 0
        while a < k1:
            b = min(a + W, k1)
            dp = wave_depths(a, b, False)
            rounds += max(dp.values())
            waves += 1
            a = b
        s = 1024.0 / len(T)
        print("   W  = %4d: %5.1f waves, %5.1f repair rounds -> %5.1f full slots (+ %4.1f confirm + append launches)"
              % (W, waves * s, rounds * s, (waves + rounds) * s, waves * s))
    # Round 6: schedule B re-priced as it could be BUILT today -- hits applied by the host (the host hears every
    # round's summary and already passes ignore words as scan arguments), only the samples whose chosen node lies on the new path
    # re-select (that is what dep_hits holds: the hits that changed a sample's choice), the second-choice rule for deferred samples
    # (it removes idle rounds, not dependencies: the bound is unchanged by it).  Every uncut wave still needs the scan of its
    # goal-biased remainder after a hit; that scan rides with a round and is not counted as a slot.
    print("B'. waves not cut at goal hits, hits applied by the HOST (two rounds after the hit is final):")
    resB = {}
    for W in (128, 256, 512, 1024):
        a, waves, rounds = k0, 0, 0
        while a < k1:
            b = min(a + W, k1)
            dp = wave_depths(a, b, False, hit_lag=2)
            rounds += max(dp.values())
            waves += 1
            a = b
        s = 1024.0 / len(T)
        resB[W] = (waves + rounds) * s
        print("   W  = %4d: %5.1f waves, %5.1f repair rounds -> %5.1f full slots" % (W, waves * s, rounds * s, (waves + rounds) * s))
    a, wavesA, roundsA = k0, 0, 0
    while a < k1:
        b = min(a + 256, k1)
        hit = np.nonzero(is_hit[a:b])[0]
        if len(hit):
            b = a + hit[0] + 1
        roundsA += max(wave_depths(a, b, True).values())
        wavesA += 1
        a = b
    slotsA = (wavesA + roundsA) * 1024.0 / len(T)
    print("   schedule A (today's, W <= 256): %.1f slots; best buildable B': %.1f slots at W = %d -> bound on the gain in slots: %+.1f %%"
          % (slotsA, min(resB.values()), min(resB, key=resB.get), 100.0 * (slotsA / min(resB.values()) - 1.0)))
    print("   (a slot of a 1024-sample wave is not the slot of a 256-sample wave: beyond 256 samples a wave leaves the fused rounds -- the round")
    print("    prologue keeps 4 x 64 flags in registers -- and beyond 341 rollouts the chain rollout's three wavefronts share SIMDs)")
    print("C. sliding window (every launch is a round for all samples in flight; a sample is final in the round that decides it from final")
    print("   inputs; a final hit is applied when that round closes; entrants are scanned `lag` rounds after the slot was freed):")
    for lag in (1, 2):
      for W in (128, 256, 512, 1024):
        f, c, e = {}, {}, {}
        hit_ready = 0               # first round that sees the newest final hit's ignore set
        run_max = 0                 # max f over the attempts so far = round in which the prefix through t became final
        nhitdep = 0
        for t in T:
            tw = t - W
            et = 1 if tw < k0 else c[tw] + lag
            if t - 1 in e:
                et = max(et, e[t - 1])
            e[t] = et
            ft = et
            cpar = dep_parent[t]
            if cpar >= k0:
                ft = max(ft, f[cpar] + 1)
            for h in dep_hits[t]:
                if h >= k0:
                    ft = max(ft, c[h] + 1)             # re-decides in the round after the hit was applied
            f[t] = ft
            run_max = max(run_max, ft, hit_ready)       # (samples behind an applied hit are re-checked in the round that sees it)
            c[t] = run_max
            if is_hit[t]:
                hit_ready = c[t] + 1
        total = c[T[-1]]
        print("   lag %d  W = %4d: %5.1f rounds per 1024 attempts (mean residence %.1f rounds)"
              % (lag, W, 1024.0 * total / len(T), np.mean([c[t] - e[t] + 1 for t in T])))
    # how long is the part of a new goal path that is not ignored yet?
    fresh = []
    for t in np.nonzero(is_hit[k0:k1])[0] + k0:
        fresh.append(int(np.sum(ign_time == t)))
    print("\nnewly ignored nodes per goal hit: mean %.1f, max %d" % (np.mean(fresh), max(fresh)))


if __name__ == "__main__":
    main()
