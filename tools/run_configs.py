#!/usr/bin/env python
"""Every BASELINE.json configuration on one GPU, exact mode: grow the tree from x0 to the config's size, then measure
extension attempts/s with the tree held within +-5 % of it (bench.py's protocol).  Not bench lines -- context for them."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lqrrt_amd
from lqrrt_amd.engine import Engine

CONFIGS = [
    ("cfg1 pendulum, 500 nodes", "pendulum", {}, 500, None),
    ("cfg2 car, 2k nodes", "car", {}, 2000, None),
    ("cfg3 boat_novice, 5k nodes (error_tol = goal_buffer/8)", "boat_novice", {}, 5000, "tight"),
    ("cfg4 boat_advanced, 10k nodes (headline)", "boat_advanced", {}, 10000, None),
    ("cfg5 12-DoF double integrator, 100k boxes, 50k nodes", "double_integrator", dict(n_boxes=100000, seed=0), 50000, None),
    # the north-star steer pipeline (finite-difference linearise -> doubling DARE -> K per recorded step, S per sample)
    ("Riccati pendulum (4 states, dt 1 ms), 500 nodes", "pendulum_lqr", {}, 500, None),
    ("Riccati boat_novice (6 states, 3 controls), 3k nodes", "boat_novice_lqr", {}, 3000, None),
]
ONLY = os.environ.get("RC_ONLY")                     # substring filter on the label (tools/riccati_ab.sh)
for label, name, kw_sys, nodes, mod in CONFIGS:
    if ONLY and ONLY not in label:
        continue
    cls = lqrrt_amd.systems.SYSTEMS[name]
    s = cls(**kw_sys) if kw_sys else cls(0)
    if mod == "tight":
        s.error_tol = np.array(s.goal_buffer, dtype=np.float64) / 8.0
    wave = 1024
    eng = Engine(s, capacity=int(nodes * 1.06) + 2 * wave + 64, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(1).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    lo, hi = int(nodes * 0.95), int(nodes * 1.05)
    t0 = time.perf_counter()
    g = eng.extend(wave, until_size=lo, max_attempts=60 * nodes)
    t_grow = time.perf_counter() - t0
    if eng.size < lo:
        print("%-58s tree saturated at %d nodes after %d attempts" % (label, eng.size, g.attempts))
        eng.close()
        continue
    eng.tree_mark()
    eng.profile_enable(True, steer=True)
    done = acc = waves = rounds = resteers = 0
    t0 = time.perf_counter()
    for _ in range(60):
        if eng.size > hi - 0.7 * wave:
            eng.tree_rewind()
        st_ = eng.extend(wave, max_attempts=wave)
        done += st_.attempts; acc += st_.accepted
        waves += st_.waves; rounds += st_.fix_rounds; resteers += st_.resteers
    dt = time.perf_counter() - t0
    pr = eng.profile_read()
    print("%-58s %9.0f attempts/s  yield %4.1f %%  grow %6.2f s (%7d attempts)  NN %6.1f us x%-5d %6.2f TB/s alg.  steer %6.1f us" % (
        label, done / dt, 100.0 * acc / max(1, done), t_grow, g.attempts, 1e3 * pr["nn_ms"] / max(1, pr["nn_launches"]), pr["nn_launches"],
        pr["nn_bytes"] / 1e12 / max(1e-9, pr["nn_ms"] / 1e3), 1e3 * pr["steer_ms"] / max(1, pr["steer_launches"])))
    if os.environ.get("RC_DETAIL"):
        k = 1024.0 / max(1, done)
        print("%-58s per 1024 attempts: %.1f waves (mean %.0f samples), %.1f repair rounds, %.0f re-steers" % ("", k * waves, done / max(1, waves), k * rounds, k * resteers))
    if getattr(s, "riccati", False):
        # what one Riccati gain costs: the batched operator, one wavefront per problem (what the rollout calls per recorded step)
        import torch
        xs = eng.states()[: min(eng.size, 2048)]
        us = np.zeros((len(xs), s.ncontrols))
        eng.lqr_dare_batch(xs[:64], us[:64], s.Q, s.R, s.eps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _S, _K, _A, _B, iters = eng.lqr_dare_batch(xs, us, s.Q, s.R, s.eps)
        torch.cuda.synchronize()
        dt5 = (time.perf_counter() - t0) / 5
        print("%-58s lqr_dare_batch of %d states: %.1f us per batch (incl. copies), %.1f doubling iterations on average" % (
            "", len(xs), 1e6 * dt5, float(np.mean(iters))))
    eng.close()
