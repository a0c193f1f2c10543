#!/usr/bin/env python
"""Where does a sharded wave spend its time?  World of one process (NCCL=RCCL), bench-like 10k-node boat tree."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
from lqrrt_amd.parallel import ShardedWave, shard_bounds

import socket
with socket.socket() as _sk:
    _sk.bind(("127.0.0.1", 0)); _port = _sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
boat, eng = bench.build_problem(10000, 1024, 0)
eng.extend(1024, until_size=9500)
eng.tree_mark()
sw = ShardedWave(eng, dist, 0, 1)
sync = torch.cuda.synchronize
for mode in ("phases", "free", "native"):
    eng.tree_rewind()
    T = dict(spec=0.0, gather=0.0, commit=0.0)
    waves = attempts = 0
    sync(); t0 = time.perf_counter()
    while attempts < 60000:
        if eng.size > 10100:
            eng.tree_rewind()
        if mode == "native":
            st = eng.extend(1024, max_attempts=1024)
            waves += st.waves
        elif mode == "free":
            st = sw.wave(1024, max_commit=1024)
            waves += 1
        else:
            W = eng.wave_suggest(1024)
            per, lo, hi = shard_bounds(W, 0, 1)
            a = time.perf_counter(); eng.wave_speculate(W, lo, hi); sync()
            b = time.perf_counter()
            send = sw.rec[:per].clone(); dist.all_gather_into_tensor(sw.rec[:per].view(-1), send.view(-1)); sync()
            c = time.perf_counter(); st = eng.wave_commit(W, 1024, -1); sync()
            d = time.perf_counter()
            T["spec"] += b - a; T["gather"] += c - b; T["commit"] += d - c
            waves += 1
        attempts += st.attempts
    sync(); el = time.perf_counter() - t0
    print("%-7s %7.0f attempts/s  waves %d  (%.1f attempts/wave, %.0f us/wave)" % (mode, attempts / el, waves, attempts / waves, 1e6 * el / waves),
          {k: "%.0f us" % (1e6 * v / waves) for k, v in T.items()} if mode == "phases" else "")
dist.destroy_process_group()
