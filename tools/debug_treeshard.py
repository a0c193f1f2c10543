import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_sharded_gpu import _make, _make_di
from lqrrt_amd.parallel import node_range
for name, nodes, world in (("double_integrator", 600, 2), ("double_integrator", 600, 4), ("boat_advanced", 800, 4), ("car", 600, 3), ("pendulum", 200, 2)):
    wave = 256
    mk = (lambda: _make_di(nodes + wave + 8, wave)[1]) if name == "double_integrator" else (lambda: _make(name, nodes + wave + 8, wave)[1])
    ref = mk(); ref.extend(wave, node_limit=nodes)
    ranks = [mk() for _ in range(world)]
    bufs = [torch.empty((world, wave, 2), dtype=torch.float64, device="cuda") for _ in range(world)]
    first_bad = None
    while ranks[0].size <= nodes:
        W = ranks[0].wave_suggest(wave)
        views = [b.view(-1)[: world * W * 2].view(world, W, 2) for b in bufs]
        for r, e in enumerate(ranks):
            lo, hi = node_range(e.size, r, world)
            e.wave_scan_nodes(W, lo, hi, views[r][r].data_ptr())
        torch.cuda.synchronize()
        for r in range(world):
            for q in range(world):
                if q != r: views[q][r].copy_(views[r][r])
        torch.cuda.synchronize()
        n0 = ranks[0].size
        for r, e in enumerate(ranks): e.wave_steer_candidates(W, world, views[r].data_ptr())
        sts = [e.wave_commit(W, W, nodes) for e in ranks]
        p = ranks[0].parents(); q = ref.parents()[:len(p)]
        if first_bad is None and not np.array_equal(p, q):
            first_bad = (n0, W, int(np.flatnonzero(p != q)[0]), [node_range(n0, r, world) for r in range(world)])
    print(name, world, "OK" if first_bad is None else first_bad)
