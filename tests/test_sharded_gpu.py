"""
GPU: the sample-sharded wave path (lqrrt_wave_speculate on a slice + record exchange +
lqrrt_wave_commit on every rank) must give exactly the tree of the single-engine path.

One GPU is enough to check the data path: two engines play rank 0 and rank 1, the exchange that
RCCL's all-gather performs between GPUs is done with tensor copies between their record buffers
(viewed through lqrrt_amd.parallel.records_tensor, the same zero-copy view the real collective uses).
A second test runs the real torch.distributed code path with a world of one process (NCCL=RCCL).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(name, cap, wave, seed=1):
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.SYSTEMS[name](0)
    eng = Engine(s, capacity=cap, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(seed).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    return s, eng


@pytest.mark.parametrize("name,nodes,world", [("boat_advanced", 1500, 2), ("car", 1200, 2), ("boat_advanced", 2500, 8)])
def test_two_rank_emulation_matches_single_engine(name, nodes, world):
    import torch
    from lqrrt_amd.parallel import pick_wave, records_tensor, shard_bounds
    wave = 256
    _, ref = _make(name, nodes + wave + 8, wave)
    ref_stats = ref.extend(wave, node_limit=nodes)
    ranks = [_make(name, nodes + wave + 8, wave)[1] for _ in range(world)]
    recs = [records_tensor(e) for e in ranks]
    attempts = 0
    while ranks[0].size <= nodes:
        W = ranks[0].wave_suggest(wave)
        assert all(e.wave_suggest(wave) == W for e in ranks)       # the policy is deterministic across replicas
        bounds = [shard_bounds(W, r, world) for r in range(world)]
        for r, e in enumerate(ranks):
            e.wave_speculate(W, bounds[r][1], bounds[r][2])
        torch.cuda.synchronize()
        for r in range(world):                       # what all_gather_into_tensor does across GPUs
            lo, hi = bounds[r][1], bounds[r][2]
            for q in range(world):
                if q != r:
                    recs[q][lo:hi].copy_(recs[r][lo:hi])
        torch.cuda.synchronize()
        sts = [e.wave_commit(W, W, nodes) for e in ranks]
        assert len({(st.attempts, st.accepted, st.tree_size) for st in sts}) == 1
        attempts += sts[0].attempts
    for e in ranks:
        assert e.size == ref.size
        np.testing.assert_array_equal(e.parents(), ref.parents())
        np.testing.assert_array_equal(e.states(), ref.states())
        np.testing.assert_array_equal(e.edge_lengths(), ref.edge_lengths())
        np.testing.assert_array_equal(e.ignored(), ref.ignored())
    assert attempts == ref_stats.attempts


def test_sharded_wave_class_world_of_one_nccl():
    import os
    import torch
    import torch.distributed as dist
    from lqrrt_amd.parallel import ShardedWave
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        _, ref = _make("boat_intermediate", 900, 128)
        ref.extend(128, node_limit=600)
        _, eng = _make("boat_intermediate", 900, 128)
        sw = ShardedWave(eng, dist, 0, 1)
        while eng.size <= 600:
            sw.wave(128, max_commit=128, node_limit=600)
        np.testing.assert_array_equal(eng.parents(), ref.parents())
        np.testing.assert_array_equal(eng.states(), ref.states())
    finally:
        dist.destroy_process_group()


def _make_di(cap, wave, seed=1, n_boxes=3000):
    import lqrrt_amd
    from lqrrt_amd.engine import Engine
    s = lqrrt_amd.systems.DoubleIntegrator(n_boxes=n_boxes, seed=0)
    eng = Engine(s, capacity=cap, max_wave=wave)
    kw = s.plan_kwargs
    eng.set_resolution(kw["dt"], kw["FPR"], int(kw["horizon"] / kw["dt"]), np.abs(s.error_tol), s.goal, np.abs(s.goal_buffer))
    space = np.array(s.sample_space, dtype=np.float64)
    eng.set_sampler(np.mean(space, axis=1), np.diff(space).flatten(), np.array(s.goal_bias, dtype=np.float64), 10)
    st = np.random.RandomState(seed).get_state()
    eng.set_mt19937(st[1], st[2])
    eng.tree_reset(s.x0)
    return s, eng


@pytest.mark.parametrize("name,nodes,world", [("boat_advanced", 1500, 2), ("double_integrator", 2500, 4), ("car", 900, 3)])
def test_tree_sharded_emulation_matches_single_engine(name, nodes, world):
    """Tree-sharded waves (lqrrt_wave_scan_nodes + candidate exchange + lqrrt_wave_steer_candidates): `world` engines on
    one GPU play the ranks, the all-gather of the (cost, id) candidates is done with tensor copies."""
    import torch
    from lqrrt_amd.parallel import node_range
    wave = 256
    mk = (lambda: _make_di(nodes + wave + 8, wave)[1]) if name == "double_integrator" else (lambda: _make(name, nodes + wave + 8, wave)[1])
    ref = mk()
    ref_stats = ref.extend(wave, node_limit=nodes)
    ranks = [mk() for _ in range(world)]
    bufs = [torch.empty((world, wave, 2), dtype=torch.float64, device="cuda") for _ in range(world)]
    attempts = 0
    while ranks[0].size <= nodes:
        W = ranks[0].wave_suggest(wave)
        assert all(e.wave_suggest(wave) == W for e in ranks)
        views = [b.view(-1)[: world * W * 2].view(world, W, 2) for b in bufs]
        for r, e in enumerate(ranks):
            lo, hi = node_range(e.size, r, world)
            e.wave_scan_nodes(W, lo, hi, views[r][r].data_ptr())
        torch.cuda.synchronize()
        for r in range(world):                       # all_gather_into_tensor across GPUs
            for q in range(world):
                if q != r:
                    views[q][r].copy_(views[r][r])
        torch.cuda.synchronize()
        for r, e in enumerate(ranks):
            e.wave_steer_candidates(W, world, views[r].data_ptr())
        sts = [e.wave_commit(W, W, nodes) for e in ranks]
        assert len({(st.attempts, st.accepted, st.tree_size) for st in sts}) == 1
        attempts += sts[0].attempts
    for e in ranks:
        assert e.size == ref.size
        np.testing.assert_array_equal(e.parents(), ref.parents())
        np.testing.assert_array_equal(e.states(), ref.states())
        np.testing.assert_array_equal(e.edge_lengths(), ref.edge_lengths())
        np.testing.assert_array_equal(e.ignored(), ref.ignored())
    assert attempts == ref_stats.attempts


def test_tree_sharded_wave_class_world_of_one_nccl():
    import os
    import socket
    import torch
    import torch.distributed as dist
    from lqrrt_amd.parallel import TreeShardedWave
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        _, ref = _make("boat_intermediate", 900, 128)
        ref.extend(128, node_limit=600)
        _, eng = _make("boat_intermediate", 900, 128)
        sw = TreeShardedWave(eng, dist, 0, 1)
        while eng.size <= 600:
            sw.wave(128, max_commit=128, node_limit=600)
        np.testing.assert_array_equal(eng.parents(), ref.parents())
        np.testing.assert_array_equal(eng.states(), ref.states())
    finally:
        dist.destroy_process_group()
