"""bench.py's launcher logic (no GPU needed): `python bench.py --gpus N` must start N ranks by itself, one per GPU, the way the
driver's own N > 1 command does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_1_is_not_respawned(bench):
    assert bench.spawn_command(bench.parse(["--gpus", "1"]), ["--gpus", "1"], {}, 8) is None
    assert bench.spawn_command(bench.parse([]), [], {}, 1) is None


def test_gpus_n_reexecutes_under_torch_distributed_run(bench):
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    cmd = bench.spawn_command(bench.parse(argv), argv, {"PATH": "/usr/bin"}, 8)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv                       # every rank sees the same command line


def test_a_rank_is_not_respawned_and_a_wrong_world_is_refused(bench):
    argv = ["--gpus", "2"]
    assert bench.spawn_command(bench.parse(argv), argv, {"WORLD_SIZE": "2", "RANK": "1"}, 2) is None
    with pytest.raises(SystemExit) as e:
        bench.spawn_command(bench.parse(argv), argv, {"WORLD_SIZE": "1"}, 2)
    assert "WORLD_SIZE=1" in str(e.value)


def test_too_few_devices_is_an_error_not_an_n1_line(bench):
    argv = ["--gpus", "8"]
    with pytest.raises(SystemExit) as e:
        bench.spawn_command(bench.parse(argv), argv, {}, 1)
    assert "--gpus 8" in str(e.value) and "only 1 HIP device" in str(e.value)
