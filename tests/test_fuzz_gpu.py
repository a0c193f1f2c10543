"""Randomised bit-for-bit comparison of the HIP engine (speculative waves) with the sequential C oracle.

Each case draws a system, scenario, tree size, wave size (1..1024), seed, sampler tries, pruning / stop-on-goal
and fixed / adaptive horizon (tools/fuzz_parity.py).  The reference semantics being checked are those of
planner.py:233-290 executed strictly one sample at a time."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_match_sequential_oracle(seed):
    import fuzz_parity
    bad = fuzz_parity.run(60, seed)
    assert not bad, bad


@pytest.mark.parametrize("wave", [64, 256])
def test_vanishing_goal_hit_regression(wave):
    """Car, seed 405465: a goal hit at wave index 34 cut the wave, vanished three rounds later, and samples beyond
    it had been steered from an in-wave parent's superseded end state."""
    import fuzz_parity
    assert not fuzz_parity.run(323, 3, only=322, wave_override=wave, names=fuzz_parity.NAMES[:6])


def test_poisoned_allocations():
    """LQRRT_POISON=1 fills every device allocation of the engine with 0xff: a kernel that reads something it (or the
    host) never wrote -- harmless on a fresh process where new memory is zero, wrong when memory is recycled, as an
    uninitialised ignore bitmap after lqrrt_tree_load once was -- fails here.  Own process: the switch is read once."""
    import subprocess
    env = dict(os.environ, LQRRT_POISON="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "31"], capture_output=True,
                         text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-k", "warm_start or teacher_forced",
                          os.path.join(ROOT, "tests", "test_teacher_gpu.py")], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_legacy_repair_rounds_still_match():
    """Whole waves of <= 256 samples repair themselves with fused rounds (RoundArgs, kernels.hpp: decision, re-steer and
    append in one launch per round).  LQRRT_FUSED_ROUNDS=0 selects the k_decide + re-steer + k_append sequence that
    sharded waves, larger waves and Riccati systems still use; it has to give the same trees.  Own process: the switch
    is read once."""
    import subprocess
    env = dict(os.environ, LQRRT_FUSED_ROUNDS="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "41"], capture_output=True,
                         text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("wavefronts", [2, 3])
def test_rollout_wavefront_modes_match(wavefronts):
    """The boats with the heading torque spread one rollout over 3 (the chain rollout) or 2 wavefronts depending on the launch size
    (kernels.hpp, DuoLds; lqrrt_amd/csrc/engine.hip steer_wavefronts).  Every mode has to reproduce the sequential oracle
    bit for bit; LQRRT_STEER_WAVEFRONTS forces one mode for all launches.  Own process: the switch is read once."""
    import subprocess
    env = dict(os.environ, LQRRT_STEER_WAVEFRONTS=str(wavefronts))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "50", str(50 + wavefronts)],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
