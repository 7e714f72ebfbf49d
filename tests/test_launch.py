"""`python bench.py --gpus N` without a launcher: the process starts its own ranks
(pysdm_amd.launch) before anything touches a GPU.  CPU: gloo ranks of a probe script, and bench.py
itself, whose ranks must fail loudly here (no GPU, no CPU fallback) with the code relayed."""
import json
import os
import subprocess
import sys

import pytest

from pysdm_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "helpers", "launch_probe.py")


@pytest.mark.timeout(300)
def test_spawned_ranks_form_one_process_group(tmp_path):
    out = tmp_path / "out.txt"
    with open(out, "w", encoding="utf-8") as sink:
        code = launch.spawn_ranks(PROBE, ["--steps", "3"], 2, stdout=sink, timeout=240)
    assert code == 0
    lines = [l for l in out.read_text().splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line == {"n_gpus": 2, "sum": 3.0, "argv": ["--steps", "3"], "master": "127.0.0.1"}


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher(tmp_path):
    with open(tmp_path / "out.txt", "w", encoding="utf-8") as sink:
        assert launch.spawn_ranks(PROBE, ["--fail"], 2, stdout=sink, timeout=240) != 0


def test_launcher_detection():
    assert launch.launched_by_torchrun({"RANK": "0", "WORLD_SIZE": "2"})
    assert not launch.launched_by_torchrun({"WORLD_SIZE": "1"})
    assert not launch.launched_by_torchrun({})


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks_and_relays_their_failure():
    """no GPU here: each rank of `bench.py --gpus 2` must die with the no-GPU error (never fall
    back to a CPU path) and the parent must exit non-zero without printing a JSON line"""
    import torch  # pylint: disable=import-outside-toplevel

    if torch.cuda.is_available():
        pytest.skip("covered by the GPU rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                           "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                          text=True, timeout=280, check=False)
    assert done.returncode != 0
    assert not [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert "torch.distributed" in done.stderr or "ChildFailedError" in done.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_with_two_self_started_ranks_reproduces_the_one_rank_state(hip_engine):  # pylint: disable=unused-argument
    """`python bench.py --gpus 2` with no launcher in the environment, on the one-GPU box (both
    ranks on card 0, gloo between them - the rehearsal knobs): one JSON line, n_gpus = 2 as the
    process group saw it, the sharded state's digest equal to the one-rank run's, and the
    collectives of a step far below one n_sd-wide exchange"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(SDM_BENCH_DIST_BACKEND="gloo", SDM_BENCH_ALL_ON_DEVICE0="1")
    common = ["--workload", "kinematic2d", "--n-sd", str(2**17), "--steps", "6", "--warmup", "2",
              "--reps", "2", "--no-cpu-baseline", "--roofline-steps", "2"]
    lines = {}
    for gpus in (1, 2):
        done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus",
                               str(gpus), *common], env=env, capture_output=True, text=True,
                              timeout=280, check=False)
        assert done.returncode == 0, done.stderr[-2000:]
        found = [json.loads(l) for l in done.stdout.splitlines() if l.startswith("{")]
        assert len(found) == 1
        lines[gpus] = found[0]
    assert lines[1]["n_gpus"] == 1 and lines[2]["n_gpus"] == 2
    assert lines[2]["state_digest"] == lines[1]["state_digest"]
    assert lines[2]["scaling"] == "strong" and lines[2]["comm"]["collectives"] > 0
    assert lines[2]["comm"]["bytes_per_step"] < 8 * 2**17  # (no n_sd-wide exchange)
