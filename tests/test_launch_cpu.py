"""`python bench.py --gpus N` must start N ranks by itself (rmt_app_amd/launch.py) - covered here on the
CPU: two children under torch.distributed.run with the gloo backend and a stand-in body, the refusal
when fewer GPUs than ranks are visible, and bench.py's own behaviour on a GPU-less machine."""
import json
import os
import subprocess
import sys

import pytest

from rmt_app_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "rank_stub.py")


def test_rank_command_is_the_drivers_launch_line():
    cmd = launch.rank_command(4, ["bench.py", "--gpus", "4"], port=29512, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                   "--master-addr", "127.0.0.1", "--master-port", "29512", "bench.py", "--gpus", "4"]


def test_spawn_two_ranks_gloo_stub(tmp_path):
    out = tmp_path / "ranks.json"
    rc = launch.spawn_ranks(2, [STUB, str(out)], require_gpus=False, timeout=300)
    assert rc == 0
    rec = json.loads(out.read_text())
    assert rec == {"rccl_ranks": 2, "max": 2.0, "per_rank": [1.0, 2.0], "local_rank": 0}


def test_a_failing_rank_fails_the_job(tmp_path):
    rc = launch.spawn_ranks(2, [STUB, str(tmp_path / "never.json"), "1"], require_gpus=False, timeout=300)
    assert rc != 0
    assert not (tmp_path / "never.json").exists()


def test_refuses_when_fewer_gpus_than_ranks(monkeypatch):
    monkeypatch.setattr(launch, "visible_gpus", lambda: 1)
    with pytest.raises(SystemExit) as e:
        launch.spawn_ranks(2, [STUB, "unused"])
    assert "2 ranks need 2 visible MI355X GPUs, this machine shows 1" in str(e.value)


def test_ranks_do_not_nest(monkeypatch):
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError):
        launch.spawn_ranks(2, [STUB, "unused"], require_gpus=False)


def _visible():
    return launch.visible_gpus()


@pytest.mark.skipif(_visible() >= 2, reason="needs a machine with fewer than two GPUs")
def test_bench_gpus_2_fails_loudly_without_two_gpus():
    """On a box with fewer than 2 GPUs `bench.py --gpus 2` must not quietly run one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "2 ranks need 2 visible MI355X GPUs" in (p.stderr + p.stdout)
    assert '"n_gpus"' not in p.stdout


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0 and "was started as one of 1 ranks" in (p.stderr + p.stdout)
