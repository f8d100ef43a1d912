"""`python bench.py --gpus N` must produce an N-rank line or fail: the self-launch, on CPU.

No GPU work: `--launch-check` makes every rank join the process group, all_reduce 2**rank and rank 0 print what it saw."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(argv, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + argv, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


def _json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    p = _run(["--gpus", "2", "--backend", "gloo", "--launch-check"])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _json(p.stdout)
    assert d == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "backend": "gloo"}


def test_gpus_3_without_a_launcher_starts_three_ranks():
    d = _json(_run(["--gpus", "3", "--backend", "gloo", "--launch-check"]).stdout)
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3


def test_rccl_with_fewer_devices_than_ranks_is_an_error_not_a_smaller_run():
    import torch

    have = torch.cuda.device_count()
    p = _run(["--gpus", str(have + 1 if have else 2), "--launch-check"])            # backend nccl (the default)
    assert p.returncode == 2 and "visible GPUs" in p.stderr and not p.stdout.strip()


def test_a_launcher_with_another_world_size_is_an_error():
    p = _run(["--gpus", "4", "--backend", "gloo", "--launch-check"], env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr


def test_single_gpu_workloads_refuse_n_ranks():
    p = _run(["--gpus", "2", "--backend", "gloo", "--workload", "warp"])
    assert p.returncode == 2 and "single-GPU" in p.stderr


def test_gpus_1_launches_nothing():
    d = _json(_run(["--gpus", "1", "--launch-check"]).stdout)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1
