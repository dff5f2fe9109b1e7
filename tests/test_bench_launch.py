"""`bench.py --gpus N` outside torchrun starts its N rank processes itself (one worker per GPU, as
the reference trainer spawns them, graphvae/train/train_ddp.py:417-421) and prints rank 0's JSON
line as its only stdout."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra, timeout):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_batch_per_gpu_under_both_scalings():
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.per_gpu_batch("c2", "weak", 8) == 256
    assert bench.per_gpu_batch("c2", "strong", 8) == 32
    assert bench.per_gpu_batch("c3", "strong", 8) == 8
    assert bench.per_gpu_batch("c3", "weak", 8) == 64
    assert bench.per_gpu_batch("c5", "strong", 4) == 128
    ns = lambda w, s: type("A", (), {"workload": w, "scaling": s})()   # noqa: E731
    assert bench.scaling_of(ns("c2", None)) == "weak"
    assert bench.scaling_of(ns("c4", None)) == "strong"
    assert bench.scaling_of(ns("c4", "weak")) == "weak"


@pytest.mark.timeout(120)
def test_launcher_without_gpu_exits_nonzero():
    """No visible GPU: the launcher says so and starts nothing."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"],
             {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}, 100)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "GPU" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_two_ranks_from_one_command():
    """Two ranks on the one GPU of the box need the gloo backend (RCCL wants a device per rank)."""
    r = _run(["--gpus", "2", "--workload", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
             {"GVAE_DIST_BACKEND": "gloo"}, 360)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 32
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_more_ranks_than_gpus_fails_fast_under_rccl():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--workload", "c1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
             {"GVAE_DIST_BACKEND": "nccl"}, 100)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "one GPU per rank" in r.stderr
