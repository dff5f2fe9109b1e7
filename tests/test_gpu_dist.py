"""Two-rank data-parallel training steps on the GPU: two fresh child processes, both on cuda:0,
gloo collectives (a one-GPU box cannot give RCCL a device per rank), the HIP engine and every
schedule of TrainStep (one stream, matching on a second stream, two steps in flight).  Checks
that no rank stalls, that parameters are identical across ranks and that they equal a single
process's run on the union batch (reference graphvae/train/train_ddp.py:196-242,291: DDP
averages the ranks' gradients)."""
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import dist_worker  # noqa: E402

pytestmark = pytest.mark.gpu
STEPS = 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("matching,mode", [("fused", "sequential"), ("fused", "overlap"),
                                           ("fused", "pipeline"), ("stream", "pipeline")])
def test_two_ranks_on_one_gpu_equal_single_process(tmp_path, matching, mode):
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_worker.py"), str(r), "2", port,
                               str(tmp_path), matching, mode, str(STEPS)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=150)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("a rank of the two-rank step did not finish within 150 s")
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
    dp = np.load(tmp_path / "params_0.npy")
    assert np.array_equal(dp, np.load(tmp_path / "params_1.npy"))
    # single process, union batch, same noise
    dev = torch.device("cuda", 0)
    model, full, eps = dist_worker.make(3, matching, STEPS)
    model, step = dist_worker.run_steps(model, full, eps, list(range(full.num_graphs)), 1, mode, STEPS, dev)
    sp = model.flat_parameters().cpu().numpy()
    # the mean over ranks of per-rank means equals the mean over the union (equal shard sizes);
    # fp32 sums differ in order only.  The two GCN biases have analytically zero gradient
    # (BatchNorm cancels them), so Adam moves them by +-lr on rounding noise: excluded.
    from graphvae.engine import PARAM_NAMES
    mask = np.ones_like(sp, dtype=bool)
    for n, v in zip(PARAM_NAMES, model._views_of(torch.arange(sp.size, dtype=torch.float32))):
        if n in ("conv1.bias", "conv2.bias"):
            mask[v.reshape(-1).long().numpy()] = False
    assert np.abs(dp - sp)[mask].max() <= 2e-4 * np.abs(sp).max()
    g_dp, g_sp = float(np.load(tmp_path / "gnorm_0.npy")[0]), float(step.grad_norm)
    assert abs(g_dp - g_sp) <= 1e-3 * g_sp
