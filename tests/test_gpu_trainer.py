"""The trainer's command line (reference graphvae/train/train_ddp.py:426-451, loop :271-355) at the
sizes of the larger BASELINE configurations: a few epochs at max_nodes 128 (dense graphs) and at
max_nodes 126 with `id` features (126 input columns), through `main(argv)` as
`python -m graphvae.train.train_ddp` runs it.  Checks: finite, decreasing epoch losses, no fault
reported by the matching kernels (the trainer checks once per epoch and would raise), and a
checkpoint with the reference's keys that loads strictly."""
import re

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dataset,feature,max_nodes", [("synthetic-dense", "deg", 128),
                                                       ("synthetic-enzymes", "id", 126)])
def test_trainer_cli_at_large_max_nodes(tmp_path, capfd, dataset, feature, max_nodes):
    from graphvae.models import GraphVAE
    from graphvae.train import train_ddp as T
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump({"trainer": {"warm_epochs": 1, "lr": 1e-3, "beta0": 0.1, "beta_final": 0.1,
                                               "batch_size": 64, "use_wandb": False}}))
    run = tmp_path / "run"
    T.main(["--dataset", dataset, "--feature_type", feature, "--max_nodes", str(max_nodes),
            "--epochs", "3", "--log_dir", str(run), "--config", str(cfg), "--no_wandb"])
    text = capfd.readouterr().out
    losses = [(float(a), float(b)) for a, b in
              re.findall(r"Ep \d+ \| train ([-\d.einfa+]+) \| val ([-\d.einfa+]+)", text)]
    assert len(losses) == 3, text
    assert all(torch.isfinite(torch.tensor(l)).all() for l in losses), losses
    assert losses[-1][0] < losses[0][0] and losses[-1][1] < losses[0][1], losses
    params = yaml.safe_load((run / "params.yaml").read_text())
    assert params["max_nodes"] == max_nodes and params["in_dim"] == (1 if feature == "deg" else max_nodes)
    sd = torch.load(run / "graphvae_ddp.pt", weights_only=True)
    model = GraphVAE(params["in_dim"], params["hid_dim"], params["z_dim"], max_nodes)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    assert int(sd["bn1.num_batches_tracked"]) == 3 * 540          # 600 graphs, 90 % train, 3 epochs
