"""A recorded run of the trainer's command line: wall time, graphs per second, peak device memory
and the epoch losses it printed.
usage: python benchmarks/trainer_run.py OUT.json -- <arguments of python -m graphvae.train.train_ddp>"""
import contextlib
import io
import json
import re
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "graph-vae_amd"))
from graphvae.train import train_ddp as T  # noqa: E402


class _Tee(io.TextIOBase):
    def __init__(self, *sinks):
        self.sinks = sinks

    def write(self, s):
        for k in self.sinks:
            k.write(s)
            k.flush()
        return len(s)


def main():
    out = Path(sys.argv[1])
    argv = sys.argv[sys.argv.index("--") + 1:]
    buf = io.StringIO()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(_Tee(sys.__stdout__, buf)):
        T.main(argv)                 # raises if the matching kernels report a fault in any epoch
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [(int(e), float(a), float(b)) for e, a, b in
              re.findall(r"Ep (\d+) \| train ([-\d.einfa+]+) \| val ([-\d.einfa+]+)", buf.getvalue())]
    log_dir = argv[argv.index("--log_dir") + 1]
    import yaml
    params = yaml.safe_load((Path(log_dir) / "params.yaml").read_text())
    n_train = int(0.9 * 600)
    steps = len(losses) * ((n_train + params["batch_size"] - 1) // params["batch_size"])
    rec = {"command": "python -m graphvae.train.train_ddp " + " ".join(argv), "wall_s": dt,
           "epochs": len(losses), "optimisation_steps": steps,
           "graphs_per_s_incl_startup_validation_and_save": len(losses) * n_train / dt,
           "peak_device_memory_MiB": torch.cuda.max_memory_allocated() / 2 ** 20,
           "matching_faults": 0, "epoch_losses_train_val": losses, "params": params,
           "device": torch.cuda.get_device_name(0)}
    out.write_text(json.dumps(rec, indent=1))
    print(json.dumps({k: rec[k] for k in ("wall_s", "optimisation_steps", "peak_device_memory_MiB")}))


if __name__ == "__main__":
    main()
