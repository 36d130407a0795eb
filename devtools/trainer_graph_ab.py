"""Epoch time of the drop-in Trainer on a synthetic basin of the reference's own shape (15 x 12 grid, x8, batch 8,
T = 5: default.yaml), eager steps against graph-replayed steps (PLC_GRAPH_STEP=1):
    python3 devtools/trainer_graph_ab.py [0|1]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PLC_GRAPH_STEP"] = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["PLC_SHUFFLE_SEED"] = "3"

import torch  # noqa: E402
from pl_convlstm_gan_amd.host import config as C  # noqa: E402
from pl_convlstm_gan_amd.host.datasets import write_synthetic_dataset  # noqa: E402

tmp = tempfile.mkdtemp()
paths = write_synthetic_dataset(os.path.join(tmp, "basin"), start_year=2012, end_year=2013, lr_hw=(15, 12), scale=8, n_stations=35)
cfg = C.Config()
for k, v in paths.items():
    setattr(cfg.data, k, v)
cfg.model.scale_factor = 8
cfg.training.batch_size = 8
cfg.training.epochs = 1
cfg.training.train_years, cfg.training.val_years, cfg.training.test_years = [2012, 2012], [2013, 2013], [2013, 2013]
cfg.output.output_dir = os.path.join(tmp, "out")
cfg.output.log_interval = 10 ** 6
os.makedirs(os.path.join(tmp, "configs"))
cfg.to_yaml(os.path.join(tmp, "configs", "synthetic.yaml"))
C.CONFIG_DIR = os.path.join(tmp, "configs")
from pl_convlstm_gan_amd.host.trainer import Trainer  # noqa: E402

torch.manual_seed(1)
t = Trainer("synthetic")
t.setup_model(t.setup_data())
t.train_epoch(0, cfg.model.T)                      # warm-up epoch (captures the graphs)
torch.cuda.synchronize()
best = float("inf")
for e in range(3):
    t0 = time.perf_counter()
    t.train_epoch(e + 1, cfg.model.T)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
n = len(t.train_loader)
print(f"PLC_GRAPH_STEP={os.environ['PLC_GRAPH_STEP']}: {best * 1e3:.1f} ms per epoch of {n} batches = {best / n * 1e3:.3f} ms per batch "
      "(data loading, step, station RMSE, host reads of the loss)")
