import os, sys, tempfile, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
os.environ["PLC_GRAPH_STEP"] = "1"; os.environ["PLC_SHUFFLE_SEED"] = "3"
import torch, numpy as np
from pl_convlstm_gan_amd.host import config as C
from pl_convlstm_gan_amd.host.datasets import write_synthetic_dataset
tmp = tempfile.mkdtemp()
paths = write_synthetic_dataset(os.path.join(tmp, "basin"), start_year=2012, end_year=2013, lr_hw=(15, 12), scale=8, n_stations=35)
cfg = C.Config()
for k, v in paths.items(): setattr(cfg.data, k, v)
cfg.model.scale_factor = 8; cfg.training.batch_size = 8; cfg.training.epochs = 1
cfg.training.train_years, cfg.training.val_years, cfg.training.test_years = [2012, 2012], [2013, 2013], [2013, 2013]
cfg.output.output_dir = os.path.join(tmp, "out"); cfg.output.log_interval = 10 ** 6
os.makedirs(os.path.join(tmp, "configs")); cfg.to_yaml(os.path.join(tmp, "configs", "synthetic.yaml"))
C.CONFIG_DIR = os.path.join(tmp, "configs")
from pl_convlstm_gan_amd.host.trainer import Trainer
torch.manual_seed(1)
t = Trainer("synthetic"); t.setup_model(t.setup_data())
t.train_epoch(0, cfg.model.T); torch.cuda.synchronize()
seg = {"load": 0.0, "h2d": 0.0, "step": 0.0, "reads": 0.0, "rmse": 0.0}; n = 0
it = iter(t.train_loader)
while True:
    t0 = time.perf_counter()
    try: batch = next(it)
    except StopIteration: break
    t1 = time.perf_counter()
    dev = t._to_device(batch); torch.cuda.synchronize(); t2 = time.perf_counter()
    step = t._graphed_step(dev)
    loss, parts, fake, skipped = step(*dev); torch.cuda.synchronize(); t3 = time.perf_counter()
    _ = bool(skipped); _ = loss.item(); _ = [float(parts[k]) for k in ("point", "conserve", "smooth", "temporal")]; t4 = time.perf_counter()
    with torch.no_grad():
        r = t.compute_station_rmse(fake, dev[3], dev[4], 8.0)[0].item()
    t5 = time.perf_counter()
    for k, v in zip(seg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): seg[k] += v
    n += 1
print({k: round(v / n * 1e3, 3) for k, v in seg.items()}, "ms per batch over", n)
