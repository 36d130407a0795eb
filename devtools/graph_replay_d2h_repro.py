"""Reproducer kept for the record: replays of the captured training step (training.GraphedTrainStep) deviate
from the eager step -- and end in NaN parameters -- when, between replays, (r) the host reads the prediction
through a few indexing kernels (the Trainer's station RMSE) AND (d or p) device tensors are copied to pageable
host memory (`.cpu()` of the optimizer state or of the parameters, as a checkpoint does).  Either alone is
harmless, and so are a plain synchronisation (y), device-side clones (c) and emptying the allocator's cache (e):

    python3 devtools/graph_replay_d2h_repro.py rd     -> "final equal: False nan", deviation from the 13th step
    python3 devtools/graph_replay_d2h_repro.py r      -> bit-identical            (also: d, ry, rc, re)

Every buffer the captured launches point into is alive throughout (static inputs, parameters, optimizer state,
the lazy block's accumulated gradients, the graph's own pool), so the overwrite does not come from a freed
tensor of this package; the cause was not found.  It depends on the allocation pattern: cloning the persistent
state after every step to compare it makes the deviation disappear.  host/trainer.py therefore runs its steps
eagerly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.data import DataLoader
from pl_convlstm_gan_amd import Generator
from pl_convlstm_gan_amd.losses import CombinedLoss
from pl_convlstm_gan_amd.optim import FusedAdam
from pl_convlstm_gan_amd.training import GraphedTrainStep, train_step
from pl_convlstm_gan_amd.host.datasets import write_synthetic_dataset, FenheDataset
MODE = sys.argv[1] if len(sys.argv) > 1 else ""
gpu = torch.device("cuda:0")
paths = write_synthetic_dataset("/tmp/r2_basin", start_year=2012, end_year=2012, lr_hw=(15, 12), scale=2, n_stations=10)
ds = FenheDataset(rain_lr_path=paths["rain_lr_path"], dem_path=paths["dem_path"], lucc_path=paths["lucc_path"],
                  rain_meta_path=paths["meta_path"], rain_station_path=paths["rain_excel_path"], shp_path=paths.get("shp_path"),
                  T=5, start_year=2012, end_year=2012)
L = (ds.lucc_onehot_list[0] if getattr(ds, "is_multiyear_lucc", False) else ds.lucc_onehot).shape[0]
def setup():
    torch.manual_seed(7)
    m = Generator(hidden_dims=[16, 32], lu_channels=L, scale_factor=2).to(gpu)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=10) if "s" in MODE else None
    return m, opt, CombinedLoss(), sch
flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
def run(graph):
    m, opt, lossm, sch = setup()
    loader = DataLoader(ds, batch_size=48, shuffle=True, drop_last=True)
    out, step, first, last = [], None, True, None
    for epoch in range(2):
        for batch in loader:
            b = tuple(t.to(gpu) for t in batch)
            if graph and not first:
                if step is None:
                    step = GraphedTrainStep(m, lossm, opt, b, max_norm=0.5)
                loss, parts, fake, skipped = step(*b)
            else:
                loss, parts, fake, skipped = train_step(m, lossm, opt, *b, max_norm=0.5, nan_check="device")
            first = False
            loss, fake = loss.detach(), fake.detach()
            sk = bool(skipped)
            li = loss.item()
            out.append(li)
            if "r" in MODE:
                pv = [float(parts[kk].detach()) for kk in ("point", "conserve", "smooth", "temporal")]
                Bq, Tq, _, Hh, Wh = fake.shape
                coords = b[3][0]
                hr = ((coords.float() + 0.5) * 2.0 - 0.5).long()
                rows, cols = hr[:, 0], hr[:, 1]
                inside = (rows >= 0) & (rows < Hh) & (cols >= 0) & (cols < Wh)
                zero = torch.tensor(0.0, device=fake.device)
                ns = int(inside.sum())
                at = fake[:, :, 0][:, :, rows[inside], cols[inside]]
                truth = b[4][:, :, inside]
                seen = ~torch.isnan(truth)
                rm = torch.sqrt(((at[seen] - truth[seen]) ** 2).mean()) if int(seen.sum()) > 0 else zero
                ri = rm.item()
                last = (at, truth, b[3], inside)
        if "d" in MODE:
            sd = opt.state_dict()
            cp = [v.cpu() for st in sd["state"].values() for v in st.values() if torch.is_tensor(v)]
        if "y" in MODE:
            torch.cuda.synchronize()
        if "p" in MODE:
            cp = [p.detach().cpu() for p in m.parameters()]
        if "c" in MODE:
            cp = [st["exp_avg"].clone() for st in opt.state.values() if "exp_avg" in st]
        if "e" in MODE:
            torch.cuda.empty_cache()
    return flat(m), out
a, oa = run(False)
b, ob = run(True)
print("E", oa); print("G", ob); print("nan a", bool(torch.isnan(a).any()), "nan b", bool(torch.isnan(b).any()))
print("MODE", MODE, "final equal:", torch.equal(a, b), float((a - b).abs().max()), [x == y for x, y in zip(oa, ob)])
