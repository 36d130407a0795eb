"""Instrumented form of graph_replay_d2h_repro.py: names the first buffer that differs from the eager run.

    python3 devtools/graph_replay_probe.py MODE [every|STOP]

The eager run records every persistent tensor after every step with per-tensor ``.cpu()`` copies (no GPU
allocation, so the allocator's pattern is the reproducer's).  The graph run compares the loss of every step and,
after step STOP (or after every step with ``every``), all persistent tensors and the captured step's own buffers
against the eager record, printing per tensor the number of differing elements and their index range."""
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
WHEN = sys.argv[2] if len(sys.argv) > 2 else "0"
EVERY = WHEN == "every"
STOP = 0 if EVERY else int(WHEN)
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
    return m, opt, CombinedLoss()


def state_of(m, opt):
    rec = {}
    for k, p in m.named_parameters():
        rec["p." + k] = p.detach().cpu()
        if p.grad is not None:
            rec["g." + k] = p.grad.detach().cpu()
        st = opt.state.get(p, {})
        for kk in ("exp_avg", "exp_avg_sq"):
            if kk in st:
                rec[kk + "." + k] = st[kk].cpu()
    if opt._dev_step is not None:
        rec["dev_step"] = opt._dev_step.cpu()
    return rec


def diff(tag, a, b):
    bad = 0
    for k in a:
        if k not in b:
            print(f"  {tag} {k}: missing in graph run"); continue
        x, y = a[k].reshape(-1), b[k].reshape(-1)
        ne = ~((x == y) | (torch.isnan(x) & torch.isnan(y)))
        n = int(ne.sum())
        if n:
            idx = ne.nonzero().reshape(-1)
            bad += 1
            print(f"  {tag} {k}: {n}/{x.numel()} differ, idx {int(idx[0])}..{int(idx[-1])}, max|d| {float((x - y).abs().nan_to_num(1e30).max()):.3e}")
    return bad


def run(graph, record):
    m, opt, lossm = setup()
    loader = DataLoader(ds, batch_size=48, shuffle=True, drop_last=True)
    losses, step, first, n = [], None, True, 0
    torch.manual_seed(99)           # the loader's shuffle
    for epoch in range(2):
        for batch in loader:
            n += 1
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
            losses.append(li)
            if "r" in MODE:
                Bq, Tq, _, Hh, Wh = fake.shape
                coords = b[3][0]
                hr = ((coords.float() + 0.5) * 2.0 - 0.5).long()
                rows, cols = hr[:, 0], hr[:, 1]
                inside = (rows >= 0) & (rows < Hh) & (cols >= 0) & (cols < Wh)
                ns = int(inside.sum())
                at = fake[:, :, 0][:, :, rows[inside], cols[inside]]
                truth = b[4][:, :, inside]
                seen = ~torch.isnan(truth)
                rm = torch.sqrt(((at[seen] - truth[seen]) ** 2).mean())
                ri = rm.item()
                last = (at, truth, b[3], inside)
            if not graph:
                record.append((li, state_of(m, opt), fake.cpu()))
            else:
                el, est, efake = record[n - 1]
                if li != el:
                    print(f"step {n}: loss {li!r} != eager {el!r}")
                if EVERY or n == STOP:
                    print(f"-- state after step {n} (graph vs eager)")
                    bad = diff("state", est, state_of(m, opt))
                    bad += diff("out", {"pred": efake}, {"pred": fake.cpu()})
                    if step is not None:
                        bad += diff("static", {f"in{i}": t.cpu() for i, t in enumerate(b)},
                                    {f"in{i}": t.cpu() for i, t in enumerate(step.static)})
                    print(f"-- {bad} tensors differ")
                    if n == STOP:
                        return losses
        if "d" in MODE:
            sd = opt.state_dict()
            cp = [v.cpu() for st in sd["state"].values() for v in st.values() if torch.is_tensor(v)]
        if "p" in MODE:
            cp = [p.detach().cpu() for p in m.parameters()]
    return losses


rec = []
oa = run(False, rec)
ob = run(True, rec)
first_bad = next((i + 1 for i, (x, y) in enumerate(zip(oa, ob)) if x != y), 0)
print("MODE", MODE, "WHEN", WHEN, "first deviating step:", first_bad, "of", len(ob))
