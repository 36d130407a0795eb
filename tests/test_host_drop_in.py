"""Config / EarlyStopping / dataset drop-ins (CPU) and the Trainer end to end on a synthetic basin (GPU)."""
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import ROOT


def test_config_schema_and_validation(tmp_path):
    from src.config import Config, load_config
    cfg = load_config("default")
    assert cfg.model.hidden_dims == [16, 32] and cfg.model.T == 5 and cfg.model.scale_factor == 8
    assert cfg.training.epochs == 500 and cfg.training.learning_rate == 0.0005
    assert cfg.training.scheduler_patience == 10 and cfg.training.scheduler_factor == 0.7
    cfg.validate()                                      # missing data files only warn
    out = tmp_path / "sub" / "c.yaml"
    cfg.to_yaml(str(out))
    again = Config.from_yaml(str(out))
    assert again == cfg
    with pytest.raises(FileNotFoundError):
        Config.from_yaml(str(tmp_path / "nope.yaml"))
    bad = yaml.safe_load(open(out))
    bad["model"]["unknown_key"] = 1
    (tmp_path / "bad.yaml").write_text(yaml.dump(bad))
    with pytest.raises(TypeError):
        Config.from_yaml(str(tmp_path / "bad.yaml"))
    cfg.training.learning_rate = 0
    with pytest.raises(ValueError):
        cfg.validate()


def test_early_stopping_semantics():
    from src.utils.early_stopping import EarlyStopping
    es = EarlyStopping(patience=2, min_delta=0.1, mode="min", verbose=False)
    assert es(1.0, 0) is True and es.counter == 0
    assert es(0.95, 1) is False and es.counter == 1          # improvement below min_delta
    assert es(0.8, 2) is True and es.counter == 0 and es.best_epoch == 2
    assert es(0.9, 3) is False and not es.early_stop
    assert es(0.9, 4) is False and es.early_stop
    es.reset()
    assert es.best_score is None and not es.early_stop


def _make_dataset(tmp_path, **kw):
    from pl_convlstm_gan_amd.host.datasets import FenheDataset, write_synthetic_dataset
    paths = write_synthetic_dataset(str(tmp_path / "basin"), **kw)
    ds = FenheDataset(paths["rain_lr_path"], paths["dem_path"], paths["lucc_path"], paths["meta_path"],
                      paths["rain_excel_path"], paths["shp_path"], T=5, start_year=paths["start_year"],
                      end_year=paths["end_year"])
    return ds, paths


def test_dataset_item_contract_and_split(tmp_path):
    from src.datasets.fenhe_dataset_split import split_dataset_by_year, split_dataset_random
    ds, _ = _make_dataset(tmp_path, start_year=2012, end_year=2013, lr_hw=(15, 12), scale=8)
    assert len(ds) == 366 + 365 - 5
    x, dem, lu, coords, vals = ds[3]
    assert x.shape == (5, 1, 15, 12) and x.dtype == torch.float32
    assert dem.shape == (1, 120, 96) and 0.0 <= float(dem.min()) and float(dem.max()) <= 1.0
    assert lu.shape == (5, 120, 96) and torch.all(lu.sum(0) == 1)
    assert coords.dtype == torch.int64 and coords.shape[1] == 2 and vals.shape == (5, coords.shape[0])
    assert float(ds.s_values.min()) >= 0.0                    # negative gauge readings were repaired
    assert int(coords[:, 0].max()) < 15 and int(coords[:, 1].max()) < 12
    tr, va, te = split_dataset_by_year(ds, (2012, 2012), (2013, 2013), (2013, 2013))
    assert len(tr) == 366 - 5 and len(va) == 365 - 5 and len(te) == 365 - 5
    a, b, c = split_dataset_random(ds)
    assert len(a) + len(b) + len(c) == len(ds) and len(a) == int(len(ds) * 0.7)


def test_negative_repair_rule():
    from pl_convlstm_gan_amd.host.datasets import repair_negative
    s = np.array([-1, 2, -1, 4, -1, -1, 3], dtype=np.float32)
    m = np.array([7, 7, 7, 7, 7, np.nan, 7], dtype=np.float32)
    out = repair_negative(s, m)
    # t=0: only right neighbour (2); t=2: mean(2,4); t=4: left 4 only (right is negative); t=5: left is the repaired 4, right 3
    assert np.allclose(out, [2, 2, 3, 4, 4, 3.5, 3])


def test_station_rmse_against_the_restatement():
    """Trainer.compute_station_rmse (trainer.py:225-272) in its no-host-wait form equals the plain restatement:
    gauges outside the grid dropped, unobserved (NaN) values skipped, 0 when nothing is observed."""
    from oracle import cpu_oracle as O
    from pl_convlstm_gan_amd.host.trainer import Trainer
    g = torch.Generator().manual_seed(4)
    pred = torch.rand(3, 4, 1, 24, 20, generator=g)
    coords = torch.tensor([[0, 0], [11, 9], [5, 3], [12, 2], [3, 10], [7, 7]])          # LR cells; x2: (12, .) and (., 10) fall outside
    coords = coords.unsqueeze(0).repeat(3, 1, 1)
    values = 20 * torch.rand(3, 4, 6, generator=g)
    values[0, 1, 2] = float("nan")
    values[2, :, 5] = float("nan")
    rmse, at, truth, inside, seen = Trainer.compute_station_rmse(None, pred, coords, values, 2.0)
    assert inside.tolist() == [True, True, True, False, False, True]
    assert at.shape == truth.shape == seen.shape == (3, 4, 4) and int(seen.sum()) == 3 * 4 * 4 - 1 - 4
    assert torch.equal(at[:, :, 1], pred[:, :, 0, 22, 18]) and torch.equal(truth[:, :, 3].isnan(), values[:, :, 5].isnan())
    assert abs(float(rmse) - O.station_rmse(pred, coords, values, 2.0)) <= 1e-6 * float(rmse)
    # the batch's host copy of the coordinates gives the same thing
    assert float(Trainer.compute_station_rmse(None, pred, coords, values, 2.0, coords_host=coords.clone())[0]) == float(rmse)
    # nothing observed / nothing inside
    assert float(Trainer.compute_station_rmse(None, pred, coords, torch.full_like(values, float("nan")), 2.0)[0]) == 0.0
    out = Trainer.compute_station_rmse(None, pred, coords + 100, values, 2.0)
    assert float(out[0]) == 0.0 and out[1] is None and not bool(out[3].any())


@pytest.mark.gpu
def test_trainer_end_to_end_on_synthetic_basin(gpu, tmp_path, monkeypatch):
    from pl_convlstm_gan_amd.host import config as C
    from pl_convlstm_gan_amd.host.datasets import write_synthetic_dataset
    paths = write_synthetic_dataset(str(tmp_path / "basin"), start_year=2012, end_year=2013, lr_hw=(15, 12), scale=2,
                                    n_stations=10)
    cfg = C.Config()
    for k, v in paths.items():
        setattr(cfg.data, k, v)
    cfg.model.scale_factor = 2
    cfg.training.batch_size = 16
    cfg.training.epochs = 2
    cfg.training.train_years, cfg.training.val_years, cfg.training.test_years = [2012, 2012], [2013, 2013], [2013, 2013]
    cfg.output.output_dir = str(tmp_path / "out")
    cfg.output.log_interval = 100
    cfg_dir = tmp_path / "configs"
    cfg.to_yaml(str(cfg_dir / "synthetic.yaml"))
    monkeypatch.setattr(C, "CONFIG_DIR", str(cfg_dir))
    from src.training.trainer import Trainer
    t = Trainer("synthetic")
    t.config.validate()
    t.train()
    assert t.history["epoch"] == [0, 1]
    assert all(np.isfinite(t.history["total_loss"])) and all(np.isfinite(t.history["rmse"]))
    assert t.history["total_loss"][1] < t.history["total_loss"][0]          # it learns
    ck = torch.load(os.path.join(t.output_dir, "best_model.pth"), weights_only=True)   # nothing in the file is executed
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "rmse", "history"}
    assert "upsample_blocks.0.conv.weight" in ck["model_state_dict"]
    # the lazily created block is not in the optimizer (reference behaviour): 18 tensors, not 20
    assert len(ck["optimizer_state_dict"]["param_groups"][0]["params"]) == 18
    t2 = Trainer("synthetic")
    t2.setup_model(t2.setup_data())
    assert t2.resume(os.path.join(t.output_dir, "best_model.pth")) == ck["epoch"] + 1


@pytest.mark.gpu
def test_trainer_with_graph_replayed_steps_matches_the_eager_trainer(gpu, tmp_path, monkeypatch):
    """PLC_GRAPH_STEP=1: the Trainer replays the step as a HIP graph (one per batch shape: the last batch of an
    epoch is shorter) between its own host work -- station RMSE through indexing kernels, validation forwards,
    checkpoints with .cpu() copies, a scheduler.  Same seeded batch order in both runs: the epoch history and the
    final parameters are bit-equal to the eager Trainer's."""
    from pl_convlstm_gan_amd.host import config as C
    from pl_convlstm_gan_amd.host.datasets import write_synthetic_dataset
    paths = write_synthetic_dataset(str(tmp_path / "basin"), start_year=2012, end_year=2013, lr_hw=(15, 12), scale=2,
                                    n_stations=10)
    cfg = C.Config()
    for k, v in paths.items():
        setattr(cfg.data, k, v)
    cfg.model.scale_factor = 2
    cfg.training.batch_size = 16
    cfg.training.epochs = 2
    cfg.training.train_years, cfg.training.val_years, cfg.training.test_years = [2012, 2012], [2013, 2013], [2013, 2013]
    cfg.output.log_interval = 100
    cfg_dir = tmp_path / "configs"
    monkeypatch.setattr(C, "CONFIG_DIR", str(cfg_dir))
    monkeypatch.setenv("PLC_SHUFFLE_SEED", "5")           # the same batch order in both runs
    from src.training.trainer import Trainer
    runs = {}
    for tag, flag in (("eager", "0"), ("graph", "1")):
        cfg.output.output_dir = str(tmp_path / f"out_{tag}")
        cfg.to_yaml(str(cfg_dir / "synthetic.yaml"))
        monkeypatch.setenv("PLC_GRAPH_STEP", flag)
        torch.manual_seed(11)                              # the same initial weights (and lazy block) in both runs
        t = Trainer("synthetic")
        assert t.graph_steps == (flag == "1")
        t.train()
        if flag == "1":
            assert len(t._graphed) == 2                    # full batches and the shorter last one
        runs[tag] = (t.history, {k: v.detach().cpu() for k, v in t.model.state_dict().items()})
    assert runs["eager"][0] == runs["graph"][0]
    for k, v in runs["eager"][1].items():
        assert torch.equal(v, runs["graph"][1][k]), k
