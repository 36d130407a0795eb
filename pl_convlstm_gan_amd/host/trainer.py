"""Trainer with the reference's interface (src/training/trainer.py:16-449) on the HIP engine.

Same constructor, attributes (config, device, output_dir, history, best_rmse, best_epoch,
early_stopping, model, optimizer, scheduler, loss_module, train_loader / val_loader / test_loader /
dataloader) and methods (setup_data, setup_model, validate, compute_station_rmse, train_epoch,
train).  Additions: batch data parallelism when launched under torchrun (each rank loads and trains on
its shard of every batch; one flat-gradient all-reduce per step, see training.py; validation batches
are dealt out over the ranks), and ``resume(path)`` (the reference only saves).

Kept from the reference on purpose: the optimizer is built before the generator's lazy upsample
blocks exist, so those blocks receive gradients and count in the clip norm but are not updated
(set ``optimize_lazy_blocks = True`` before ``train()`` to register them after the first forward);
ReduceLROnPlateau steps on the validation RMSE; only the best model is saved.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from ..losses import CombinedLoss
from ..models import Generator
from ..training import train_step
from .config import load_config
from .datasets import FenheDataset, split_dataset_by_year, split_dataset_random
from .early_stopping import EarlyStopping

HISTORY_KEYS = ("epoch", "total_loss", "point_loss", "conserve_loss", "smooth_loss", "temporal_loss", "rmse",
                "learning_rate")


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's slice of a batch of n samples, or None when the batch has fewer samples than
    ranks.  Every rank sees the same n, so all take the same decision without talking: a short batch is
    skipped by ALL ranks (a rank with an empty slice would otherwise miss the gradient all-reduce the
    others are waiting in)."""
    if n < world:
        return None
    return rank * n // world, (rank + 1) * n // world


_STATION_INDEX_CACHE = {}      # compute_station_rmse: (gauge coordinates, grid, scale, device) -> device index tensors


class ShardedBatchSampler:
    """Data parallelism at the loader: wraps a batch sampler and yields, on every rank, only this rank's slice
    of each batch's indices (shard_bounds), so a rank loads and copies to its GPU just the samples it trains on.
    Batches with fewer samples than ranks are dropped by all ranks alike.  A shuffling base sampler must draw
    the same permutation on every rank: give it a generator with the same seed everywhere."""

    def __init__(self, batch_sampler, rank, world):
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world

    def __iter__(self):
        for idx in self.batch_sampler:
            bounds = shard_bounds(len(idx), self.rank, self.world)
            if bounds is not None:
                yield idx[bounds[0]:bounds[1]]

    def __len__(self):
        return len(self.batch_sampler)


def sum_over_ranks(values):
    """Element-wise sum of a short list of host numbers over the ranks (identity for one process)."""
    rank, world = _rank_world()
    if world == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]


def mean_over_ranks(x):
    """Mean of a host scalar over the ranks (identity for one process): what feeds the LR scheduler,
    early stopping and the best-model decision must be the same number on every rank."""
    rank, world = _rank_world()
    if world == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / world


class Trainer:
    def __init__(self, config_name="default"):
        self.config = load_config(config_name)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.output_dir = self.config.output.output_dir
        os.makedirs(self.output_dir, exist_ok=True)
        self.history = {k: [] for k in HISTORY_KEYS}
        self.best_rmse = float("inf")
        self.best_epoch = -1
        self.optimize_lazy_blocks = False
        # PLC_GRAPH_STEP=1: replay the training step as a HIP graph (training.GraphedTrainStep) from the second batch
        # on, one captured graph per batch shape.  It pays where a step is launch-bound (the 15 x 12 basin grid of
        # default.yaml: about a fifth off the step) and costs a few per cent at 256 x 256; single process only.
        self.graph_steps = os.environ.get("PLC_GRAPH_STEP", "") not in ("", "0")
        self._graphed = {}
        self._eager_steps = 0
        self.early_stopping = None
        tr = self.config.training
        if getattr(tr, "use_early_stopping", False):
            self.early_stopping = EarlyStopping(patience=getattr(tr, "early_stopping_patience", 20),
                                                min_delta=getattr(tr, "early_stopping_min_delta", 0.0001),
                                                mode="min", verbose=True)

    # ------------------------------------------------------------------ data
    def setup_data(self):
        d, tr = self.config.data, self.config.training
        full = FenheDataset(rain_lr_path=d.rain_lr_path, dem_path=d.dem_path, lucc_path=d.lucc_path,
                            rain_meta_path=d.meta_path, rain_station_path=d.rain_excel_path, shp_path=d.shp_path,
                            T=self.config.model.T, start_year=d.start_year, end_year=d.end_year)
        if tr.use_split:
            if tr.split_method == "year":
                parts = split_dataset_by_year(full, tuple(tr.train_years), tuple(tr.val_years), tuple(tr.test_years))
            else:
                parts = split_dataset_random(full, 0.7, 0.15, 0.15)
            self.train_loader = self._train_loader(parts[0], tr.batch_size)
            self.val_loader = DataLoader(parts[1], batch_size=tr.batch_size, shuffle=False)
            self.test_loader = DataLoader(parts[2], batch_size=tr.batch_size, shuffle=False)
        else:
            self.train_loader = self._train_loader(full, tr.batch_size)
            self.val_loader = self.test_loader = None
        self.dataloader = self.train_loader
        return full

    def _train_loader(self, dataset, batch_size):
        """The reference's shuffled loader (trainer.py:94-108).  Under data parallelism every rank walks the same
        shuffled batches (one seed for all ranks) but loads only its own slice of each."""
        rank, world = _rank_world()
        seed = os.environ.get("PLC_SHUFFLE_SEED")
        if world == 1 and seed is None:
            return DataLoader(dataset, batch_size=batch_size, shuffle=True)     # unseeded, as the reference
        from torch.utils.data import BatchSampler, RandomSampler
        order = RandomSampler(dataset, generator=torch.Generator().manual_seed(int(seed or "0")))
        return DataLoader(dataset, batch_sampler=ShardedBatchSampler(BatchSampler(order, batch_size, drop_last=False), rank, world))

    # ------------------------------------------------------------------ model
    def setup_model(self, dataset):
        onehot = dataset.lucc_onehot_list[0] if getattr(dataset, "is_multiyear_lucc", False) else dataset.lucc_onehot
        kw = {"hidden_dims": self.config.model.hidden_dims, "lu_channels": onehot.shape[0]}
        mc = self.config.model
        if getattr(mc, "target_grid_size", None):
            kw["target_grid_size"] = tuple(mc.target_grid_size)
        elif getattr(mc, "scale_factor", None):
            kw["scale_factor"] = mc.scale_factor
        rank, world = _rank_world()
        if world > 1:
            torch.manual_seed(0)      # identical initial weights on every rank
        self.model = self._make_model(**kw).to(self.device)
        tr = self.config.training
        self.optimizer = self._make_optimizer(self.model.parameters(), tr.learning_rate)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=tr.scheduler_factor,
                                                                    patience=tr.scheduler_patience)
        self.loss_module = CombinedLoss(lambda_point=tr.lambda_point, lambda_conserve=tr.lambda_conserve,
                                        lambda_smooth=tr.lambda_smooth, lambda_temporal=tr.lambda_temporal,
                                        use_weighted_loss=getattr(tr, "use_weighted_loss", True),
                                        weight_strategy=getattr(tr, "weight_strategy", "log"))

    # (the two objects a test without a GPU replaces by CPU stand-ins)
    def _make_model(self, **kw):
        return Generator(**kw)

    def _make_optimizer(self, params, lr):
        # Adam as the reference (trainer.py:153-158); FusedAdam keeps torch.optim.Adam's state_dict layout
        from ..optim import FusedAdam
        return FusedAdam(params, lr=lr)

    def _forward_kwargs(self):
        igs = getattr(self.config.model, "input_grid_size", None)
        return {"input_grid_size": tuple(igs)} if igs else {}

    def _to_device(self, batch):
        return tuple(t.to(self.device) for t in batch)

    # ------------------------------------------------------------------ evaluation
    def compute_station_rmse(self, fake_hr, s_coords, s_values, scale_factor=1.0, coords_host=None):
        """(rmse, pred_at_stations, true_vals, valid_station_mask, mask): trainer.py:225-272.

        Same quantities as the reference, arranged so that the host does not wait for the GPU: the gauge pixels are
        a few dozen integers, so their index arithmetic runs on the host (``coords_host``: the batch's s_coords
        before it went to the device, when the caller still has it; otherwise one small copy back) and the
        gathers take index tensors instead of boolean masks (a mask makes PyTorch count its set bits on the host
        first).  The mean over the observed values is a masked sum over a count, decided on the device."""
        B, T, _, H, W = fake_hr.shape
        coords = s_coords[0] if s_coords.dim() == 3 else s_coords
        if coords_host is None:
            coords_host = coords
        ch = coords_host[0] if coords_host.dim() == 3 else coords_host
        ch = ch.detach().cpu()
        # The gauges are the same in every batch: their device-side index tensors are kept, keyed by the coordinates
        # themselves, so that a batch uploads nothing here (an upload from pageable memory makes the host wait for
        # everything queued on the stream, i.e. for the training step just launched).
        key = (ch.numpy().tobytes(), tuple(ch.shape), H, W, float(scale_factor), str(fake_hr.device))
        cache = _STATION_INDEX_CACHE
        if key not in cache:
            if len(cache) >= 16:
                cache.clear()
            hr = ((ch.float() + 0.5) * scale_factor - 0.5).long()                    # gauge pixel, as combined_loss.py:79-141
            rows, cols = hr[:, 0], hr[:, 1]
            inside_h = (rows >= 0) & (rows < H) & (cols >= 0) & (cols < W)
            keep = torch.nonzero(inside_h).flatten()
            cache[key] = (inside_h.to(fake_hr.device), bool(inside_h.any()),
                          torch.stack([rows[keep], cols[keep], keep]).to(fake_hr.device))
        inside, any_inside, idx = cache[key]
        zero = fake_hr.new_zeros(())
        if not any_inside:
            return zero, None, None, inside, None
        at = fake_hr[:, :, 0][:, :, idx[0], idx[1]]
        truth = s_values.index_select(2, idx[2])
        seen = ~torch.isnan(truth)
        diff = torch.where(seen, at - truth, zero)
        count = seen.sum()
        rmse = torch.where(count > 0, torch.sqrt((diff * diff).sum() / count.clamp(min=1)), zero)
        return rmse, at, truth, inside, seen

    def validate(self):
        """trainer.py:182-223: mean over the validation batches of the batch loss and the batch RMSE.  Under data
        parallelism the BATCHES are dealt out over the ranks (rank r evaluates batches r, r + world, ...: every
        batch is evaluated whole, exactly as a single process would) and the sums and counts are added over
        the ranks, so every rank gets the same two means."""
        if self.val_loader is None:
            return None
        rank, world = _rank_world()
        self.model.eval()
        losses, rmses = [], []
        with torch.no_grad():
            for k, batch in enumerate(self.val_loader):
                if k % world != rank:
                    continue
                lr, dem, lu, s_coords, s_values = self._to_device(batch)
                fake_hr = self.model(lr, dem, lu, **self._forward_kwargs())
                sf = fake_hr.shape[-2] / lr.shape[-2]
                loss, _ = self.loss_module(fake_hr, lr, s_coords, s_values, sf)
                losses.append(loss.detach().float().reshape(()))
                rmses.append(self.compute_station_rmse(fake_hr, s_coords, s_values, sf, coords_host=batch[3])[0].float().reshape(()))
            # (one read for the whole pass instead of two host waits per batch)
            losses = torch.stack(losses).tolist() if losses else []
            rmses = torch.stack(rmses).tolist() if rmses else []
        self.model.train()
        loss_sum, rmse_sum, n = sum_over_ranks([float(np.sum(losses)), float(np.sum(rmses)), len(losses)])
        return {"loss": loss_sum / n, "rmse": rmse_sum / n} if n else {"loss": float("nan"), "rmse": float("nan")}

    # ------------------------------------------------------------------ training
    def train_epoch(self, epoch, T):
        self.model.train()
        rmses = []
        sums = {k: [] for k in ("total", "point", "conserve", "smooth", "temporal")}
        last = (None, None, None, None)
        # The step is queued without waiting for its loss, so the host fetches the NEXT batch from the loader while the
        # GPU works on this one, and reads this one's loss afterwards (same batches in the same order).
        batches = iter(self.train_loader)
        upcoming = next(batches, None)
        i = -1
        while upcoming is not None:
            batch, i = upcoming, i + 1
            # (under data parallelism the loader already yields this rank's slice of the batch, and drops batches
            # with fewer samples than ranks on every rank alike: ShardedBatchSampler)
            lr, dem, lu, s_coords, s_values = self._to_device(batch)
            step = self._graphed_step((lr, dem, lu, s_coords, s_values))
            if step is not None:
                loss, parts, fake_hr, skipped = step(lr, dem, lu, s_coords, s_values)
            else:
                loss, parts, fake_hr, skipped = train_step(
                    self.model, self.loss_module, self.optimizer, lr, dem, lu, s_coords, s_values,
                    max_norm=self.config.training.grad_clip_norm, forward_kwargs=self._forward_kwargs(),
                    nan_check="device")     # the skip is decided by the optimizer kernel; `skipped` is read below, after the step is queued
                self._eager_steps += 1
                if self.graph_steps:
                    # nothing of this step's autograd graph may outlive it: a capture must not find the
                    # parameters' gradient-accumulation nodes bound to this step's stream
                    loss, fake_hr = loss.detach(), fake_hr.detach()
                    parts = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
            if self.optimize_lazy_blocks and self.model.upsample_blocks is not None and len(self.optimizer.param_groups) == 1 \
                    and len(self.model.upsample_blocks) > 0:
                self.optimizer.add_param_group({"params": list(self.model.upsample_blocks.parameters())})
            # the station RMSE of this batch is queued behind the step (nothing in it waits for the GPU) ...
            with torch.no_grad():
                sf = fake_hr.shape[-2] / lr.shape[-2]
                rmse, at, truth, inside, _ = self.compute_station_rmse(fake_hr.detach(), s_coords, s_values, sf,
                                                                       coords_host=batch[3])
                # ... and everything the log needs comes back in ONE read, after the next batch has been fetched
                scalars = torch.stack([loss.detach().float().reshape(())]
                                      + [torch.as_tensor(parts[k], dtype=torch.float32, device=loss.device).detach().reshape(())
                                         for k in ("point", "conserve", "smooth", "temporal")]
                                      + [rmse.float().reshape(())])
            upcoming = next(batches, None)
            if skipped:
                print("NaN detected, skip this batch")
                continue
            total, point, conserve, smooth, temporal, batch_rmse = scalars.tolist()
            for k, v in zip(("total", "point", "conserve", "smooth", "temporal"), (total, point, conserve, smooth, temporal)):
                sums[k].append(v)
            rmses.append(batch_rmse)
            last = (at, truth, s_coords, inside)
            if i % self.config.output.log_interval == 0 and _rank_world()[0] == 0:
                print(f"Epoch {epoch} | Loss: {total:.4f} | Point: {point:.4f} | Conserve: {conserve:.4f} | "
                      f"Smooth: {smooth:.4f} | Temporal: {temporal:.4f} | Batch RMSE: {batch_rmse:.4f}")
        avg = {k: (float(np.mean(v)) if v else 0.0) for k, v in sums.items()}
        return rmses, last[0], last[1], last[2], last[3], avg

    def _graphed_step(self, batch):
        """The captured step for this batch's shape, or None when the step runs eagerly: the switch is off, more
        than one process trains, or this is the first step (which creates the lazy upsample blocks and, with
        ``optimize_lazy_blocks``, changes what the optimizer owns)."""
        if not self.graph_steps or _rank_world()[1] > 1 or self._eager_steps == 0 or not batch[0].is_cuda:
            return None
        key = tuple(tuple(t.shape) for t in batch)
        if key not in self._graphed:
            from ..training import GraphedTrainStep
            self._graphed[key] = GraphedTrainStep(self.model, self.loss_module, self.optimizer, batch,
                                                  max_norm=self.config.training.grad_clip_norm,
                                                  forward_kwargs=self._forward_kwargs())
        return self._graphed[key]

    def _checkpoint(self, epoch, rmse):
        path = os.path.join(self.output_dir, "best_model.pth")
        if os.path.exists(path):
            os.remove(path)
        # plain Python numbers only, so the file loads with torch.load(weights_only=True)
        history = {k: [float(x) if isinstance(x, (float, np.floating)) else int(x) for x in v]
                   for k, v in self.history.items()}
        sched = {k: (float(v) if isinstance(v, np.floating) else v) for k, v in self.scheduler.state_dict().items()}
        torch.save({"epoch": int(epoch), "model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict(),
                    "scheduler_state_dict": sched, "rmse": float(rmse), "history": history}, path)
        print(f"New best model saved: epoch {epoch + 1}, RMSE {rmse:.4f}")

    def resume(self, path):
        """Load a checkpoint written by ``train`` (not in the reference).  Needs setup_model first; the
        lazy upsample blocks are materialised from the checkpoint's keys."""
        from ..models import UpsampleBlock
        # weights_only: nothing in the file is executed (a checkpoint may come from a third party)
        ck = torch.load(path, map_location=self.device, weights_only=True)
        n_blocks = len({k.split(".")[1] for k in ck["model_state_dict"] if k.startswith("upsample_blocks.")})
        if self.model.upsample_blocks is None:
            self.model.upsample_blocks = torch.nn.ModuleList(
                [UpsampleBlock(self.model.hidden_dims[1], 2) for _ in range(n_blocks)]).to(self.device)
        self.model.load_state_dict(ck["model_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        self.history = ck["history"]
        self.best_rmse, self.best_epoch = ck["rmse"], ck["epoch"]
        return ck["epoch"] + 1

    def train(self, start_epoch=0):
        if not hasattr(self, "model"):
            dataset = self.setup_data()
            self.setup_model(dataset)
        T = self.config.model.T
        rank = _rank_world()[0]
        for epoch in range(start_epoch, self.config.training.epochs):
            rmses, at, truth, s_coords, inside, avg = self.train_epoch(epoch, T)
            # every rank trained on its own shards: the epoch figures that feed the scheduler, early
            # stopping and the best-model decision are averaged over ranks so that all ranks decide alike
            train_rmse = mean_over_ranks(float(np.mean(rmses)) if rmses else float("nan"))
            self.history["epoch"].append(epoch)
            for k in ("total", "point", "conserve", "smooth", "temporal"):
                self.history[f"{k}_loss"].append(mean_over_ranks(avg[k]))
            self.history["rmse"].append(train_rmse)
            self.history["learning_rate"].append(self.optimizer.param_groups[0]["lr"])
            val = self.validate()
            current = float(val["rmse"]) if val else train_rmse     # (the same number on every rank: see validate)
            if rank == 0:
                print(f"Epoch {epoch} | Train RMSE: {train_rmse:.4f}" +
                      (f" | Val Loss: {val['loss']:.4f} | Val RMSE: {val['rmse']:.4f}" if val else ""))
            self.scheduler.step(current)
            if (epoch + 1) % 10 == 0 and rank == 0:
                self._plot_curves("training_curves.png")
            if self.early_stopping is not None:
                is_best = self.early_stopping(current, epoch)
            else:
                is_best = current < self.best_rmse
            if is_best:
                self.best_rmse, self.best_epoch = current, epoch
                if rank == 0:
                    self._checkpoint(epoch, current)
            if self.early_stopping is not None and self.early_stopping.early_stop:
                print(f"Early stop at epoch {epoch + 1}; best epoch {self.best_epoch + 1}, RMSE {self.best_rmse:.4f}")
                break
        if rank == 0:
            self._plot_curves("final_training_curves.png")
            print(f"Training completed. Best model: epoch {self.best_epoch + 1}, RMSE {self.best_rmse:.4f}; "
                  f"results in {self.output_dir}/")

    def _plot_curves(self, name):
        try:
            from .visualization import plot_training_curves
            plot_training_curves(self.history, save_path=os.path.join(self.output_dir, name))
        except Exception as exc:   # plotting is optional (matplotlib may be missing or headless)
            print(f"(plot skipped: {exc})")
