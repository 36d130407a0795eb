"""FenheDataset with the reference's constructor and item contract (src/datasets/fenhe_dataset.py:20-218)
and the two split helpers (src/datasets/fenhe_dataset_split.py:20-106).

CPU-side ETL, not accelerated.  The basin files are proprietary and absent, so besides the
reference's file types (``.xlsx`` tables, ``.shp`` outline; pandas+openpyxl / geopandas are imported
lazily and only for those) the station tables may be ``.csv`` and the outline a ``.json`` holding
``[min_lat, max_lat, min_lon, max_lon]`` -- that is what ``write_synthetic_dataset`` produces and
what the tests use.  Parity of the Excel / shapefile branches is unpinned (the libraries are not
installed here); the arithmetic after loading is shared by both branches.

Item: (rain_lr [T,1,H,W] float, dem [1,Hd,Wd] float (min-max normalised), lu one-hot [C,Hl,Wl] float,
s_coords [N,2] long (row, col on the LR grid), s_values [T,N] float).
"""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset, Subset


def _is_leap(year: int) -> bool:
    return (year % 4 == 0 and year % 100 != 0) or year % 400 == 0


def _days(year: int) -> int:
    return 366 if _is_leap(year) else 365


def get_shapefile_extent(shp_path):
    """[min_lat, max_lat, min_lon, max_lon] of the basin outline."""
    if str(shp_path).endswith(".json"):
        with open(shp_path) as fh:
            return [float(v) for v in json.load(fh)]
    import geopandas as gpd          # lazy: only for real shapefiles
    minx, miny, maxx, maxy = gpd.read_file(shp_path).total_bounds
    return [miny, maxy, minx, maxx]


def _read_table(path, **kw):
    import pandas as pd              # lazy
    if str(path).endswith(".csv"):
        kw.pop("usecols", None)
        return pd.read_csv(path)
    return pd.read_excel(path, **kw)


def repair_negative(series: np.ndarray, spatial_mean: np.ndarray) -> np.ndarray:
    """Replace negative readings (fenhe_dataset.py:163-179): mean of the non-negative neighbours in
    time (values already repaired count), else that day's spatial mean, else 0."""
    out = series.copy()
    n = len(out)
    for t in range(n):
        if out[t] >= 0:
            continue
        near = [out[k] for k in (t - 1, t + 1) if 0 <= k < n and out[k] >= 0]
        if near:
            out[t] = np.mean(near)
        elif not np.isnan(spatial_mean[t]):
            out[t] = spatial_mean[t]
        else:
            out[t] = 0.0
    return out


class FenheDataset(Dataset):
    def __init__(self, rain_lr_path, dem_path, lucc_path, rain_meta_path, rain_station_path, shp_path, T=5,
                 start_year=2012, end_year=2021):
        self.rain_lr = np.nan_to_num(np.load(rain_lr_path).astype(np.float32), nan=0.0, posinf=0.0, neginf=0.0)
        self.dem = np.load(dem_path)
        self.dem_norm = (self.dem - self.dem.min()) / (self.dem.max() - self.dem.min() + 1e-7)

        lucc = np.load(lucc_path)
        self.is_multiyear_lucc = lucc.ndim == 3
        if self.is_multiyear_lucc:
            self.lucc_multiyear = lucc.astype(int)
            mapping = Path(lucc_path).parent / "lucc_year_mapping.npy"
            self.lucc_years = np.load(mapping) if mapping.exists() else np.arange(start_year, end_year + 1)
            self.lucc_onehot_list = [self._lucc_to_onehot(self.lucc_multiyear[i]) for i in range(len(self.lucc_years))]
            self._create_date_to_year_mapping(start_year, end_year)
        else:
            self.lucc = lucc.astype(int)
            self.lucc_onehot = self._lucc_to_onehot(self.lucc)

        self.T = T
        self.start_year = start_year
        self.end_year = end_year
        self.grid_extent = get_shapefile_extent(shp_path)
        self.s_coords, self.s_values = self._prepare_stations(rain_meta_path, rain_station_path, start_year, end_year)

    def _create_date_to_year_mapping(self, start_year, end_year):
        idx = []
        for year in range(start_year, end_year + 1):
            hit = np.where(self.lucc_years == year)[0]
            which = hit[0] if len(hit) else int(np.argmin(np.abs(self.lucc_years - year)))
            idx.extend([which] * _days(year))
        self.day_to_year_idx = np.array(idx)

    def _lucc_to_onehot(self, lucc, ignore_index=0):
        labels = np.unique(lucc[lucc != ignore_index])
        onehot = np.zeros((len(labels),) + lucc.shape, dtype=np.float32)
        for i, label in enumerate(labels):
            onehot[i][lucc == label] = 1.0
        return onehot

    def _prepare_stations(self, rain_meta_path, rain_station_path, start_year, end_year):
        meta = _read_table(rain_meta_path, usecols=["F_站号", "经度", "纬度"])
        rain = _read_table(rain_station_path)
        rain = (rain[(rain["year"] >= start_year) & (rain["year"] <= end_year)]
                .sort_values(["year", "month", "day"]).reset_index(drop=True))
        min_lat, max_lat, min_lon, max_lon = self.grid_extent
        n_rows, n_cols = self.rain_lr.shape[-2:]
        readings = rain.drop(columns=["year", "month", "day"], errors="ignore")
        spatial_mean = readings.mask(readings < 0).mean(axis=1).to_numpy()

        coords, columns = [], []
        for _, st in meta.iterrows():
            st_id = int(st["F_站号"])
            lat, lon = st["纬度"], st["经度"]
            r = int((max_lat - lat) / (max_lat - min_lat) * (n_rows - 1))
            c = int((lon - min_lon) / (max_lon - min_lon) * (n_cols - 1))
            if not (0 <= r < n_rows and 0 <= c < n_cols):
                print(f"Station {st_id} ({lat},{lon}) out of bounds, skipped")
                continue
            if str(st_id) not in rain.columns:
                continue
            columns.append(repair_negative(rain[str(st_id)].to_numpy(dtype=np.float32), spatial_mean))
            coords.append([r, c])
        if not columns:
            raise ValueError("No valid stations found within grid bounds!")
        values = np.nan_to_num(np.stack(columns, axis=1).astype(np.float32), nan=0.0, posinf=0.0, neginf=0.0)
        return np.array(coords), values

    def __len__(self):
        return self.rain_lr.shape[0] - self.T

    def __getitem__(self, idx):
        x_lr = torch.FloatTensor(self.rain_lr[idx:idx + self.T, None, ...])
        dem = torch.FloatTensor(self.dem_norm[None, ...])
        if self.is_multiyear_lucc:
            mid = min(idx + self.T // 2, len(self.day_to_year_idx) - 1)
            lu = torch.FloatTensor(self.lucc_onehot_list[self.day_to_year_idx[mid]])
        else:
            lu = torch.FloatTensor(self.lucc_onehot)
        return x_lr, dem, lu, torch.LongTensor(self.s_coords), torch.FloatTensor(self.s_values[idx:idx + self.T])


def split_dataset_by_year(dataset, train_years, val_years, test_years):
    """Year-range subsets; the last T days of each year form no window (fenhe_dataset_split.py:20-67)."""
    def indices(first, last):
        out, day0 = [], 0
        for year in range(dataset.start_year, dataset.end_year + 1):
            if first <= year <= last:
                out.extend(range(day0, day0 + _days(year) - dataset.T))
            day0 += _days(year)
        return out

    parts = [indices(*yr) for yr in (train_years, val_years, test_years)]
    for name, yr, p in zip(("train", "val", "test"), (train_years, val_years, test_years), parts):
        print(f"  {name}: {yr[0]}-{yr[1]}, {len(p)} samples")
    return tuple(Subset(dataset, p) for p in parts)


def split_dataset_random(dataset, train_ratio=0.7, val_ratio=0.15, test_ratio=0.15, seed=42):
    """Seeded random permutation split (fenhe_dataset_split.py:70-106)."""
    assert abs(train_ratio + val_ratio + test_ratio - 1.0) < 1e-6, "ratios must sum to 1"
    np.random.seed(seed)
    order = np.random.permutation(len(dataset))
    n_train, n_val = int(len(dataset) * train_ratio), int(len(dataset) * val_ratio)
    cuts = (order[:n_train], order[n_train:n_train + n_val], order[n_train + n_val:])
    return tuple(Subset(dataset, c.tolist()) for c in cuts)


def write_synthetic_dataset(out_dir, *, start_year=2012, end_year=2013, lr_hw=(15, 12), scale=8, n_classes=5,
                            n_stations=12, seed=0, multiyear_lucc=True):
    """Write a small synthetic basin in the file layout FenheDataset reads; returns the path dict
    (the ``data`` section of a config).  Rain is sparse and non-negative, a few gauge readings are
    negative (to exercise the repair), DEM is smooth, land use has ``n_classes`` labels 1..n."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    H, W = lr_hw
    days = sum(_days(y) for y in range(start_year, end_year + 1))
    rain = (rng.random((days, H, W)) > 0.7) * rng.gamma(2.0, 4.0, (days, H, W))
    np.save(os.path.join(out_dir, "rain_lr.npy"), rain.astype(np.float32))
    yy, xx = np.mgrid[0:H * scale, 0:W * scale]
    np.save(os.path.join(out_dir, "dem.npy"), (800 + 300 * np.sin(yy / 17.0) + 200 * np.cos(xx / 11.0)).astype(np.float32))
    n_years = end_year - start_year + 1
    lucc = rng.integers(1, n_classes + 1, (n_years, H * scale, W * scale) if multiyear_lucc else (H * scale, W * scale))
    np.save(os.path.join(out_dir, "lucc.npy"), lucc)
    extent = [35.0, 39.0, 110.5, 113.5]
    with open(os.path.join(out_dir, "extent.json"), "w") as fh:
        json.dump(extent, fh)
    ids = [50000 + i for i in range(n_stations)]
    lat = rng.uniform(extent[0] + 0.05, extent[1] - 0.05, n_stations)
    lon = rng.uniform(extent[2] + 0.05, extent[3] - 0.05, n_stations)
    pd.DataFrame({"F_站号": ids, "经度": lon, "纬度": lat}).to_csv(os.path.join(out_dir, "meta.csv"), index=False)
    dates = [(y, m, d) for y in range(start_year, end_year + 1)
             for m, nd in zip(range(1, 13), (31, 29 if _is_leap(y) else 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31))
             for d in range(1, nd + 1)]
    table = {"year": [t[0] for t in dates], "month": [t[1] for t in dates], "day": [t[2] for t in dates]}
    for sid in ids:
        series = (rng.random(days) > 0.7) * rng.gamma(2.0, 5.0, days)
        series[rng.random(days) < 0.01] = -9999.0
        table[str(sid)] = series.astype(np.float32)
    pd.DataFrame(table).to_csv(os.path.join(out_dir, "rain.csv"), index=False)
    return {"rain_lr_path": os.path.join(out_dir, "rain_lr.npy"), "dem_path": os.path.join(out_dir, "dem.npy"),
            "lucc_path": os.path.join(out_dir, "lucc.npy"), "meta_path": os.path.join(out_dir, "meta.csv"),
            "rain_excel_path": os.path.join(out_dir, "rain.csv"), "shp_path": os.path.join(out_dir, "extent.json"),
            "start_year": start_year, "end_year": end_year}
