"""YAML configuration with the reference's schema (src/config.py:7-132, configs/default.yaml).

Four sections -> four dataclasses; an unknown key raises TypeError from the dataclass
constructor, exactly as in the reference.  ``load_config(name)`` resolves
``<repo>/configs/<name>.yaml`` and falls back to the ``CONFIG_NAME`` environment variable.
"""
from __future__ import annotations

import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional

import yaml

CONFIG_DIR = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "configs"))


@dataclass
class DataConfig:
    rain_lr_path: str = ""
    dem_path: str = ""
    lucc_path: str = ""
    meta_path: str = ""
    rain_excel_path: str = ""
    shp_path: str = ""
    start_year: int = 2012
    end_year: int = 2021


@dataclass
class ModelConfig:
    hidden_dims: List[int] = field(default_factory=lambda: [16, 32])
    T: int = 5
    scale_factor: int = 8
    target_grid_size: Optional[List[int]] = None
    input_grid_size: Optional[List[int]] = None


@dataclass
class TrainingConfig:
    batch_size: int = 8
    epochs: int = 50
    learning_rate: float = 0.001
    lambda_point: float = 1.0
    lambda_conserve: float = 1.0
    lambda_smooth: float = 0.1
    lambda_temporal: float = 0.05
    grad_clip_norm: float = 0.5
    scheduler_patience: int = 5
    scheduler_factor: float = 0.5
    use_weighted_loss: bool = True
    weight_strategy: str = "log"
    use_early_stopping: bool = True
    early_stopping_patience: int = 20
    early_stopping_min_delta: float = 0.0001
    use_split: bool = True
    split_method: str = "year"
    train_years: List[int] = field(default_factory=lambda: [2012, 2018])
    val_years: List[int] = field(default_factory=lambda: [2019, 2020])
    test_years: List[int] = field(default_factory=lambda: [2021, 2021])


@dataclass
class OutputConfig:
    output_dir: str = "output"
    log_interval: int = 10
    save_model_interval: int = 10   # parsed but unused, as in the reference
    plot_dpi: int = 300             # parsed but unused, as in the reference


_SECTIONS = (("data", DataConfig), ("model", ModelConfig), ("training", TrainingConfig), ("output", OutputConfig))


@dataclass
class Config:
    data: DataConfig = field(default_factory=DataConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
    output: OutputConfig = field(default_factory=OutputConfig)

    @classmethod
    def from_yaml(cls, yaml_path: str) -> "Config":
        if not os.path.exists(yaml_path):
            raise FileNotFoundError(f"Config file not found: {yaml_path}")
        with open(yaml_path, "r", encoding="utf-8") as fh:
            raw = yaml.safe_load(fh) or {}
        return cls(**{name: kind(**raw.get(name, {})) for name, kind in _SECTIONS})

    def to_yaml(self, yaml_path: str):
        parent = os.path.dirname(yaml_path)
        if parent:
            os.makedirs(parent, exist_ok=True)
        with open(yaml_path, "w", encoding="utf-8") as fh:
            yaml.dump({name: asdict(getattr(self, name)) for name, _ in _SECTIONS}, fh,
                      default_flow_style=False, allow_unicode=True)

    def validate(self):
        for key, value in asdict(self.data).items():
            if key.endswith("_path") and value and not os.path.exists(value):
                print(f"Warning: {key} does not exist: {value}")
        if self.training.learning_rate <= 0:
            raise ValueError("Learning rate must be positive")
        if self.training.batch_size <= 0:
            raise ValueError("Batch size must be positive")
        if self.model.T <= 0:
            raise ValueError("Time window T must be positive")
        if self.training.epochs <= 0:
            raise ValueError("Epochs must be positive")


def load_config(config_name: str = "default") -> Config:
    path = os.path.join(CONFIG_DIR, f"{config_name}.yaml")
    if not os.path.exists(path):
        path = os.path.join(CONFIG_DIR, f"{os.getenv('CONFIG_NAME', 'default')}.yaml")
    return Config.from_yaml(path)
