#!/usr/bin/env python3
"""Entry point with the reference's CLI (main.py:11-24): ``python main.py --config <name>``.
Under ``python -m torch.distributed.run --nproc-per-node N main.py ...`` it trains data-parallel
(one process per GPU, RCCL gradient all-reduce)."""
import argparse
import os

import torch


def main():
    ap = argparse.ArgumentParser(description="ConvLSTM precipitation downscaling on the MI355X engine")
    ap.add_argument("--config", type=str, default="default", help="name of configs/<name>.yaml")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from src.training.trainer import Trainer
    trainer = Trainer(config_name=args.config)
    trainer.config.validate()
    trainer.train()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
