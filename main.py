#!/usr/bin/env python3
"""Entry point with the reference's CLI (main.py:11-24): ``python main.py --config <name>``.
Under ``python -m torch.distributed.run --nproc-per-node N main.py ...`` it trains data-parallel
(one process per GPU, RCCL gradient all-reduce)."""
import argparse
import datetime
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
        # a collective that does not complete aborts the ranks instead of hanging (PLC_PG_TIMEOUT seconds, default 600)
        patience = datetime.timedelta(seconds=int(os.environ.get("PLC_PG_TIMEOUT", "600")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=patience)
    from src.training.trainer import Trainer
    trainer = Trainer(config_name=args.config)
    trainer.config.validate()
    trainer.train()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
