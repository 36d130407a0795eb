#!/usr/bin/env python3
"""Benchmark of the ConvLSTM rollout hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--config c3|c2|c1|c4|c5|p2|c0]
                    [--global-batch G | --shard]

With --gpus N > 1 and no torchrun environment the script starts the N ranks itself (one process per
GPU, `python -m torch.distributed.run --nproc-per-node N`) before anything touches a GPU, and fails
at once, with a non-zero exit code, when the box has fewer than N GPUs or when the number of ranks it
finds itself in disagrees with --gpus.

One "step" = one pass of the hot path over one batch of synthetic, HBM-resident input:
  train (default): zero_grad -> Generator rollout forward -> CombinedLoss -> backward (BPTT)
                   -> gradient all-reduce (N > 1) -> clip_grad_norm(0.5) -> Adam step
                   (the reference's step, src/training/trainer.py:290-315)
  fwd:             the rollout forward under no_grad (inference)
The default workload is BASELINE.json configs[2], the configuration the metric is quoted on: 256x256
frames, T=10, GLOBAL batch 32, scale_factor=1.  The batch is sharded over the ranks (32 / N per GPU: the
whole batch on one GPU at N=1, the 4-per-GPU shard at N=8), the only collective is the gradient
all-reduce, total work is fixed (strong scaling).  --shard runs the configuration's per-GPU shard of its
own GPU count on every rank instead (c3: B=4 per GPU whatever N is; weak scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (global batch, GPUs the configuration names, T, H, W, scale)   -- H, W are the LR grid; output is (H*scale, W*scale)
    "c1": (2, 1, 10, 64, 64, 1),      # BASELINE configs[0] shape (CPU plumbing case)
    "c2": (8, 1, 10, 128, 128, 1),    # configs[1]
    "c3": (32, 8, 10, 256, 256, 1),   # configs[2]: the configuration the metric is quoted on (4 per GPU on its 8 GPUs)
    "c4": (16, 4, 20, 256, 256, 1),   # configs[3] (generator only; 4 per GPU on its 4 GPUs)
    "c5": (8, 8, 10, 512, 512, 1),    # configs[4] (1 per GPU on its 8 GPUs); its gate convs default to the fp8 (e4m3) path
    "p2": (32, 8, 10, 32, 32, 8),     # default.yaml-faithful x8 downscaling to 256x256, c3's batch
    "c0": (8, 1, 5, 15, 12, 8),       # the reference's own shape: default.yaml (batch 8, T=5, x8) on its 15x12 basin grid -> 120x96
}
FLOP_PER_PX_FWD = 171104          # algorithmic forward FLOPs per output pixel per frame (scale 1, L=5)
FWD_BYTES_PER_PX = 804            # algorithmic forward tensor traffic per pixel per frame: 402 elements as bf16
CELL2_FLOP_PER_PX = 2 * 55296     # cell-2 gate conv: (16+32)*9*128 MACs per pixel
CELL2_BYTES_PER_PX = 416          # x 32 + h 64 + c 128 read, h' 64 + c' 128 written (bf16 h, fp32 c)
PEAK_BF16 = 2.5e15                # dense bf16 MFMA peak of MI355X, FLOP/s
PEAK_FP8 = 5.0e15                 # dense fp8 peak (block-scaled MFMA), FLOP/s
PEAK_HBM = 8.0e12                 # HBM3E peak of MI355X, B/s
# pointwise LSTM backward, per 8-channel unit of one pixel: two dh (bf16) 32 + dc' (bf16) 16 + c (fp32) 32
# + gates i,f,o,g (bf16) 64 read; dz (bf16) 64 + dc (bf16) 16 written
PW_BWD_BYTES_PER_UNIT = 224
N_STATIONS = 35


def synthetic(B, T, H, W, s, seed, device):
    g = torch.Generator().manual_seed(seed)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)
    dem = torch.rand(B, 1, H * s, W * s, generator=g)
    lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, H * s, W * s), generator=g), 5)
    lu = lu.permute(0, 3, 1, 2).float().contiguous()
    coords = torch.randint(0, min(H, W), (N_STATIONS, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    values = 20 * torch.rand(B, T, N_STATIONS, generator=g)
    return tuple(t.to(device) for t in (rain, dem, lu, coords, values))


def time_cell2_kernel(B, H, W, device, iters=50):
    """Average duration of the dominant kernel (cell-2 gate conv + fused LSTM epilogue), measured
    with events on the stream it is launched on."""
    from pl_convlstm_gan_amd import _lib, ops
    L = _lib.lib()
    hw = H * W
    x = torch.randn(B, 16, H, W, device=device)
    h = torch.tanh(torch.randn(B, 32, H, W, device=device))
    x8, h8 = ops.to_cb8_bf16(x), ops.to_cb8_bf16(h)
    c = torch.randn(B * 32 * hw, device=device)
    w = 0.05 * torch.randn(128, 48, 3, 3, device=device)
    b = torch.zeros(128, device=device)
    wpk = torch.empty(L.plc_packed_weight_bytes(128, 16, 32, 3), dtype=torch.uint8, device=device)
    bpk = torch.empty(L.plc_packed_bias_bytes(128), dtype=torch.uint8, device=device)
    st = ops._stream()
    _lib.check(L.plc_pack_conv_weight(w.data_ptr(), b.data_ptr(), wpk.data_ptr(), bpk.data_ptr(), 128, 16, 32, 3, 1, st), "pack")
    ho = torch.empty_like(h8)
    co = torch.empty_like(c)

    def run():
        _lib.check(L.plc_convlstm_cell_fwd_cb8(x8.data_ptr(), h8.data_ptr(), c.data_ptr(), wpk.data_ptr(), bpk.data_ptr(),
                                               ho.data_ptr(), co.data_ptr(), None, B, 16, 32, H, W, st), "cell2")
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def time_cell2_kernel_fp8(B, H, W, device, iters=50):
    """The same measurement for the fp8 (e4m3, block-scaled MFMA) cell-2 gate conv."""
    from pl_convlstm_gan_amd import _lib, ops
    L = _lib.lib()
    hw = H * W
    x = torch.randn(B, 16, H, W, device=device)
    h = torch.tanh(torch.randn(B, 32, H, W, device=device))
    st = ops._stream()
    x8 = torch.empty(B, 1, H, W, 16, dtype=torch.uint8, device=device)
    h8 = torch.empty(B, 2, H, W, 16, dtype=torch.uint8, device=device)
    _lib.check(L.plc_cb8_bf16_to_cb16_fp8(ops.to_cb8_bf16(x).data_ptr(), x8.data_ptr(), B, 16, H, W, st), "fp8 x")
    _lib.check(L.plc_cb8_bf16_to_cb16_fp8(ops.to_cb8_bf16(h).data_ptr(), h8.data_ptr(), B, 32, H, W, st), "fp8 h")
    c = torch.randn(B * 32 * hw, device=device)
    w = 0.05 * torch.randn(128, 48, 3, 3, device=device)
    b = torch.zeros(128, device=device)
    wpk16 = torch.empty(L.plc_packed_weight_bytes(128, 16, 32, 3), dtype=torch.uint8, device=device)
    bpk = torch.empty(L.plc_packed_bias_bytes(128), dtype=torch.uint8, device=device)
    _lib.check(L.plc_pack_conv_weight(w.data_ptr(), b.data_ptr(), wpk16.data_ptr(), bpk.data_ptr(), 128, 16, 32, 3, 1, st), "pack")
    wpk = torch.empty(L.plc_fp8_packed_weight_bytes(16, 32), dtype=torch.uint8, device=device)
    sw = torch.zeros(2, dtype=torch.int32, device=device)
    _lib.check(L.plc_pack_lstm_weight_fp8(w.data_ptr(), wpk.data_ptr(), sw.data_ptr(), 16, 32, st), "pack fp8")
    ho = torch.empty(B, 4, H, W, 8, dtype=torch.int16, device=device)
    h8o = torch.empty_like(h8)
    co = torch.empty_like(c)

    def run():
        _lib.check(L.plc_convlstm_cell_fwd_fp8(x8.data_ptr(), h8.data_ptr(), c.data_ptr(), wpk.data_ptr(), sw.data_ptr(),
                                               bpk.data_ptr(), ho.data_ptr(), h8o.data_ptr(), co.data_ptr(), None,
                                               B, 16, 32, H, W, st), "cell2 fp8")
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def time_pointwise_bwd_launch(B, H, W, device, iters=50):
    """Average duration of the launch that takes the largest share of a train step, as the step issues it:
    the pointwise LSTM backward of cell 2 (step t - 1) and of cell 1 (step t) in ONE launch
    (plc_lstm_pointwise_bwd_pair_kernel), each with two incoming dh and dc' as in every step of the loop
    but its ends, over B images (the step runs it on each half of its batch).  Measured with events on the
    stream it is launched on."""
    from pl_convlstm_gan_amd import _lib, ops
    L = _lib.lib()
    bf = lambda *shape: (torch.randn(*shape, device=device) * 0.5).to(torch.bfloat16).view(torch.int16)
    # operand sets used in rotation (> 1 GB in total), so that no launch finds its inputs in the
    # 256 MB Infinity Cache: in the step the gates and cell states were written long before
    per_set = B * 6 * H * W * (3 * 16 + 64 + 32 + 64 + 16)
    sets = []
    for _ in range(max(2, min(6, (1 << 30) // per_set + 1))):
        cells = []
        for Ch in (32, 16):
            ncb = Ch // 8
            gates = torch.rand(B, 4 * ncb, H, W, 8, device=device).to(torch.bfloat16).view(torch.int16)
            c_prev = torch.randn(B, ncb, H, W, 8, device=device)
            t = (bf(B, ncb, H, W, 8), bf(B, ncb, H, W, 8), bf(B, ncb, H, W, 8), gates, c_prev, torch.empty_like(gates),
                 bf(B, ncb, H, W, 8))
            a = _lib.PointwiseBwdArgs()
            a.dh0, a.dh1, a.dc_next, a.gates, a.c_prev, a.dz, a.dc = (x.data_ptr() for x in t)
            a.dh0_image_bytes = a.dh1_image_bytes = ncb * H * W * 16
            a.Ch = Ch
            cells.append((a, t))
        sets.append(cells)
    st = ops._stream()

    def run(k):
        (a2, _), (a1, _) = sets[k % len(sets)]
        _lib.check(L.plc_lstm_pointwise_bwd_pair_bf16dc(a2, a1, B, H, W, st), "pointwise_bwd_pair")
    for k in range(5):
        run(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(iters):
        run(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def cpu_baseline(mode, B, T, H, W, s, params):
    """The CPU oracle (a port of the reference) timed on this host: best of 3 on a bounded sample
    (one batch element of the per-GPU workload when training, the full shard for inference)."""
    from oracle import cpu_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    Bc = 1 if mode == "train" else B
    rain, dem, lu, coords, values = synthetic(Bc, T, H, W, s, 99, "cpu")
    best = float("inf")
    if mode == "train":
        O.grads_of(params, rain[:, :2], dem, lu, coords, values[:, :2], fwd_kw=dict(scale_factor=s))
        for _ in range(3):
            t0 = time.perf_counter()
            O.grads_of(params, rain, dem, lu, coords, values, fwd_kw=dict(scale_factor=s))
            best = min(best, time.perf_counter() - t0)
        what = "forward + CombinedLoss + autograd backward (no optimizer)"
    else:
        with torch.no_grad():
            O.generator_forward(params, rain[:1, :1], dem[:1], lu[:1], scale_factor=s)
            for _ in range(3):
                t0 = time.perf_counter()
                O.generator_forward(params, rain, dem, lu, scale_factor=s)
                best = min(best, time.perf_counter() - t0)
        what = "forward"
    return {"value": Bc * T / best, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"B={Bc} of the per-GPU workload (T={T}, {H * s}x{W * s}), fp32 PyTorch-CPU oracle {what}, best of 3"}


def first_existing(*names):
    """The first of profiles/<name> that exists (this round's record before an older one), else the first name."""
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return os.path.join(ROOT, "profiles", n)
    return os.path.join(ROOT, "profiles", names[0])


def launch_ranks(n, argv):
    """Start n ranks of this script under torch.distributed.run (a child process; nothing here has
    touched a GPU yet) and return its exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total batch over all ranks (default: the configuration's own global batch, 32 for "
                         "BASELINE configs[2]; strong scaling)")
    ap.add_argument("--shard", action="store_true",
                    help="every rank runs the configuration's per-GPU shard of its own GPU count (c3: B=4 per GPU) "
                         "whatever --gpus is: weak scaling")
    ap.add_argument("--hidden", default="16,32", help="hidden_dims of the generator (the benchmark configs use 16,32; "
                    "32,64 is the reference's class default and runs on the run-time-width kernels)")
    ap.add_argument("--gate-dtype", default=None, choices=["bf16", "fp8"],
                    help="operand type of the two gate convs (default: fp8 for c5, BASELINE configs[4]; bf16 otherwise)")
    ap.add_argument("--graph", action="store_true",
                    help="train mode, one GPU: capture the step into a HIP graph and replay it (pays on launch-bound "
                         "shapes such as c0 / p2 / c1; the large configs are GPU-bound either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the timed steps and only time the roofline kernels' stand-alone launches (for a profiler "
                         "trace that holds nothing else under those kernel names); the line's throughput fields are null")
    ap.add_argument("--shape", default=None, help="B,T,H,W,scale in place of the named configuration's (measurement aid: "
                                                  "the line then describes that shape, not a BASELINE configuration)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    backend = os.environ.get("PLC_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: become the launcher.  device_count() does not initialise the GPU.
        have = torch.cuda.device_count()
        if backend == "nccl" and have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this box has {have} GPU(s); nothing was run")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; nothing was run")
    # PLC_BENCH_BACKEND=gloo is a rehearsal aid: several ranks share one GPU to exercise the
    # multi-process control flow on a single-GPU box (numbers from it are meaningless)
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s): one rank per GPU")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that does not complete within the timeout aborts the process (non-zero exit)
        # instead of hanging until the job's lease runs out
        patience = datetime.timedelta(seconds=int(os.environ.get("PLC_BENCH_PG_TIMEOUT", "180")))
        if backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=patience)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)

    def dbg(msg):
        if os.environ.get("PLC_BENCH_DEBUG"):
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    dbg("process group ready")
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.training import train_step

    GB, native_gpus, T, H, W, s = CONFIGS[args.config]
    if args.shard and args.global_batch:
        raise SystemExit("bench.py: --shard and --global-batch exclude each other")
    if args.shape:
        B, T, H, W, s = (int(v) for v in args.shape.split(","))
        args.config = "custom"
        scaling = "weak"                      # (the given B on every rank)
    elif args.shard:
        B, scaling = GB // native_gpus, "weak"
    else:
        GB = args.global_batch or GB
        if GB % world:
            raise SystemExit(f"bench.py: the global batch {GB} does not divide over {world} ranks")
        B, scaling = GB // world, "strong"
    torch.manual_seed(0)                       # identical initial weights on every rank
    hidden = [int(v) for v in args.hidden.split(",")]
    model = Generator(hidden_dims=hidden, lu_channels=5, scale_factor=s)
    rain, dem, lu, coords, values = synthetic(B, T, H, W, s, 1234 + rank, device)
    model = model.to(device)
    gate_dtype = args.gate_dtype or ("fp8" if args.config == "c5" and hidden == [16, 32] else "bf16")
    model.gate_dtype = gate_dtype
    with torch.no_grad():
        model(rain[:1, :1].contiguous(), dem[:1].contiguous(), lu[:1].contiguous())   # materialise lazy blocks
    loss_module = CombinedLoss()
    from pl_convlstm_gan_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=5e-4)   # torch.optim.Adam's rule; clip + step in two launches
    dbg("model and data ready")

    graphed = None
    if args.graph:
        if world > 1 or args.mode != "train":
            raise SystemExit("bench.py: --graph is for --mode train on one GPU")
        from pl_convlstm_gan_amd.training import GraphedTrainStep
        graphed = GraphedTrainStep(model, loss_module, opt, (rain, dem, lu, coords, values), max_norm=0.5)

    def step_train():
        # nan_check="device": the reference's NaN-batch skip is carried out by the optimizer kernel
        # (no host synchronisation inside the step), as the Trainer runs it
        if graphed is not None:
            return graphed(rain, dem, lu, coords, values)[0]
        return train_step(model, loss_module, opt, rain, dem, lu, coords, values, max_norm=0.5, nan_check="device")[0]

    def step_fwd():
        with torch.no_grad():
            return model(rain, dem, lu)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for i in range(warmup):
            fn()
            dbg(f"warmup step {i} done")
        barrier()
        dbg("timed region starts")
        t0 = time.perf_counter()
        for _ in range(steps):
            last = fn()
        barrier()
        dbg("timed region done")
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert torch.isfinite(last.detach()).all()
        return dt

    main_fn = step_train if args.mode == "train" else step_fwd
    if args.roofline_only:
        dt = dt_fwd = float("nan")
        args.no_cpu_baseline = True
    else:
        dt = timed(main_fn, args.steps, args.warmup)
        # the forward-only leg of a train run gets the same warm-up and step count as the train leg
        dt_fwd = timed(step_fwd, args.steps, args.warmup) if args.mode == "train" else dt
    n_fwd = args.steps

    if rank == 0:
        frames = world * B * T * args.steps
        what = ("train step (forward + CombinedLoss + BPTT backward + grad all-reduce + clip + Adam)"
                if args.mode == "train" else "rollout forward (inference)")
        res = {
            "metric": "predicted frames/sec (whole node), BxTxCx256x256" if args.config in ("c3", "c4", "p2")
                      else "predicted frames/sec (whole node)",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": gate_dtype if gate_dtype == "fp8" else "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {what}, B={B}/GPU, T={T}, {H}x{W} -> {H * s}x{W * s}, 1 channel, "
                                   f"hidden [{args.hidden}], L=5, scale_factor={s}, {N_STATIONS} stations, gate convs {gate_dtype}"
                                   + (" (e4m3 operands, fp32 accumulate; every other conv and the backward pass bf16; the path BASELINE "
                                      "configs[4] names -- with --gate-dtype bf16 the same configuration measures within 2 %)" if gate_dtype == "fp8" else ""),
                       "global_batch": world * B, "T": T, "mode": args.mode + (" (HIP-graph replay)" if args.graph else ""),
                       "parallelism": f"dp{world} (batch sharded; one flat-gradient all-reduce per step)"},
            "fwd_frames_per_s": world * B * T * n_fwd / dt_fwd,
        }
        # whole-path view next to the per-kernel rooflines: algorithmic forward FLOPs per frame
        # (171 104 per output pixel at scale 1; a train step is 3x: data + weight gradients) and, for the
        # forward, the algorithmic bf16 tensor traffic per frame (402 elements per pixel) over the step time
        if s == 1 and hidden == [16, 32]:
            px = H * W
            fwd_flop = FLOP_PER_PX_FWD * px * world * B * T
            res["step_mfma_frac"] = (3 if args.mode == "train" else 1) * fwd_flop / (dt / args.steps) / (world * PEAK_BF16)
            res["fwd_mfma_frac"] = fwd_flop / (dt_fwd / n_fwd) / (world * PEAK_BF16)
            res["fwd_hbm_frac"] = FWD_BYTES_PER_PX * px * world * B * T / (dt_fwd / n_fwd) / (world * PEAK_HBM)
        # the train step against the HBM roof by MEASURED traffic: bytes of every dispatch of one step from the
        # committed PMC passes (profiles/step_traffic.py) over this run's step time
        sj = first_existing("r3_train_c3_step_traffic.json" if B == 32 else "r3_train_c3_shard_step_traffic.json")
        if args.mode == "train" and args.config == "c3" and hidden == [16, 32] and gate_dtype == "bf16" \
                and B in (4, 32) and os.path.exists(sj):
            step_bytes = json.load(open(sj)).get("hbm_bytes_per_step")
            if step_bytes:
                res["step_hbm"] = {"traffic_bytes_per_step": step_bytes, "source": os.path.relpath(sj, ROOT),
                                   "achieved": step_bytes / (dt / args.steps) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                                   "frac": step_bytes / (dt / args.steps) / PEAK_HBM}
        # dominant kernel: cell-2 gate conv, one launch per timestep
        fp8 = gate_dtype == "fp8"
        # (a forward recurrence that runs as two half-batch chains launches the cell on half the batch)
        Bk = B // 2 if (not fp8 and B >= 2 and B * H * W >= (1 << 17)) else B
        tk = time_cell2_kernel_fp8(Bk, H, W, device) if fp8 else time_cell2_kernel(Bk, H, W, device)
        flops = CELL2_FLOP_PER_PX * Bk * H * W
        traffic = None
        tj = first_existing("r3_cell2_traffic.json")
        if os.path.exists(tj) and args.config == "c3":
            # (the record of the launch size timed here, or none)
            traffic = json.load(open(tj))["by_images"].get(str(Bk), {}).get("hbm_bytes_per_launch")
        peak = PEAK_FP8 if fp8 else PEAK_BF16
        fp8_traffic = None
        fj = os.path.join(ROOT, "profiles", "r2_cell2_fp8_traffic.json")
        if fp8 and os.path.exists(fj) and args.config == "c5":
            fp8_traffic = json.load(open(fj)).get("hbm_bytes_per_launch")
        # fp8 operands: x 16 + h 32 bytes per pixel read (instead of 32 + 64), h' written twice (bf16 64 + fp8 32)
        cell_bytes = (CELL2_BYTES_PER_PX - 48 + 32) if fp8 else CELL2_BYTES_PER_PX
        mfma_roof = {"bound": "mfma",
                     "kernel": ("plc_lstm_fp8_kernel<1,2,4> (cell-2 gate conv, e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4, + LSTM epilogue)"
                                if fp8 else "plc_conv_mfma_kernel<2,4,4,3,CB8,LSTM> (cell-2 gate conv + LSTM epilogue)"),
                     "achieved": flops / tk / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                     "frac": flops / tk / peak, "traffic": fp8_traffic if fp8 else traffic,
                     "launch_us": tk * 1e6, "flop_per_launch": flops,
                     "images_per_launch": Bk,
                     "alg_bytes_per_launch": cell_bytes * Bk * H * W,
                     "hbm_frac": cell_bytes * Bk * H * W / tk / PEAK_HBM}
        if args.mode == "train":
            # the kernel with the largest share of a train step is the HBM-bound pointwise LSTM
            # backward (2T launches); the cell-2 conv, top kernel of the forward, follows as roofline_mfma
            # the launch as the step issues it: both cells' passes in one launch, on each half of the batch when the
            # backward recurrence runs as two chains (B >= 2 and at least 2^17 pixels per time step)
            nb = B // 2 if (B >= 2 and B * H * W >= (1 << 17)) else B
            tp = time_pointwise_bwd_launch(nb, H, W, device)
            pw_bytes = PW_BWD_BYTES_PER_UNIT * nb * 6 * H * W         # cell 2: 4 channel blocks per pixel, cell 1: 2
            pw_traffic = None
            pj = first_existing("r3_pointwise_bwd_pair_traffic.json")
            if os.path.exists(pj) and args.config == "c3":
                pw_traffic = json.load(open(pj))["by_images"].get(str(nb), {}).get("hbm_bytes_per_launch")
            res["roofline"] = {"bound": "hbm",
                               "kernel": f"plc_lstm_pointwise_bwd_pair_kernel (cell 2 of step t-1 + cell 1 of step t, {nb} images: "
                                         "the launch of the skewed BPTT loop" + (", one of its two half-batch chains)" if nb != B else ")"),
                               "achieved": pw_bytes / tp / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                               "frac": pw_bytes / tp / PEAK_HBM, "traffic": pw_traffic,
                               "launch_us": tp * 1e6, "alg_bytes_per_launch": pw_bytes}
            res["roofline_mfma"] = mfma_roof
        else:
            res["roofline"] = mfma_roof
        if hidden != [16, 32]:
            note = "timed at hidden_dims [16,32]: the instantiated kernels, not the run-time-width ones this run used"
            for k in ("roofline", "roofline_mfma"):
                if k in res:
                    res[k]["note"] = note
        if not args.no_cpu_baseline and world == 1:
            params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            res["cpu_baseline"] = cpu_baseline(args.mode, B, T, H, W, s, params)
        if args.roofline_only:
            res = {k: v for k, v in res.items() if k in ("n_gpus", "dtype", "config", "roofline", "roofline_mfma")}
            res["roofline_only"] = True
        print(json.dumps(res), flush=True)
    if world > 1:
        barrier()        # rank 0 is still timing the roofline kernel / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
