#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full ProtoPFormer train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" (tools/engine_proto.py:41-81): resident synthetic batch -> forward (train branch, DropPath 0.1 active) -> CE +
0.1*PPC_sigma + 0.5*PPC_mu -> backward -> gradient all-reduce (N > 1) -> AdamW (reference param groups) -> EMA.
Workload = BASELINE.json configs[2]: deit_small_patch16_224, 2000x384 prototypes, 200 classes, k = 81, batch 256 per GPU
(weak scaling), bf16 MFMA operands / fp32 accumulate, random-init weights, synthetic N(0,1) images.
Prints ONE JSON line (rank 0) with the roofline of the dominant kernel (measured live with HIP events on the launch
stream) and the CPU baseline (oracle port on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

# ROCm maps HIP streams round-robin onto GPU_MAX_HW_QUEUES (default 4) hardware queues; with the RCCL / comm streams of a
# multi-GPU run the weight-gradient lane could land on the main stream's queue and serialise with it.  Must be set before HIP loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARCH = "deit_small_patch16_224"
BATCH = 256
P, DP, C, K_TOK, GPC = 2000, 384, 200, 81, 10
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table
TRAIN_GFLOP_PER_IMG = 28.10        # algorithmic FLOPs of one train step per image (deit_small, Dp=384)


def build(device, seed):
    from protopformer_amd.engine import FlatAdamW, make_grad_sync
    from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
    torch.manual_seed(seed)
    model = construct_PPNet(ARCH, pretrained=False, img_size=224, prototype_shape=(P, DP, 1, 1), num_classes=C, reserve_layers=[11],
                            reserve_token_nums=[K_TOK], use_global=True, use_ppc_loss=True, ppc_cov_thresh=1., ppc_mean_thresh=2.,
                            global_coe=0.5, global_proto_per_class=GPC, prototype_activation_function="log", add_on_layers_type="regular")
    model = model.to(device)
    model.train()
    opt = FlatAdamW(model, weight_decay=0.05, ema_decay=0.99996)
    sync = make_grad_sync(model) if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0") else None
    return model, opt, CrossEntropyLoss(), sync


def cpu_baseline(batch=16, steps=2):
    """The oracle (a CPU port of the reference arithmetic, fp32) timed on this box's host cores on a bounded sample:
    `steps` train steps of the same architecture / head at a reduced batch."""
    from oracle import ppf_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    cfg = O.make_cfg(ARCH, P, DP, C, 11, K_TOK, global_per_class=GPC)
    sd = O.init_state_dict(cfg, seed=1028)
    params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
    opt = torch.optim.AdamW(O.adamw_groups(params), weight_decay=0.05, eps=1e-8)
    g = torch.Generator().manual_seed(1028)
    img = torch.randn(batch, 3, 224, 224, generator=g)
    label = torch.randint(0, C, (batch,), generator=g)
    rates = [r for i in range(12) for r in (0.1 * i / 11,)]
    ema = {k: v.detach().clone() for k, v in params.items()}
    O.train_step(params, opt, img, label, cfg, droppath=O.droppath_scales(batch, rates, g), ema=ema)      # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(params, opt, img, label, cfg, droppath=O.droppath_scales(batch, rates, g), ema=ema)
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{steps} fp32 train steps of {ARCH}+{P}x{DP} protos at batch {batch} (oracle/ppf_oracle.py, torch CPU, {threads} threads)"}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/r1_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 2x FETCH correction applied)."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    try:
        ks = json.load(open(path))["kernels"]
        for name, v in ks.items():
            if "gemm_kernel<true, true, 7" in name:
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", init_method="env://", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from protopformer_amd import _lib, ops
    from protopformer_amd.engine import train_one_step
    model, opt, crit, sync = build(device, seed=1028)          # same seed on every rank, as the reference (main.py:254-255)
    g = torch.Generator(device=device).manual_seed(1028 + rank)
    img = torch.randn(args.batch, 3, 224, 224, device=device, generator=g)
    label = torch.randint(0, C, (args.batch,), device=device, generator=g)

    def step():
        return train_one_step(model, crit, img, label, opt, epoch=20, grad_sync=sync)

    for _ in range(args.warmup):
        step()
    # roofline probe: HIP events around every launch of the dominant kernel (wgrad GEMM) on its launch stream
    _lib.call("ppf_gemm_probe", 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _, _ = step()
    t_enqueue = time.perf_counter() - t0          # host time to queue the steps: must stay well below the GPU time
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    _lib.call("ppf_gemm_probe", 0)
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    import ctypes
    c_ms, c_n, c_fl, c_by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    _lib.call("ppf_gemm_probe_read", ctypes.addressof(c_ms), ctypes.addressof(c_n), ctypes.addressof(c_fl), ctypes.addressof(c_by))
    kern_ms, n_launch, probe_flops, probe_bytes = c_ms.value, int(c_n.value), c_fl.value, c_by.value
    if rank == 0:
        ips = world * args.batch * args.steps / dt
        achieved = (probe_flops / 1e12) / (kern_ms / 1e3) if kern_ms > 0 else 0.0
        out = {
            "metric": "images/sec train step, deit_small+2000 protos, bs256, 1/2/4/8 MI355X", "value": ips, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{ARCH}, synthetic 224x224, {P}x{DP} prototypes, {C} classes, k={K_TOK}, batch {args.batch}/GPU, "
                                   "train step = fwd+CE+PPC+bwd+allreduce+AdamW+EMA, DropPath 0.1", "global_batch": world * args.batch,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": ops.DOMINANT_NAME, "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(), "traffic_unit": "bytes/launch (PMC, see profiles/r1_pmc_traffic.json)",
                         "algorithmic_bytes": probe_bytes / max(n_launch, 1), "launches": n_launch,
                         "avg_launch_ms": kern_ms / max(n_launch, 1)},
            "step_mfma_frac": (ips / world) * TRAIN_GFLOP_PER_IMG / 1e3 / PEAK_BF16_TFLOPS,
            "final_loss": float(loss),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
