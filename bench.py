#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full ProtoPFormer train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config deit_small|deit_tiny|cait_xxs24] [--graph]

With --gpus N > 1 and no torch.distributed environment, this process only LAUNCHES: it spawns N rank processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) before anything here touches the GPU, relays rank 0's
JSON line and exits non-zero if any rank failed.  Started by `python -m torch.distributed.run ... bench.py --gpus N` it is a rank.

A "step" (tools/engine_proto.py:41-81): resident synthetic batch -> forward (train branch, DropPath 0.1 active) -> CE +
0.1*PPC_sigma + 0.5*PPC_mu -> backward -> gradient all-reduce (N > 1) -> AdamW (reference param groups) -> EMA.
Default workload = BASELINE.json configs[2]: deit_small_patch16_224, 2000x384 prototypes, 200 classes, k = 81, batch 256 per GPU
(weak scaling), bf16 MFMA operands / fp32 accumulate, random-init weights, synthetic N(0,1) images.  Kernels are enqueued from
the host on two HIP streams each step; --graph replays one captured HIP graph per step instead (engine.GraphedTrainStep: measured
SLOWER on ROCm 7.2 -- hipGraphLaunch re-enqueues every node, 12.5 ms of host time per replay, and spreads the two-stream step over
four queues with half the overlap: 23.6-27.0 ms vs 18.9 ms per step, profiles/r2_graph_timeline.txt -- so it is opt-in).
Prints ONE JSON line (rank 0) with the roofline of the dominant kernel (HIP events on its launch stream, recorded inside the
timed region) and the CPU baseline (oracle port on the host cores, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# ROCm maps HIP streams round-robin onto GPU_MAX_HW_QUEUES (default 4) hardware queues; with the RCCL / comm streams of a
# multi-GPU run the weight-gradient lane could land on the main stream's queue and serialise with it.  Must be set before HIP loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table
METRIC = "images/sec train step, deit_small+2000 protos, bs256, 1/2/4/8 MI355X"

# BASELINE.json configs 3 (headline) / 2 / 5.  gflop = algorithmic FLOPs of one train step per image.
CONFIGS = {
    "deit_small": dict(arch="deit_small_patch16_224", batch=256, P=2000, Dp=384, C=200, k=81, gpc=10, layer=11, gflop=28.10,
                       D=384, H=6, depth=12, N=197, label="BASELINE.json configs[2]"),
    "deit_tiny": dict(arch="deit_tiny_patch16_224", batch=128, P=2000, Dp=192, C=200, k=81, gpc=10, layer=11, gflop=7.85,
                      D=192, H=3, depth=12, N=197, label="BASELINE.json configs[1]"),
    "cait_xxs24": dict(arch="cait_xxs24_224", batch=128, P=1960, Dp=192, C=196, k=121, gpc=5, layer=1, gflop=15.75,
                       D=192, H=4, depth=24, N=196, label="BASELINE.json configs[4] (per-GPU shape)"),
}


def wgrad_uncontended(cfg, batch, device):
    """The dominant kernel (weight-gradient GEMM + nothing else on the chip): the four per-block shapes of the configuration, timed with
    events on an otherwise idle GPU after the timed region.  `roofline.frac` is the CONTENDED in-step figure the contract asks for (the
    kernel shares the chip with the main stream's chain); this is the same kernel's own speed, reported next to it."""
    import torch
    from protopformer_amd import ops
    from protopformer_amd.backbone import EPI_ATOMIC
    D, M = cfg["D"], batch * cfg["N"]
    shapes = [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D)]                     # qkv, proj, fc1, fc2 weights [out, in]
    g = torch.Generator(device=device).manual_seed(3)
    tot_ms, tot_fl = 0.0, 0.0
    for n_out, n_in in shapes:
        dy = torch.randn(M, n_out, device=device, generator=g).bfloat16()
        x = torch.randn(M, n_in, device=device, generator=g).bfloat16()
        gw = torch.zeros(n_out, n_in, device=device)
        fn = lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, epi=EPI_ATOMIC, out=gw)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fn()
        e1.record()
        torch.cuda.synchronize()
        tot_ms += e0.elapsed_time(e1) / 8
        tot_fl += 2.0 * M * n_out * n_in
    return {"avg_launch_ms_incl_reduce": tot_ms / len(shapes), "achieved": tot_fl / 1e12 / (tot_ms / 1e3), "unit": "TFLOP/s",
            "frac": tot_fl / 1e12 / (tot_ms / 1e3) / PEAK_BF16_TFLOPS,
            "how": "qkv / proj / fc1 / fc2 weight gradients of one block, 8 back-to-back launches each on an idle GPU after the timed region; "
                   "includes the ordered split-K reduce launch behind every GEMM"}


def executed_gflop_per_img(cfg):
    """FLOPs the kernels really execute: DeiT blocks from the reservation layer on run on the 1+k reserved rows only
    (backbone.deit_blocks_fwd), so their linear parts scale with (1+k)/N and their attention with ((1+k)/N)^2."""
    if not cfg["arch"].startswith("deit"):
        return cfg["gflop"]
    D, H, N = cfg["D"], cfg["H"], cfg["N"]
    hd = D // H
    lin = 2 * N * D * 3 * D + 2 * N * D * D + 4 * N * D * 4 * D
    att = 4 * H * N * N * hd
    r = (1 + cfg["k"]) / N
    saved_fwd = (cfg["depth"] - cfg["layer"]) * (lin * (1 - r) + att * (1 - r * r))
    return cfg["gflop"] - 3 * saved_fwd / 1e9


def build(cfg, device, seed):
    import torch
    import torch.distributed as dist
    from protopformer_amd.engine import FlatAdamW, make_grad_sync
    from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
    torch.manual_seed(seed)
    model = construct_PPNet(cfg["arch"], pretrained=False, img_size=224, prototype_shape=(cfg["P"], cfg["Dp"], 1, 1), num_classes=cfg["C"],
                            reserve_layers=[cfg["layer"]], reserve_token_nums=[cfg["k"]], use_global=True, use_ppc_loss=True, ppc_cov_thresh=1.,
                            ppc_mean_thresh=2., global_coe=0.5, global_proto_per_class=cfg["gpc"], prototype_activation_function="log",
                            add_on_layers_type=cfg.get("addon", "regular"))
    model = model.to(device)
    model.train()
    opt = FlatAdamW(model, weight_decay=0.05, ema_decay=0.99996)
    sync = None
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0"):
        sync = make_grad_sync(model, opt)            # rank-0 broadcast of parameters / optimizer state, as DDP does at wrap time
    return model, opt, CrossEntropyLoss(), sync


def _lscpu():
    info = {}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            info[k.strip()] = v.strip()
    except Exception:
        pass
    try:
        sockets, cps = int(info.get("Socket(s)", "0")), int(info.get("Core(s) per socket", "0"))
    except ValueError:
        sockets = cps = 0
    return {"model": info.get("Model name", "unknown"), "sockets": sockets, "physical_cores": sockets * cps, "logical_cpus": os.cpu_count()}


def _cpu_train_steps(arch, P, Dp, C, k, gpc, batch, warmup, steps):
    import torch
    from oracle import ppf_oracle as O
    cfg = O.make_cfg(arch, P, Dp, C, 11, k, global_per_class=gpc)
    sd = O.init_state_dict(cfg, seed=1028)
    params = {n: v.clone().requires_grad_(n not in O.FROZEN_KEYS) for n, v in sd.items()}
    opt = torch.optim.AdamW(O.adamw_groups(params), weight_decay=0.05, eps=1e-8)
    g = torch.Generator().manual_seed(1028)
    img = torch.randn(batch, 3, 224, 224, generator=g)
    label = torch.randint(0, C, (batch,), generator=g)
    rates = [0.1 * i / 11 for i in range(12)]
    ema = {n: v.detach().clone() for n, v in params.items()}
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        O.train_step(params, opt, img, label, cfg, droppath=O.droppath_scales(batch, rates, g), ema=ema)
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"img_per_s": batch / med, "median_s_per_step": med, "batch": batch, "warmup": warmup, "timed_steps": steps}


def cpu_baseline():
    """the oracle (fp32 CPU port of the reference arithmetic) timed on this box's host cores, same step
    definition, >= 3 warm-up + >= 5 timed steps, median; like-for-like = deit_small 2000x384 at bs32 (the headline workload at a
    bounded batch), plus BASELINE.json configs[0] (deit_tiny, 2000x192, bs32)."""
    import torch
    # every host core the box has.  torch's intra-op pool is sized by set_num_threads; the physical core count
    # (not the SMT count) is what the fp32 GEMM-bound oracle scales with, so that is the thread count, all sockets included.
    cpu = _lscpu()
    allc = cpu["physical_cores"] or (os.cpu_count() or 1)
    tried = {}
    torch.set_num_threads(allc)
    small = _cpu_train_steps("deit_small_patch16_224", 2000, 384, 200, 81, 10, batch=32, warmup=3, steps=5)
    tried[allc] = small["img_per_s"]
    threads = allc
    if cpu["sockets"] > 1:                       # one socket's worth of threads (no cross-socket traffic): report whichever is faster
        one = max(1, allc // cpu["sockets"])
        torch.set_num_threads(one)
        alt = _cpu_train_steps("deit_small_patch16_224", 2000, 384, 200, 81, 10, batch=32, warmup=2, steps=5)
        tried[one] = alt["img_per_s"]
        if alt["img_per_s"] > small["img_per_s"]:
            small, threads = alt, one
    torch.set_num_threads(threads)
    tiny = _cpu_train_steps("deit_tiny_patch16_224", 2000, 192, 200, 81, 10, batch=32, warmup=3, steps=5)
    return {"value": small["img_per_s"], "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"median of {small['timed_steps']} fp32 train steps (after {small['warmup']} warm-up) of deit_small_patch16_224+2000x384 "
                      f"protos at batch 32 (oracle/ppf_oracle.py, torch CPU, {threads} threads = the faster of {sorted(tried)} threads tried)",
            "median_s_per_step": small["median_s_per_step"], "cpu": cpu, "img_per_s_by_threads": {str(k): v for k, v in tried.items()},
            "config1_deit_tiny_bs32": {"value": tiny["img_per_s"], "median_s_per_step": tiny["median_s_per_step"],
                                       "sample": f"median of {tiny['timed_steps']} steps after {tiny['warmup']} warm-up, deit_tiny_patch16_224+2000x192 protos, batch 32"}}


def pmc_traffic(config="deit_small"):
    """HBM bytes per launch of the dominant kernel (the weight-gradient GEMM group) from the committed PMC passes of this same command and
    configuration (profiles/r6_*pmc_traffic.json, older rounds for deit_small as a fallback: separate --pmc FETCH_SIZE / WRITE_SIZE runs,
    gfx950 2x FETCH correction).  Round 6: every configuration has its own table (deit_tiny, cait_xxs24: profiles/r6_<config>_pmc_traffic.json)."""
    tag = {"deit_small": "", "deit_tiny": "deit_tiny_", "cait_xxs24": "cait_"}.get(config)
    if tag is None:
        return None, None
    names = [f"r6_{tag}pmc_traffic.json"] + (["r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json"] if config == "deit_small" else [])
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        try:
            ks = json.load(open(path))["kernels"]
            # the weight gradient is three kernels since round 4 (wgrad8 256x128 / 128x256 + the 128x128 one): launch-weighted mean
            hit = [v for kname, v in ks.items() if "gemm_kernel<true, true, 7" in kname or "wgrad8_kernel" in kname]
            if hit:
                n = sum(v["launches"] for v in hit)
                return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / max(n, 1), name
        except Exception:
            continue
    return None, None


def secondary_configs(names=("deit_tiny", "cait_xxs24"), steps=20, warmup=5, timeout=600):
    """The other two single-GPU BASELINE configurations, one child process each (same script, --config, no CPU baseline): their value,
    ms per step, host enqueue time and step MFMA fraction, so that the driver's one default run shows all three."""
    out = {}
    for name in names:
        cfg = CONFIGS[name]
        key = f"{name}_bs{cfg['batch']}"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", str(warmup),
                                "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=timeout)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[key] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            out[key] = {k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "host_enqueue_ms_per_step",
                                          "host_enqueue_ms_one_step_empty_queue", "step_mfma_frac", "dtype", "final_loss")}
            out[key]["workload"] = j["config"]["workload"]
        except Exception as e:                      # never let a side measurement take the headline line down
            out[key] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


# ------------------------------------------------------------------------------------------------ launcher (N > 1, no dist env)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Parent of a multi-GPU run: spawns one rank per GPU and relays rank 0's JSON line.  Makes NO GPU call itself."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else None,
                                      stderr=None, text=True))
    deadline = time.time() + args.launch_timeout
    out0 = ""
    rcs = [None] * len(procs)
    try:
        out0, _ = procs[0].communicate(timeout=args.launch_timeout)
        rcs[0] = procs[0].returncode
        for i, p in enumerate(procs[1:], 1):
            rcs[i] = p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        pass
    finally:
        for p in procs:                       # the exact processes started above, nothing else
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(i, rc) for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed or timed out (rank, rc): {bad}\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------------ dry run (CPU, gloo): launch plumbing
def dry_run(args, world, rank):
    """The N > 1 control path without a GPU: rendezvous, chunked gradient all-reduce (engine.GradSync) over gloo, barrier +
    max-over-ranks timing, one JSON line from rank 0.  Used by tests/test_bench_launch_cpu.py; not a measurement."""
    import torch
    import torch.distributed as dist
    from protopformer_amd.engine import GradSync
    if world > 1:
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    g = torch.full((1 << 16,), float(rank + 1))
    sync = GradSync(g, [0, 1 << 14, 1 << 15, 1 << 16], use_side_stream=False)
    for _ in range(args.warmup):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.fill_(float(rank + 1))
        for c in (2, 1, 0):
            sync.chunk_ready(c)
        scale = sync.finish()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ok = abs(float(g[0]) * scale - (world + 1) / 2.0) < 1e-6
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * float(dt) / max(args.steps, 1), "dry_run": True, "allreduce_ok": bool(ok), "scaling": "weak",
                          "higher_is_better": True, "data": "synthetic"}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


PEAK_HBM_TBPS = 8.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E peak (6.3 TB/s measured achievable)


def named_path_report(_lib, steps):
    """roofline.named_path: the kernels BASELINE.json's north_star names, timed inside real train steps by the library's path probe
    (HIP events on the launch stream around each launch; csrc/ppf_runtime.hip).  Algorithmic flops / bytes as DESIGN.md section 4 states them."""
    import ctypes
    out = {"how": f"HIP events around every launch in {steps} train steps after the timed region (contended in-step time); "
                  "flops / bytes are algorithmic (DESIGN.md section 4)", "peak_mfma_tflops": PEAK_BF16_TFLOPS, "peak_hbm_tbps": PEAK_HBM_TBPS}
    for tag, name in ((0, "attention_fwd"), (1, "attention_bwd"), (2, "prototype_fwd")):
        c_ms, c_n, c_fl, c_by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.call("ppf_path_probe_read", tag, ctypes.addressof(c_ms), ctypes.addressof(c_n), ctypes.addressof(c_fl), ctypes.addressof(c_by))
        ms, n = c_ms.value, int(c_n.value)
        if n == 0 or ms <= 0:
            out[name] = {"launches": 0}
            continue
        tf, tb = c_fl.value / 1e12 / (ms / 1e3), c_by.value / 1e12 / (ms / 1e3)
        out[name] = {"launches_per_step": n / steps, "avg_launch_us": 1e3 * ms / n, "ms_per_step": ms / steps,
                     "gflop_per_launch": c_fl.value / n / 1e9, "mbytes_per_launch": c_by.value / n / 1e6,
                     "tflops": tf, "mfma_frac": tf / PEAK_BF16_TFLOPS, "hbm_tbps": tb, "hbm_frac": tb / PEAK_HBM_TBPS}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="deit_small")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay one captured HIP graph per step instead of enqueueing every kernel from the host")
    ap.add_argument("--eager", action="store_true", help="run the Python orchestration every step instead of replaying the recorded command list")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo check of the multi-rank launch + all-reduce plumbing (no GPU, no measurement)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0)
    ap.add_argument("--wgrad-alone", action="store_true", help="(internal) only time the weight-gradient GEMMs of --config on an idle GPU and print that JSON")
    ap.add_argument("--addon", choices=["regular", "bottleneck"], default="regular",
                    help="add_on_layers_type: main.py:49 passes 'regular' (the BASELINE configurations); 'bottleneck' is the reference signature's default (not the headline metric)")
    ap.add_argument("--proto-dim", type=int, default=None, help="prototype width (with --addon bottleneck: < D/2 gives the reference's multi-stage tail)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other two BASELINE configurations after the headline one (--no-cpu-baseline skips them too)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path)")
    cfg = dict(CONFIGS[args.config])
    cfg["addon"] = args.addon
    if args.proto_dim:
        cfg["Dp"] = args.proto_dim
    batch = args.batch or cfg["batch"]
    # test-only: PPF_BENCH_ONE_GPU=1 puts every rank on cuda:0 and makes gloo the process-group backend, so that the whole
    # multi-rank flow of this file (broadcast, chunked all-reduce, guard, replay with live collectives, probe legs) runs on a 1-GPU box
    # (scripts/gpu/bench_two_ranks_check.sh: ~10 s per step through gloo, too slow for the test suite); RCCL itself refuses two ranks on one device
    if os.environ.get("PPF_BENCH_ONE_GPU", "0") != "0":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.wgrad_alone:
        print(json.dumps(wgrad_uncontended(cfg, batch, torch.device("cuda", local_rank))), flush=True)
        return
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = "gloo" if os.environ.get("PPF_BENCH_ONE_GPU", "0") != "0" else "nccl"
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)

    from protopformer_amd import _lib, ops
    from protopformer_amd.engine import GraphedTrainStep, ReplayedTrainStep, train_one_step
    # the reference seeds every rank with seed + rank (main.py:254) and relies on DDP's rank-0 broadcast (main.py:370) to make the
    # replicas identical; same here: make_grad_sync() broadcasts rank 0's parameters / optimizer state
    model, opt, crit, sync = build(cfg, device, seed=1028 + rank)
    g = torch.Generator(device=device).manual_seed(1028 + rank)
    img = torch.randn(batch, 3, 224, 224, device=device, generator=g)
    label = torch.randint(0, cfg["C"], (batch,), device=device, generator=g)

    graphed = None
    graph_note = "kernels enqueued from the host each step by the Python orchestration, two HIP streams"
    if args.graph:
        graphed = GraphedTrainStep(model, crit, opt, epoch=20, grad_sync=sync, warmup=2, adopt_inputs=True)     # the batch is resident
    replayed = None
    if not args.graph and not args.eager:
        # default: two eager steps, then one recorded step; every later step replays the recorded launches (same kernels, arguments,
        # streams and dependencies, enqueued by one ctypes call each: engine.ReplayedTrainStep) -- the full step is still executed
        replayed = ReplayedTrainStep(model, crit, opt, epoch=20, grad_sync=sync, warmup=2, adopt_inputs=True)
        graph_note = "kernels enqueued from the host each step from a recorded command list (engine.ReplayedTrainStep), two HIP streams"

    def step():
        if graphed is not None:
            return graphed(img, label)
        if replayed is not None:
            return replayed(img, label)
        return train_one_step(model, crit, img, label, opt, epoch=20, grad_sync=sync)

    done = 0
    if graphed is not None:
        try:
            for _ in range(3):                                        # two eager steps, then capture + first replay
                step(); done += 1
            torch.cuda.synchronize()
            graph_note = "one captured HIP graph per step (engine.GraphedTrainStep), replayed"
        except Exception as e:                                        # capture refused (e.g. a collective that cannot be captured)
            graph_note = f"host-enqueued (graph capture failed: {type(e).__name__}: {str(e)[:200]})"
            graphed = None
            torch.cuda.synchronize()
    for _ in range(max(0, args.warmup - done)):
        step()
    # roofline probe: HIP events around every launch of the dominant kernel (wgrad GEMM) on its launch stream.  Event-record nodes of
    # a captured graph cannot be read back on ROCm 7.2 (hipEventElapsedTime: invalid resource handle), so in --graph mode the probe
    # runs over 3 host-enqueued steps AFTER the timed region.
    if graphed is None and os.environ.get("PPF_BENCH_PROBE", "1") != "0":     # (0: A/B of what the probe's event records cost)
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
    # host cost of ONE step enqueued into an EMPTY queue (the figure above includes the time the host is blocked on a full queue
    # whenever the GPU is the slower side): median of 5
    one = []
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        one.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    one.sort()
    t_enqueue_one = one[len(one) // 2]
    probe_note = "HIP events around every launch inside the timed region"
    if graphed is None:
        _lib.call("ppf_gemm_probe", 0)
    else:
        _lib.call("ppf_gemm_probe", 1)
        for _ in range(3):
            train_one_step(model, crit, img, label, opt, epoch=20, grad_sync=sync)
        torch.cuda.synchronize()
        _lib.call("ppf_gemm_probe", 0)
        probe_note = "HIP events around every launch of 3 host-enqueued steps after the timed region (graph nodes cannot be timed)"
    # the kernels the north star names (attention forward / backward, prototype forward): in-step HIP-event time over 3 more steps of
    # the same kind AFTER the timed region (the ~50 extra event records per step must not touch `value`)
    # EVERY rank runs the three steps (they contain the gradient collectives); rank 0 reports its own kernels' times.
    named = None
    if os.environ.get("PPF_BENCH_PROBE", "1") != "0":
        _lib.call("ppf_path_probe", 1)
        for _ in range(3):
            step() if graphed is None else train_one_step(model, crit, img, label, opt, epoch=20, grad_sync=sync)
        torch.cuda.synchronize()
        _lib.call("ppf_path_probe", 0)
        if rank == 0:
            try:
                named = named_path_report(_lib, 3)
            except Exception as e:                                      # informational leg: never takes the line down
                named = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    import ctypes
    c_ms, c_n, c_fl, c_by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    try:
        _lib.call("ppf_gemm_probe_read", ctypes.addressof(c_ms), ctypes.addressof(c_n), ctypes.addressof(c_fl), ctypes.addressof(c_by))
    except RuntimeError as e:
        probe_note = f"probe read failed: {e}"
    kern_ms, n_launch, probe_flops, probe_bytes = c_ms.value, int(c_n.value), c_fl.value, c_by.value
    if rank == 0:
        ips = world * batch * args.steps / dt
        achieved = (probe_flops / 1e12) / (kern_ms / 1e3) if kern_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(args.config)
        ex = executed_gflop_per_img(cfg)
        out = {
            "metric": METRIC if args.config == "deit_small" and batch == 256 and args.addon == "regular" and not args.proto_dim else
                      f"images/sec train step, {cfg['arch']}+{cfg['P']} protos, bs{batch}, {world} MI355X ({cfg['label']}; not the headline metric)",
            "value": ips, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps, "host_enqueue_ms_one_step_empty_queue": 1e3 * t_enqueue_one,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg['arch']}, synthetic 224x224, {cfg['P']}x{cfg['Dp']} prototypes, {cfg['C']} classes, k={cfg['k']}, batch {batch}/GPU "
                                   f"({cfg['label']}), train step = fwd+CE+PPC+bwd+allreduce+AdamW+EMA, DropPath 0.1; blocks after the token "
                                   "reservation run compacted on the 1+k reserved rows (equal to the masked full-length blocks to bf16 rounding); "
                                   "bf16 backbone: the reserved tokens differ from the fp32 reference's in ~2-3 % of positions (any bf16 rounding upstream "
                                   "of the rollout's 90 % discard does that: profiles/r6_reserve_precision.txt; PPNet.precise = fp32 mode, bit-exact)",
                       "global_batch": world * batch, "parallelism": f"dp{world}", "execution": graph_note},
            "rccl_world": dist.get_world_size() if dist.is_initialized() else 1,
            "rccl_backend": dist.get_backend() if dist.is_initialized() else None,
            "roofline": {"bound": "mfma", "kernel": ops.DOMINANT_NAME, "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "traffic_unit": f"bytes/launch (PMC, see profiles/{traffic_src})" if traffic_src else None,
                         "algorithmic_bytes": probe_bytes / max(n_launch, 1), "launches": n_launch,
                         "avg_launch_ms": kern_ms / max(n_launch, 1),
                         "how": probe_note + "; the kernel runs on the weight-gradient stream concurrently with the main chain (contended time)"},
            "step_mfma_frac": (ips / world) * cfg["gflop"] / 1e3 / PEAK_BF16_TFLOPS,
            "step_mfma_frac_executed": (ips / world) * ex / 1e3 / PEAK_BF16_TFLOPS,
            "gflop_per_img": {"algorithmic": cfg["gflop"], "executed": ex},
            "final_loss": float(loss),
        }
        if named is not None:
            out["roofline"]["named_path"] = named
        if world == 1 and not args.no_cpu_baseline:
            # the dominant kernel alone on the chip, in a CHILD process after the timed region (its launches must not show up in a
            # kernel trace / counter pass of this process: profiles/ hold the in-step averages)
            try:
                del model, opt, crit, img, label, loss, step, replayed, graphed
                torch.cuda.empty_cache()
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", args.config, "--wgrad-alone"] +
                                   (["--batch", str(args.batch)] if args.batch else []), capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out["roofline"]["uncontended"] = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-160:]}
            except Exception as e:                                      # informational leg: never takes the line down
                out["roofline"]["uncontended"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
        if world == 1 and not args.no_secondary and not args.no_cpu_baseline and args.config == "deit_small" and args.batch is None:
            # BASELINE.json configs[1] and configs[4] (per-GPU shape), 20 steps each in a child process of their own AFTER the headline
            # measurement (this process has released its device memory; nothing below changes `value`)
            out["secondary"] = secondary_configs()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        line = json.dumps(out)
    else:
        line = None
    if dist.is_initialized():
        try:                                      # every rank empties its C stdio buffer (RCCL's banner) BEFORE rank 0 prints the line
            import ctypes
            ctypes.CDLL(None).fflush(None)
            if world > 1:
                dist.barrier()
        except Exception:
            pass
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes its version banner to the C library's stdout, where it sits in the buffer until exit -- BEHIND a line printed from Python (seen with
        # one forced rank in round 6: the JSON line came first, five banner lines after it).  The contract is ONE JSON line, and a consumer that takes the
        # last line of stdout must find it: flush C stdio first, print the line last.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
