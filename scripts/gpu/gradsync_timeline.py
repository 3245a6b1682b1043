"""When does each gradient chunk become ready, relative to the end of backward?  

One rank, backend nccl (= RCCL), PPF_FORCE_GRADSYNC=1, the bench workload (configs[2] by default) replayed from the recorded command
list; GradSync._exchange is wrapped with timing events on the communication stream: `ready` (the stream has waited for every kernel that
writes the chunk) and `done`; `bwd_end` is recorded on the main stream where finish() starts waiting.  With one rank the collective itself
is empty, so `ready` times are exactly the hardware readiness of the chunks; the 8-GPU exposure is then MODELLED: chunk exchanges are
serialised on the communication stream, each takes ALPHA + bytes / ALGBW (ring all-reduce over xGMI: ALGBW = busbw * n / (2 (n - 1))),
exposed time = completion of the last chunk - bwd_end.  ONE partition per process (a second model in the same process measured 30 % slower
steps, whatever its partition): PLAN = equal (the pre-round-6 three equal block groups) | default (engine.readiness_cuts) | a cut list "1+2+4+8";
PAYLOAD = fp32 | bf16.
    python scripts/gpu/gradsync_timeline.py PLAN PAYLOAD [config] [busbw GB/s] [alpha us]"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29579")
os.environ["PPF_FORCE_GRADSYNC"] = "1"
import torch
import torch.distributed as dist

import bench
from protopformer_amd import backbone, engine
from protopformer_amd.engine import ReplayedTrainStep

PLAN = sys.argv[1] if len(sys.argv) > 1 else "default"
PAYLOAD = sys.argv[2] if len(sys.argv) > 2 else "fp32"
name = sys.argv[3] if len(sys.argv) > 3 else "deit_small"
BUSBW = float(sys.argv[4]) if len(sys.argv) > 4 else 300.0           # GB/s, large-message ring all-reduce on 8 GPUs (7 xGMI links x 153 GB/s per GPU)
ALPHA = float(sys.argv[5]) if len(sys.argv) > 5 else 30.0            # us per collective (launch + 2 (n - 1) hops)
NGPU = 8
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1, device_id=dev)


class Timed(engine.GradSync):
    log = None

    def _exchange(self, lo, hi):
        if Timed.log is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream())
            super()._exchange(lo, hi)
            b.record(torch.cuda.current_stream())
            Timed.log.append((lo, hi, a, b))
        else:
            super()._exchange(lo, hi)

    def finish(self):
        if Timed.log is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            Timed.log.append(("bwd_end", e))
        return super().finish()


engine.GradSync = Timed


def measure(label, payload, **plan):
    backbone._KEEP_CACHE.clear()
    torch.manual_seed(0)
    model, opt, crit, _ = bench.build(cfg, dev, 0)
    sync = engine.make_grad_sync(model, opt, payload=payload, **plan)
    B = cfg["batch"]
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    lab = torch.randint(0, cfg["C"], (B,), device=dev, generator=g)
    step = ReplayedTrainStep(model, crit, opt, epoch=20, grad_sync=sync, warmup=2, adopt_inputs=True)
    for _ in range(6):
        step(img, lab)
    torch.cuda.synchronize()
    rows = []
    steps = 5
    tot = 0.0
    for _ in range(steps):
        Timed.log = []
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        step(img, lab)
        t1.record()
        torch.cuda.synchronize()
        log, Timed.log = Timed.log, None
        tot += t0.elapsed_time(t1)
        end = next(x[1] for x in log if x[0] == "bwd_end")
        rows.append([(lo, hi, t0.elapsed_time(a), t0.elapsed_time(b), t0.elapsed_time(end)) for lo, hi, a, b in (x for x in log if x[0] != "bwd_end")])
    n = len(rows[0])
    elt = 2 if payload == "bf16" else 4
    algbw = BUSBW * NGPU / (2 * (NGPU - 1))                            # GB/s
    print(f"== {name} {label} payload={payload}: step {tot / steps:.3f} ms (single rank, collectives forced), {n} chunks; model: {NGPU} GPUs, busbw {BUSBW:.0f} GB/s -> algbw {algbw:.0f} GB/s, alpha {ALPHA:.0f} us")
    print("   chunk   MB(fp32)   ready ms   ready - bwd_end ms   modelled exchange us   modelled done - bwd_end ms")
    t_free, out = 0.0, []
    for c in range(n):
        lo, hi = rows[0][c][0], rows[0][c][1]
        ready = sum(r[c][2] for r in rows) / steps
        bend = sum(r[c][4] for r in rows) / steps
        dur = ALPHA + (hi - lo) * elt / (algbw * 1e3)                  # us
        start = max(ready, t_free)
        t_free = start + dur / 1e3
        out.append(dict(chunk=c, mb=(hi - lo) * 4 / 1e6, ready_ms=ready, ready_rel_ms=ready - bend, xfer_us=dur, done_rel_ms=t_free - bend))
        print(f"   {c:5d} {(hi - lo) * 4 / 1e6:10.2f} {ready:10.3f} {ready - bend:20.3f} {dur:22.0f} {t_free - bend:28.3f}")
    exposed = max(0.0, out[-1]["done_rel_ms"])
    print(f"   modelled exposed exchange after backward: {exposed * 1e3:.0f} us = {100 * exposed / (tot / steps):.2f} % of the step")
    del model, opt, step, sync
    torch.cuda.empty_cache()
    return dict(label=label, payload=payload, step_ms=tot / steps, exposed_us=exposed * 1e3, chunks=out)


plan = dict(n_chunks=4) if PLAN == "equal" else ({} if PLAN == "default" else dict(cuts=[int(v) for v in PLAN.split("+")]))
res = measure(f"plan {PLAN}", PAYLOAD, **plan)
print("GRADSYNC_TIMELINE " + json.dumps(res))
dist.destroy_process_group()
