"""Train-step loop of the hot path (tools/engine_proto.py:41-81; optimizer groups tools/create_optimizer.py:27-39;
schedule tools/create_scheduler.py:20-32; checkpoint layout main.py:393-407, 436-471).

* FlatAdamW          : torch.optim.AdamW semantics as ONE fused HIP kernel over the flat (p, g, m, v[, ema]) buffers that also
                       re-emits the bf16 weight shadow; param_groups expose lr / weight_decay so schedulers work unchanged;
                       state_dict()/load_state_dict() speak torch.optim.AdamW's per-parameter format (reference --resume files).
* CosineLRScheduler  : the timm scheduler the reference builds (warm-up + cosine, stepped per epoch).
* GradSync           : data-parallel gradient exchange = chunked RCCL all-reduce of the flat gradient buffer on a side
                       stream, launched as soon as a chunk's layers have finished their backward (overlap), averaged by
                       folding 1/world into the optimizer kernel; replicas are made identical by a rank-0 broadcast.
* train_one_step / GraphedTrainStep / train_one_epoch : the reference's step body (forward, CE + PPC, backward, clip, step,
                       EMA), eagerly or as one captured HIP graph that is replayed per step.
* save_checkpoint / load_checkpoint : the reference's checkpoint dict {model, optimizer, lr_scheduler, epoch, model_ema}.
"""
import math
import os
import sys

import torch
import torch.distributed as dist

from . import _lib, ops
from .protopformer import CrossEntropyLoss, WeightedLossFn

DEFAULT_LRS = {"features": 1e-4, "add_on_layers": 3e-3, "prototype_vectors": 3e-3}       # main.py:64-66

# device-resident step state of ppf_adamw_step_dev (include/ppf_hip.h)
_HY_LR, _HY_WD, _HY_BC1, _HY_BC2, _HY_GSCALE, _HY_CLIP, _HY_N = 0, 8, 16, 17, 18, 19, 20


def _unwrap(model):
    """Data parallelism here is GradSync on the flat gradient buffer; a DistributedDataParallel wrapper would see every
    parameter as unused (the kernels write gradients straight into the flat buffer) and silently skip the reduction."""
    if isinstance(model, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)):
        raise RuntimeError("protopformer_amd: do not wrap the model in DistributedDataParallel / DataParallel -- gradients are written into "
                           "a flat buffer by the HIP kernels and exchanged by engine.GradSync (pass grad_sync=make_grad_sync(model, optimizer))")
    return model


class FlatAdamW:
    def __init__(self, ppnet, joint_optimizer_lrs=None, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, ema_decay=None):
        ppnet = _unwrap(ppnet)
        lrs = dict(DEFAULT_LRS if joint_optimizer_lrs is None else joint_optimizer_lrs)
        self.ppnet = ppnet
        self.store = ppnet.flat_store()
        st = self.store
        # the reference's four groups, in flat-buffer order (create_optimizer.py:31-39)
        wds = {"features": 1e-3, "add_on_layers": 1e-3, "prototype_vectors": weight_decay, "prototype_vectors_global": weight_decay}
        lr_of = {"features": lrs["features"], "add_on_layers": lrs["add_on_layers"], "prototype_vectors": lrs["prototype_vectors"],
                 "prototype_vectors_global": lrs["prototype_vectors"]}
        self.betas, self.eps = betas, eps
        self.param_groups = []
        for n, b, e in st.segments:
            params = [p for _, p, o, _ in st.entries if b <= o < e]
            self.param_groups.append(dict(name=n, lr=lr_of[n], weight_decay=wds[n], begin=b, end=e, initial_lr=lr_of[n], params=params,
                                          betas=betas, eps=eps, amsgrad=False))
        self.exp_avg = ops.zeros(st.params.shape, torch.float32, st.device)
        self.exp_avg_sq = ops.zeros(st.params.shape, torch.float32, st.device)
        self.ema = st.params.clone() if ema_decay is not None else None
        self.ema_decay = 0.0 if ema_decay is None else float(ema_decay)
        self.step_count = 0
        self.grad_scale = 1.0
        self._bounds = torch.tensor([g["begin"] for g in self.param_groups] + [self.param_groups[-1]["end"]], dtype=torch.int64)
        # step-dependent scalars live in device memory (refreshed from this host mirror before every step), so that the
        # optimizer launch itself is step-independent and a captured graph of the step can be replayed
        self._hyper_host = torch.zeros(_HY_N, dtype=torch.float32)
        self._hyper_host[_HY_CLIP] = 1.0
        self._hyper = torch.empty(_HY_N, dtype=torch.float32, device=st.device)
        _lib.call("ppf_hyper_set", self._hyper, self._hyper_host.data_ptr(), _HY_N)
        self._clip_partial = None
        self._clip_issued = False                 # ppf_clip_grad_scale wrote hyper[19] for the step about to be applied
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=st.device)
        # set by the optimizer kernel when a step's loss was NaN / inf (that step is then skipped): engine_proto.py:66-70 without a sync
        self.nonfinite = ops.zeros((1,), torch.int32, st.device)

    # ------------------------------------------------------------------ consistency with the model's flat store
    def _check_store(self):
        if self.ppnet.flat_store() is not self.store:
            raise RuntimeError("FlatAdamW: the model's flat parameter store was rebuilt after this optimizer was created (model.to()/"
                               ".cuda()/.float() re-materialised the parameters); create the optimizer after moving the model")

    def zero_grad(self, set_to_none=False):
        self._check_store()
        self.store.zero_grad()

    # ------------------------------------------------------------------ the step
    def refresh_hyper(self):
        """Advance the step count and push lr / weight decay / bias corrections / gradient scale to device memory (a one-wave kernel
        that carries the values as launch arguments).  GraphedTrainStep calls this right before each replay."""
        self.step_count += 1
        h = self._hyper_host
        for i, g in enumerate(self.param_groups):
            h[_HY_LR + i] = g["lr"]
            h[_HY_WD + i] = g["weight_decay"]
        h[_HY_BC1] = 1.0 - self.betas[0] ** self.step_count
        h[_HY_BC2] = math.sqrt(1.0 - self.betas[1] ** self.step_count)
        h[_HY_GSCALE] = float(self.grad_scale)
        # hyper[19] (clip coefficient) is written on the device by ppf_clip_grad_scale; a step without clipping must not inherit the
        # previous step's coefficient: the upload then covers it too (host mirror value 1.0)
        n = _HY_CLIP if self._clip_issued else _HY_N
        self._clip_issued = False
        _lib.call("ppf_hyper_set", self._hyper, h.data_ptr(), n)

    def clip_grad_norm(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ over all trainable gradients (timm dispatch_clip_grad mode 'norm'), folded into the
        optimizer kernel as a coefficient in device memory; self.grad_norm holds the (averaged-gradient) norm."""
        if self._clip_partial is None:
            self._clip_partial = torch.empty(_lib.lib().ppf_clip_grad_blocks(), dtype=torch.float32, device=self.store.device)
        _lib.call("ppf_clip_grad_scale", self.store.grads, self.store.total, float(max_norm), float(self.grad_scale), self._clip_partial,
                  self._hyper, self.grad_norm)
        self._clip_issued = True

    def launch_update(self, loss_guard=None):
        """loss_guard: the step's loss (device scalar).  A non-finite loss then skips the update and raises self.nonfinite."""
        st = self.store
        _lib.call("ppf_adamw_step_guarded", st.params, st.grads, self.exp_avg, self.exp_avg_sq, self.ema, st.bf16, st.total, len(self.param_groups),
                  self._bounds.data_ptr(), self._hyper, self.betas[0], self.betas[1], self.eps, self.ema_decay,
                  loss_guard, self.nonfinite if loss_guard is not None else None)
        st.mark_bf16_written()                    # the kernel re-emitted the bf16 shadow (its transposed copies are now stale)

    def step(self, loss_guard=None):
        self._check_store()
        clip = self._clip_issued
        # the step-dependent scalars are read on the host NOW: inside a recorded step (engine.ReplayedTrainStep) this is a live entry
        _lib.run_live(lambda: self._live_refresh(clip))
        self.launch_update(loss_guard)

    def _live_refresh(self, clip_issued):
        self._clip_issued = clip_issued           # a replay re-runs the recorded clip kernel in front of this point
        self.refresh_hyper()

    # ------------------------------------------------------------------ checkpoint format: torch.optim.AdamW's
    def state_dict(self):
        """The dict torch.optim.AdamW.state_dict() would give for the reference's four groups (create_optimizer.py:31-39):
        per-parameter {'step', 'exp_avg', 'exp_avg_sq'} keyed by the parameter's index, and the groups with index lists."""
        state, groups, i = {}, [], 0
        for g in self.param_groups:
            ids = []
            for _, p, o, n in self.store.entries:
                if not (g["begin"] <= o < g["end"]):
                    continue
                if self.step_count > 0:
                    state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[o:o + n].view(p.shape).clone(),
                                    exp_avg_sq=self.exp_avg_sq[o:o + n].view(p.shape).clone())
                ids.append(i)
                i += 1
            groups.append(dict(lr=g["lr"], betas=tuple(self.betas), eps=self.eps, weight_decay=g["weight_decay"], amsgrad=False,
                               initial_lr=g["initial_lr"], params=ids))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        if "exp_avg" in sd and "state" not in sd:           # round-1 flat format
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_count = int(sd["step"])
            if self.ema is not None and sd.get("ema") is not None:
                self.ema.copy_(sd["ema"])
            for g, s in zip(self.param_groups, sd["param_groups"]):
                g.update(lr=s["lr"], weight_decay=s["weight_decay"])
            return
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError(f"optimizer checkpoint has {len(groups)} param groups, expected {len(self.param_groups)}")
        steps = set()
        i = 0
        for g, s in zip(self.param_groups, groups):
            mine = [(p, o, n) for _, p, o, n in self.store.entries if g["begin"] <= o < g["end"]]
            if len(mine) != len(s["params"]):
                raise ValueError(f"group {g['name']}: checkpoint has {len(s['params'])} parameters, model has {len(mine)}")
            for (p, o, n), pid in zip(mine, s["params"]):
                ent = sd["state"].get(pid, sd["state"].get(str(pid)))
                if ent is None:
                    self.exp_avg[o:o + n].zero_(); self.exp_avg_sq[o:o + n].zero_()
                    steps.add(0)
                    continue
                if tuple(ent["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {pid}: shape {tuple(ent['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
                self.exp_avg[o:o + n].copy_(ent["exp_avg"].reshape(-1)); self.exp_avg_sq[o:o + n].copy_(ent["exp_avg_sq"].reshape(-1))
                steps.add(int(float(ent["step"])))
                i += 1
            g.update(lr=s["lr"], weight_decay=s["weight_decay"])
            if "initial_lr" in s:
                g["initial_lr"] = s["initial_lr"]
        if len(steps) > 1:
            raise ValueError(f"the fused kernel keeps one step count for all parameters; checkpoint has {sorted(steps)}")
        self.step_count = steps.pop() if steps else 0

    def ema_state_dict(self):
        """timm get_state_dict(model_ema) (main.py:444): the EMA weights under the model's state-dict keys; frozen entries
        (ones, last_layer*.weight) are copied from the model -- they never change, so their average equals them."""
        out = {k: v.detach().clone() for k, v in self.ppnet.state_dict().items()}
        if self.ema is not None:
            for name, p, o, n in self.store.entries:
                out[name] = self.ema[o:o + n].view(p.shape).clone()
        return out

    def load_ema_state_dict(self, sd):
        """utils._load_checkpoint_for_ema (main.py:405)."""
        if self.ema is None:
            return
        for name, p, o, n in self.store.entries:
            if name in sd:
                self.ema[o:o + n].copy_(sd[name].reshape(-1))


class CosineLRScheduler:
    """timm.scheduler.CosineLRScheduler as tools/create_scheduler.py:20-32 configures it (t_in_epochs, cycle_limit 1, no noise,
    warmup_prefix False, cycle_mul 1, cycle_decay 1): written from the timm 0.5.4 API -- timm is not vendored in the reference
    and not installed here, so this boundary is unpinned (tests pin it to the closed-form table).

        t <  warmup_t : lr_g = warmup_lr_init + t * (base_g - warmup_lr_init) / warmup_t
        t >= warmup_t : lr_g = lr_min + 0.5 * (base_g - lr_min) * (1 + cos(pi * t / t_initial))      for t < t_initial
                        lr_g = lr_min                                                                 afterwards (cool-down)

    base_g is each group's `initial_lr`.  Construction sets every group to warmup_lr_init; main.py:434 calls step(epoch) AFTER
    epoch `epoch` has trained (so epochs 0 and 1 both run at warmup_lr_init -- the reference's off-by-one, reproduced by
    keeping the same call)."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_lr_init=0.0, warmup_t=0, cycle_limit=1, t_in_epochs=True, **_unused):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_lr_init, self.warmup_t = int(t_initial), float(lr_min), float(warmup_lr_init), int(warmup_t)
        self.cycle_limit, self.t_in_epochs = int(cycle_limit), bool(t_in_epochs)
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        if self.warmup_t:
            self.warmup_steps = [(v - self.warmup_lr_init) / self.warmup_t for v in self.base_values]
            self._update([self.warmup_lr_init] * len(self.base_values))
        else:
            self.warmup_steps = [1.0 for _ in self.base_values]

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        i = t // self.t_initial
        t_curr = t - self.t_initial * i
        if i < self.cycle_limit:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1.0 + math.cos(math.pi * t_curr / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def get_cycle_length(self, cycles=0):
        return self.t_initial * max(1, cycles or self.cycle_limit)

    def _update(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def step(self, epoch, metric=None):
        if self.t_in_epochs:
            self._update(self._get_lr(epoch))

    def step_update(self, num_updates, metric=None):
        if not self.t_in_epochs:
            self._update(self._get_lr(num_updates))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def create_scheduler(args, optimizer):
    """tools/create_scheduler.py:4-36 for --sched cosine (the only schedule scripts/train_*.sh use).  Returns (scheduler, num_epochs)."""
    if getattr(args, "sched", "cosine") != "cosine":
        raise NotImplementedError("only the cosine schedule of scripts/train_cub.sh is on the path")
    if getattr(args, "lr_noise", None) is not None:
        raise NotImplementedError("lr noise is off in every reference script")
    s = CosineLRScheduler(optimizer, t_initial=args.epochs, lr_min=getattr(args, "min_lr", 1e-5), warmup_lr_init=getattr(args, "warmup_lr", 1e-6),
                          warmup_t=getattr(args, "warmup_epochs", 5), cycle_limit=getattr(args, "lr_cycle_limit", 1), t_in_epochs=True)
    return s, s.get_cycle_length() + getattr(args, "cooldown_epochs", 10)


class GradSync:
    """Chunked all-reduce (sum) of a flat gradient buffer over the default process group, overlapped with backward.

    payload='bf16' (or PPF_GRADSYNC_BF16=1): the chunks travel as bf16 -- half the link bytes (46.7 instead of 93.4 MB per step at
    deit_small) for one rounding of every SUMMAND and of every partial sum on the wire (relative 2^-9 each; the fp32 buffer receives the
    widened result).  Narrowing / widening run on the communication stream (ppf_cast_f32_bf16 / ppf_cast_bf16_f32), so they cost HBM
    traffic (3 x the chunk) but no compute-stream time.  The reference exchanges fp32 (DistributedDataParallel, main.py:369-371): default."""

    def __init__(self, flat_grads, chunk_bounds, use_side_stream=True, payload=None):
        self.g = flat_grads
        self.bounds = list(chunk_bounds)                      # ascending element offsets; chunk c = [bounds[c], bounds[c+1])
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.force = os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0"     # single-rank runs still issue the collectives (tests)
        self.cuda = flat_grads.is_cuda
        self.stream = torch.cuda.Stream() if (self.cuda and use_side_stream) else None
        self.pending = []
        self.guard = None                                     # the loss summed over the ranks (reduce_guard)
        self.launched = 0                                     # collectives issued so far (tests: the forced single-rank path really ran)
        if payload is None:
            payload = "bf16" if os.environ.get("PPF_GRADSYNC_BF16", "0") != "0" else "fp32"
        if payload not in ("fp32", "bf16"):
            raise ValueError(f"GradSync payload must be 'fp32' or 'bf16', got {payload!r}")
        self.payload = payload
        self.wire = torch.empty(flat_grads.numel(), dtype=torch.bfloat16, device=flat_grads.device) if payload == "bf16" else None

    def _narrow(self, lo, hi):
        if self.cuda:
            _lib.call("ppf_cast_f32_bf16", self.g[lo:hi], self.wire[lo:hi], hi - lo)      # (on the override stream pushed by the caller)
        else:
            self.wire[lo:hi].copy_(self.g[lo:hi])

    def _widen(self, lo, hi):
        if self.cuda:
            _lib.call("ppf_cast_bf16_f32", self.wire[lo:hi], self.g[lo:hi], hi - lo)
        else:
            self.g[lo:hi].copy_(self.wire[lo:hi])

    def _exchange(self, lo, hi):
        """The collective of one chunk on the CURRENT torch stream (the communication stream when there is one)."""
        if self.wire is None:
            self.pending.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))      # waited for in finish()
            return
        if self.cuda and (lo % 8 or hi % 8):
            raise ValueError("GradSync bf16 payload: chunk bounds must be multiples of 8 elements (FlatStore segments are)")
        raw = torch.cuda.current_stream().cuda_stream if self.cuda else None
        if raw is not None:
            _lib.push_stream(raw)
        try:
            self._narrow(lo, hi)
            w = dist.all_reduce(self.wire[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
            w.wait()                                          # NCCL: the current stream waits (no host block); gloo: the host waits
            self._widen(lo, hi)
        finally:
            if raw is not None:
                _lib.pop_stream()

    def chunk_ready(self, c, also=()):
        """Launch the all-reduce of chunk c: every kernel that writes it has been enqueued on the current stream or on one of
        the `also` streams (the weight-gradient lane) -- the communication stream waits for all of them, the compute streams
        never wait for each other here."""
        if self.world == 1 and not self.force:
            return
        lo, hi = self.bounds[c], self.bounds[c + 1]
        if hi <= lo:
            return
        self.launched += 1
        if self.stream is not None:
            evs = []
            for st in (torch.cuda.current_stream(),) + tuple(s_ for s_ in also if s_ is not None):
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            with torch.cuda.stream(self.stream):
                for ev in evs:
                    self.stream.wait_event(ev)
                self._exchange(lo, hi)
        else:
            for st in also:
                if st is not None:
                    torch.cuda.current_stream().wait_stream(st)
            self._exchange(lo, hi)

    def reduce_guard(self, loss):
        """The device-side non-finite stop (ppf_adamw_step_guarded) must take the same decision on every rank: returns a copy of the loss
        scalar that is being summed over the ranks on the communication stream (a NaN / inf on ANY rank makes the sum non-finite, so every
        rank skips that update -- otherwise the rank with the bad loss would skip while the others apply the all-reduced NaN gradients).
        Single rank without PPF_FORCE_GRADSYNC: the loss itself.  finish() waits for it with the gradient chunks."""
        if self.world == 1 and not self.force:
            return loss.detach().reshape(1)
        if self.guard is None:                                # one buffer for the life of the object: a recorded step binds its address
            self.guard = torch.zeros(1, dtype=torch.float32, device=loss.device)
        g = self.guard
        g.copy_(loss.detach().reshape(1))
        self.launched += 1
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self.pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        return g

    def finish(self):
        """Wait for all chunks; returns the scale (1/world) the optimizer must apply to the summed gradients."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return 1.0 / self.world


def broadcast_replica_state(ppnet, optimizer=None, src=0):
    """What DistributedDataParallel does at wrap time (main.py:370): every rank starts from rank `src`'s parameters and buffers
    -- the reference seeds each rank differently (main.py:254 seed + rank), so without this the replicas would apply averaged
    gradients to diverging weights.  Also aligns the optimizer moments / EMA copy."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    st = ppnet.flat_store()
    with torch.no_grad():
        dist.broadcast(st.params, src)
        for name, t in ppnet.state_dict().items():
            if not any(t.data_ptr() == p.data_ptr() for _, p, _, _ in st.entries):          # frozen tensors: ones, last_layer*.weight
                dist.broadcast(t, src)
        if optimizer is not None:
            dist.broadcast(optimizer.exp_avg, src)
            dist.broadcast(optimizer.exp_avg_sq, src)
            if optimizer.ema is not None:
                dist.broadcast(optimizer.ema, src)            # rank src's EMA survives (a checkpoint restored before make_grad_sync)
    st.invalidate()


def readiness_cuts(depth):
    """Blocks at which a new gradient chunk starts.  Backward completes the blocks depth-1 .. 0 and the weight-gradient lane lags the main chain,
    so the chunk that becomes ready LAST is the one whose exchange the optimizer waits for: it is made the smallest -- block 0 + the embedding
    (8.6 of 93.4 MB at deit_small) -- and the rest of the blocks is cut in two: [final norm + heads] | [upper half] | [blocks 1 .. half) |
    [embedding + block 0].  Measured on one MI355X (profiles/r6_gradsync.txt): every collective costs ~0.25 % of the step on the launch side even with
    zero link time, so FEW chunks; the r5 partition (three equal block groups) left 29.9 MB to exchange after the last backward kernel (modelled
    193 us = 1.3 % of the step at 8 GPUs), this one leaves it 0.07 ms of slack with the same number of collectives.  A finer geometric split
    ({1, 2, 4, 8}) hides no more and costs +0.5 %."""
    if depth < 2:
        return []
    return sorted({1, (depth + 1) // 2})


def make_grad_sync(ppnet, optimizer=None, n_chunks=None, cuts=None, payload=None):
    """Chunk the flat gradient in backward-completion order: [heads + norm | late blocks | ... | block 1 | embedding + block 0], after making
    the replicas identical (rank-0 broadcast of parameters, frozen tensors and optimizer state).
    cuts: block indices that start a chunk (default readiness_cuts(depth); PPF_GRADSYNC_CUTS="8,4" or "8+4" overrides); n_chunks: the pre-round-6
    equal partition into n_chunks - 1 block groups (kept for A/B)."""
    ppnet = _unwrap(ppnet)
    broadcast_replica_state(ppnet, optimizer)
    st = ppnet.flat_store()
    env = os.environ.get("PPF_GRADSYNC_CUTS")
    if cuts is None and env:
        cuts = [int(v) for v in env.replace("+", ",").split(",") if v.strip()]
    bounds, block_chunk = chunk_plan([(name, o) for name, p, o, n in st.entries], st.total, cuts=cuts, n_chunks=n_chunks)
    sync = GradSync(st.grads, bounds, payload=payload)
    sync.block_chunk = block_chunk                    # chunk that becomes complete when block i's backward has been enqueued (it STARTS at block i)
    sync.tail_chunk = len(bounds) - 2                 # final norm + add-on + prototypes: ready first
    sync.head_chunk = 0                               # embedding + block 0 (.. first cut): ready last
    return sync


def chunk_plan(entries, total, cuts=None, n_chunks=None):
    """(bounds, block_chunk) of the flat gradient buffer whose segments are `entries` = [(parameter name, element offset)] in flat order
    ([embedding | blocks 0 .. depth-1 | (class-attention blocks) | final norm + heads]): ascending chunk bounds and {block index that starts a
    chunk: chunk index}.  Pure (tests/test_dist_cpu.py)."""
    block_offsets = [o for name, o in entries if name.startswith("features.blocks.") and name.split(".")[3] == "norm1" and name.endswith("weight")]
    depth = len(block_offsets)
    if cuts is None and n_chunks is not None:
        per = max(1, math.ceil(depth / max(1, n_chunks - 1)))
        cuts = list(range(per, depth, per))
    if cuts is None:
        cuts = readiness_cuts(depth)
    cuts = sorted({int(c) for c in cuts if 0 < int(c) < depth})
    norm_off = next(o for name, o in entries if name == "features.norm.weight")
    bounds = sorted({0, norm_off, total} | {block_offsets[c] for c in cuts})
    return bounds, {i: bounds.index(block_offsets[i]) for i in cuts}


def _forward_backward(model, criterion, samples, targets, optimizer, epoch, ppc_cov_coe, ppc_mean_coe, use_ppc_loss, grad_sync, check_finite,
                      max_norm):
    outputs, auxi = model(samples)
    loss = criterion(outputs, targets)
    cov = mean = None
    if use_ppc_loss:
        cov, mean = model.get_PPC_loss(auxi[2], auxi[3], auxi[4], targets)
        if epoch >= 20:                                           # engine_proto.py:61-64
            loss = WeightedLossFn.apply(loss, cov, mean, ppc_cov_coe, ppc_mean_coe)
    if check_finite and not math.isfinite(loss.item()):           # engine_proto.py:66-70 (host sync: opt-in)
        print("Loss is {}, stopping training".format(loss.item()))
        sys.exit(1)
    optimizer.zero_grad()
    guard = loss.detach().reshape(1)
    if grad_sync is not None:
        # the guard of the device-side non-finite stop: the loss summed over the ranks (in place, on the communication stream, under backward)
        box = [guard]
        _lib.run_live(lambda: box.__setitem__(0, grad_sync.reduce_guard(loss)))
        guard = box[0]
    model._grad_sync = grad_sync
    loss.backward(gradient=ops.const_scalar(loss.device, 1.0))    # cached seed: no ones_like fill, and the loss Fns skip their scaling
    model._grad_sync = None
    if grad_sync is not None:
        _lib.run_live(lambda: setattr(optimizer, "grad_scale", grad_sync.finish()))
    if max_norm is not None:                                      # loss_scaler(..., clip_grad=max_norm) (engine_proto.py:74-76)
        optimizer.clip_grad_norm(max_norm)
    return loss, cov, mean, guard


def train_one_step(model, criterion, samples, targets, optimizer, epoch=20, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True,
                   grad_sync=None, check_finite=False, max_norm=None):
    """One iteration of tools/engine_proto.py:41-81 (without logging). Returns the detached loss tensors.
    The reference's NativeScaler (fp16 loss scaling) has no role here: the kernels accumulate in fp32 from bf16 operands."""
    _unwrap(model)
    loss, cov, mean, guard = _forward_backward(model, criterion, samples, targets, optimizer, epoch, ppc_cov_coe, ppc_mean_coe, use_ppc_loss,
                                               grad_sync, check_finite, max_norm)
    optimizer.step(loss_guard=guard)
    return loss.detach(), (cov.detach() if cov is not None else None), (mean.detach() if mean is not None else None)


class GraphedTrainStep:
    """The same step as ONE captured HIP graph (both compute streams and, under data parallelism, the RCCL collectives included).

    ~400 kernel launches per step cost 12-19 ms of host time through ctypes; a replay costs one launch.  Everything that changes from
    step to step lives in device memory: the batch (static input buffers), the AdamW step state (FlatAdamW.refresh_hyper copies
    lr / weight decay / bias corrections before each replay) and the DropPath step counter (advanced by the kernel itself).
    The first `warmup` calls run eagerly (they also size the split-K workspaces); the next call captures; later calls replay.
    Outputs are the graph's static loss tensors (overwritten by the next replay)."""

    def __init__(self, model, criterion, optimizer, epoch=20, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True, grad_sync=None,
                 max_norm=None, warmup=2, adopt_inputs=False):
        self.model, self.criterion, self.optimizer = _unwrap(model), criterion, optimizer
        self.kw = dict(epoch=epoch, ppc_cov_coe=ppc_cov_coe, ppc_mean_coe=ppc_mean_coe, use_ppc_loss=use_ppc_loss, grad_sync=grad_sync,
                       check_finite=False, max_norm=max_norm)
        self.grad_sync = grad_sync
        self.warmup = warmup
        self.adopt_inputs = adopt_inputs          # True: the first captured batch's tensors BECOME the static buffers (resident data)
        self.calls = 0
        self.graph = None
        self.static_in = None
        self.out = None

    def check_matches(self, epoch, ppc_cov_coe, ppc_mean_coe, use_ppc_loss, max_norm):
        """The captured graph bakes in the `epoch >= 20` branch (PPC terms added or not, engine_proto.py:61-64), the PPC coefficients
        and whether gradients are clipped.  A step function built for one phase must not be replayed in the other: raise, so that the
        caller builds a new GraphedTrainStep (e.g. one for epochs < 20 and one from epoch 20 on)."""
        kw = self.kw
        want = dict(ppc_branch=bool(use_ppc_loss) and epoch >= 20, use_ppc_loss=bool(use_ppc_loss), max_norm=max_norm)
        have = dict(ppc_branch=bool(kw["use_ppc_loss"]) and kw["epoch"] >= 20, use_ppc_loss=bool(kw["use_ppc_loss"]), max_norm=kw["max_norm"])
        if want["ppc_branch"]:
            want.update(ppc_cov_coe=float(ppc_cov_coe), ppc_mean_coe=float(ppc_mean_coe))
            have.update(ppc_cov_coe=float(kw["ppc_cov_coe"]), ppc_mean_coe=float(kw["ppc_mean_coe"]))
        if want != have:
            raise RuntimeError(f"GraphedTrainStep was built for {have} but train_one_epoch(epoch={epoch}) needs {want}: "
                               "build a new GraphedTrainStep for this phase (the branch is baked into the captured graph)")

    def _capture(self, samples, targets):
        opt = self.optimizer
        self.static_in = (samples, targets) if self.adopt_inputs else (samples.clone(), targets.clone())
        if self.grad_sync is not None:
            opt.grad_scale = 1.0 / self.grad_sync.world
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss, cov, mean, guard = _forward_backward(self.model, self.criterion, self.static_in[0], self.static_in[1], opt, **self.kw)
            opt.launch_update(guard)
            self.out = (loss.detach(), cov.detach() if cov is not None else None, mean.detach() if mean is not None else None)

    def __call__(self, samples, targets):
        opt = self.optimizer
        opt._check_store()
        self.calls += 1
        if self.graph is None:
            if self.calls <= self.warmup:
                return train_one_step(self.model, self.criterion, samples, targets, opt, **self.kw)
            self._capture(samples, targets)
        if samples.shape != self.static_in[0].shape or targets.shape != self.static_in[1].shape or samples.dtype != self.static_in[0].dtype:
            # a batch the list was not recorded for (a short last batch): the eager step, same kernels and streams, nothing recorded
            return train_one_step(self.model, self.criterion, samples, targets, opt, **self.kw)
        if samples.data_ptr() != self.static_in[0].data_ptr():
            self.static_in[0].copy_(samples, non_blocking=True)
        if targets.data_ptr() != self.static_in[1].data_ptr():
            self.static_in[1].copy_(targets, non_blocking=True)
        self._refresh_shadow()
        opt.refresh_hyper()
        self.graph.replay()
        return self.out

    def _refresh_shadow(self):
        """The frozen step starts from the bf16 weight shadow its own optimizer kernel left behind; the cast of TokensFn.forward is not
        part of it (the shadow was fresh when the step was captured / recorded).  After model.load_state_dict(), load_checkpoint,
        broadcast_replica_state or any other edit that invalidated the store, re-cast eagerly in front of the replay (the frozen
        step's own W^T refresh follows it)."""
        st = self.optimizer.store
        if not st.bf16_fresh:
            st.refresh_bf16()


class ReplayedTrainStep(GraphedTrainStep):
    """The same step as a recorded COMMAND LIST (protopformer_amd/_lib.py Recorder): after `warmup` eager calls, one step is executed
    eagerly while every library call (kernel launch, stream dependency) is recorded with its raw arguments; later calls replay the
    list -- one pre-bound ctypes call per launch instead of the Python orchestration, allocation and autograd of the eager step
    (~20 us -> ~3 us of host time per launch).  Unlike a HIP graph the replay enqueues ordinary launches on the two ordinary streams,
    so the overlap of the eager step is kept.  The recorded step's tensors become static buffers (inputs are copied in, the returned
    loss tensors are overwritten by the next call); optimizer scalars and the gradient collectives run live inside the list."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.rec = None

    def __call__(self, samples, targets):
        opt = self.optimizer
        opt._check_store()
        self.calls += 1
        if self.rec is None:
            if self.calls <= self.warmup:
                return train_one_step(self.model, self.criterion, samples, targets, opt, **self.kw)
            self.static_in = (samples, targets) if self.adopt_inputs else (samples.clone(), targets.clone())
            _lib.start_recording()
            try:
                self.out = train_one_step(self.model, self.criterion, self.static_in[0], self.static_in[1], opt, **self.kw)
            except BaseException:
                # a step that raised half-way (out of memory, a kernel error) leaves a truncated list: never keep it
                _lib.stop_recording()
                self.static_in, self.out, self.calls = None, None, 0
                raise
            self.rec = _lib.stop_recording()
            return self.out
        if samples.shape != self.static_in[0].shape or targets.shape != self.static_in[1].shape or samples.dtype != self.static_in[0].dtype:
            # a batch the list was not recorded for (a short last batch): the eager step, same kernels and streams, nothing recorded
            return train_one_step(self.model, self.criterion, samples, targets, opt, **self.kw)
        if samples.data_ptr() != self.static_in[0].data_ptr():
            self.static_in[0].copy_(samples, non_blocking=True)
        if targets.data_ptr() != self.static_in[1].data_ptr():
            self.static_in[1].copy_(targets, non_blocking=True)
        self._refresh_shadow()
        _lib.replay(self.rec)
        return self.out


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, args=None, grad_sync=None, log_every=30, logger=print,
                    max_norm=None, step_fn=None):
    """Epoch loop with the reference's signature shape (engine_proto.py:24-113); data_loader yields (samples, targets).
    step_fn: a ReplayedTrainStep (recorded command list) or GraphedTrainStep (captured HIP graph) instead of the eager train_one_step."""
    model.train(True)
    use_ppc = True if args is None else bool(getattr(args, "use_ppc_loss", True))
    cov_coe = 0.1 if args is None else getattr(args, "ppc_cov_coe", 0.1)
    mean_coe = 0.5 if args is None else getattr(args, "ppc_mean_coe", 0.5)
    if max_norm is None and args is not None:
        max_norm = getattr(args, "clip_grad", None)
    if step_fn is not None:
        step_fn.check_matches(epoch=epoch, ppc_cov_coe=cov_coe, ppc_mean_coe=mean_coe, use_ppc_loss=use_ppc, max_norm=max_norm)
    total, n = 0.0, 0
    for it, (samples, targets) in enumerate(data_loader):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if step_fn is not None:
            loss, _, _ = step_fn(samples, targets)
        else:
            loss, _, _ = train_one_step(model, criterion, samples, targets, optimizer, epoch, cov_coe, mean_coe, use_ppc, grad_sync,
                                        max_norm=max_norm)
        if it % log_every == 0:
            v = float(loss)
            # engine_proto.py:66-70 checks every step; here every step's loss is checked ON THE DEVICE by the optimizer kernel (a non-finite
            # loss skips that update and raises optimizer.nonfinite) and the host reads the flag at the logging cadence: same outcome
            # (training stops, no update was applied from a bad loss) without a host synchronisation per step
            if not math.isfinite(v) or int(optimizer.nonfinite) != 0:
                logger("Loss is {}, stopping training".format(v if not math.isfinite(v) else "non-finite in an earlier step"))
                sys.exit(1)
            logger(f"Epoch: [{epoch}] it {it} loss {v:.4f} lr {optimizer.param_groups[0]['lr']:.6f}")
            total += v
            n += 1
    # a non-finite loss between the last log read and the end of the epoch must not reach the epoch-end checkpoint / evaluation
    if int(optimizer.nonfinite) != 0:
        logger("Loss is non-finite in an earlier step, stopping training")
        sys.exit(1)
    return {"loss": total / max(n, 1), "lr": optimizer.param_groups[0]["lr"]}


@torch.no_grad()
def evaluate(data_loader, model, device):
    """engine_proto.py:143-184 reduced to the metrics (acc1 / global / local)."""
    model.eval()
    crit = CrossEntropyLoss()
    correct = correct_g = correct_l = count = 0
    loss_sum = 0.0
    for images, target in data_loader:
        images, target = images.to(device, non_blocking=True), target.to(device, non_blocking=True)
        output, aux = model(images)
        loss_sum += float(crit(output, target)) * images.shape[0]
        correct += int((output.argmax(1) == target).sum())
        correct_g += int((aux[2].argmax(1) == target).sum())
        correct_l += int((aux[3].argmax(1) == target).sum())
        count += images.shape[0]
    return dict(acc1=100.0 * correct / count, global_acc1=100.0 * correct_g / count, local_acc1=100.0 * correct_l / count, loss=loss_sum / count)


# ---------------------------------------------------------------------------------------------------- checkpoint I/O
def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, args=None, master_only=True):
    """The reference's checkpoint dict (main.py:436-447, 460-471; tools/utils.py:242-244 save_on_master):
    {'model', 'optimizer', 'lr_scheduler', 'epoch', 'model_ema', 'args'} -- 'scaler' is omitted (no fp16 loss scaling here)."""
    flag = getattr(optimizer, "nonfinite", None)
    if flag is not None and int(flag) != 0:
        raise RuntimeError("save_checkpoint: the optimizer's non-finite-loss flag is raised (a step was skipped because its loss was NaN / "
                           "inf, engine_proto.py:66-70 would have stopped there); refusing to write a checkpoint from this state")
    if master_only and dist.is_initialized() and dist.get_rank() != 0:
        return
    ck = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
          "optimizer": _to_cpu(optimizer.state_dict()),
          "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None,
          "epoch": int(epoch),
          "model_ema": {k: v.cpu() for k, v in optimizer.ema_state_dict().items()} if optimizer.ema is not None else None,
          "args": args}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ck, path)


def _to_cpu(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _to_cpu(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cpu(v) for v in x)
    return x


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, strict=True, eval_only=False):
    """main.py:393-407 (--resume) and main_visualize.py:289 (strict=False).  Returns the epoch to start from."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model"] if "model" in ck else ck, strict=strict)
    start = 0
    if not eval_only and optimizer is not None and all(k in ck for k in ("optimizer", "lr_scheduler", "epoch")):
        optimizer.load_state_dict(ck["optimizer"])
        if lr_scheduler is not None and ck["lr_scheduler"] is not None:
            lr_scheduler.load_state_dict(ck["lr_scheduler"])
        start = ck["epoch"] + 1
        if ck.get("model_ema") is not None:
            optimizer.load_ema_state_dict(ck["model_ema"])
    return start
