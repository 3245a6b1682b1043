"""Train-step loop of the hot path (tools/engine_proto.py:41-81; optimizer groups tools/create_optimizer.py:27-39).

* FlatAdamW  : torch.optim.AdamW semantics as ONE fused HIP kernel over the flat (p, g, m, v[, ema]) buffers that also
               re-emits the bf16 weight shadow; param_groups expose lr / weight_decay so schedulers work unchanged.
* GradSync   : data-parallel gradient exchange = chunked RCCL all-reduce of the flat gradient buffer on a side
               stream, launched as soon as a chunk's layers have finished their backward (overlap), averaged by
               folding 1/world into the optimizer kernel.
* train_one_step / train_one_epoch : the reference's step body (forward, CE + PPC, backward, step, EMA).
"""
import math
import os
import sys

import torch
import torch.distributed as dist

from . import _lib
from .protopformer import CrossEntropyLoss

DEFAULT_LRS = {"features": 1e-4, "add_on_layers": 3e-3, "prototype_vectors": 3e-3}       # main.py:64-66


class FlatAdamW:
    def __init__(self, ppnet, joint_optimizer_lrs=None, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, ema_decay=None):
        lrs = dict(DEFAULT_LRS if joint_optimizer_lrs is None else joint_optimizer_lrs)
        self.ppnet = ppnet
        self.store = ppnet.flat_store()
        st = self.store
        # the reference's four groups, in flat-buffer order (create_optimizer.py:31-39)
        wds = {"features": 1e-3, "add_on_layers": 1e-3, "prototype_vectors": weight_decay, "prototype_vectors_global": weight_decay}
        lr_of = {"features": lrs["features"], "add_on_layers": lrs["add_on_layers"], "prototype_vectors": lrs["prototype_vectors"],
                 "prototype_vectors_global": lrs["prototype_vectors"]}
        self.param_groups = [dict(name=n, lr=lr_of[n], weight_decay=wds[n], begin=b, end=e, initial_lr=lr_of[n]) for n, b, e in st.segments]
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(st.params)
        self.exp_avg_sq = torch.zeros_like(st.params)
        self.ema = st.params.clone() if ema_decay is not None else None
        self.ema_decay = 0.0 if ema_decay is None else float(ema_decay)
        self.step_count = 0
        self.grad_scale = 1.0
        self._bounds = torch.tensor([g["begin"] for g in self.param_groups] + [self.param_groups[-1]["end"]], dtype=torch.int64)
        self._lr = torch.zeros(len(self.param_groups), dtype=torch.float32)
        self._wd = torch.zeros(len(self.param_groups), dtype=torch.float32)

    def zero_grad(self, set_to_none=False):
        self.store.zero_grad()

    def step(self):
        st = self.store
        self.step_count += 1
        for i, g in enumerate(self.param_groups):
            self._lr[i] = g["lr"]
            self._wd[i] = g["weight_decay"]
        _lib.call("ppf_adamw_step", st.params, st.grads, self.exp_avg, self.exp_avg_sq, self.ema, st.bf16, st.total, len(self.param_groups),
                  self._bounds.data_ptr(), self._lr.data_ptr(), self._wd.data_ptr(), self.betas[0], self.betas[1], self.eps,
                  self.step_count, self.ema_decay, float(self.grad_scale))
        st.bf16_fresh = True                      # the kernel re-emitted the bf16 shadow

    def state_dict(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, step=self.step_count, ema=self.ema,
                    param_groups=[{k: v for k, v in g.items()} for g in self.param_groups])

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_count = sd["step"]
        if self.ema is not None and sd.get("ema") is not None:
            self.ema.copy_(sd["ema"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(lr=s["lr"], weight_decay=s["weight_decay"])

    def ema_state_dict(self):
        """EMA weights under the model's state-dict keys (trainable tensors only)."""
        out = {}
        for name, p, o, n in self.store.entries:
            out[name] = self.ema[o:o + n].view(p.shape).clone()
        return out


class GradSync:
    """Chunked all-reduce (sum) of a flat gradient buffer over the default process group, overlapped with backward."""

    def __init__(self, flat_grads, chunk_bounds, use_side_stream=True):
        self.g = flat_grads
        self.bounds = list(chunk_bounds)                      # ascending element offsets; chunk c = [bounds[c], bounds[c+1])
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.force = os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0"     # single-rank runs still issue the collectives (tests)
        self.cuda = flat_grads.is_cuda
        self.stream = torch.cuda.Stream() if (self.cuda and use_side_stream) else None
        self.pending = []

    def chunk_ready(self, c, also=()):
        """Launch the all-reduce of chunk c: every kernel that writes it has been enqueued on the current stream or on one of
        the `also` streams (the weight-gradient lane) -- the communication stream waits for all of them, the compute streams
        never wait for each other here."""
        if self.world == 1 and not self.force:
            return
        lo, hi = self.bounds[c], self.bounds[c + 1]
        if hi <= lo:
            return
        view = self.g[lo:hi]
        if self.stream is not None:
            evs = []
            for st in (torch.cuda.current_stream(),) + tuple(s_ for s_ in also if s_ is not None):
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            with torch.cuda.stream(self.stream):
                for ev in evs:
                    self.stream.wait_event(ev)
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            for st in also:
                if st is not None:
                    torch.cuda.current_stream().wait_stream(st)
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Wait for all chunks; returns the scale (1/world) the optimizer must apply to the summed gradients."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return 1.0 / self.world


def make_grad_sync(ppnet, n_chunks=4):
    """Chunk the flat gradient in backward-completion order: [heads+norm | late blocks | ... | early blocks+embedding]."""
    st = ppnet.flat_store()
    block_offsets = []
    for name, p, o, n in st.entries:
        if name.startswith("blocks.") and name.split(".")[2] == "norm1" and name.endswith("weight"):
            block_offsets.append(o)
    depth = len(block_offsets)
    per = max(1, math.ceil(depth / max(1, n_chunks - 1)))
    cuts = sorted({block_offsets[i] for i in range(0, depth, per)})
    norm_off = next(o for name, p, o, n in st.entries if name == "norm.weight")
    bounds = [0] + [c for c in cuts if c > 0] + [norm_off, st.total]
    bounds = sorted(set(bounds))
    sync = GradSync(st.grads, bounds)
    # chunk index that becomes complete when block i's backward has been enqueued
    sync.block_chunk = {}
    for i, off in enumerate(block_offsets):
        if off in bounds and off != 0:
            sync.block_chunk[i] = bounds.index(off)
    sync.tail_chunk = len(bounds) - 2                 # final norm + add-on + prototypes: ready first
    sync.head_chunk = 0                               # embedding + first blocks: ready last
    return sync


def train_one_step(model, criterion, samples, targets, optimizer, epoch=20, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True,
                   grad_sync=None, check_finite=False):
    """One iteration of tools/engine_proto.py:41-81 (without logging). Returns the detached loss tensors."""
    outputs, auxi = model(samples)
    loss = criterion(outputs, targets)
    cov = mean = None
    if use_ppc_loss:
        cov, mean = model.get_PPC_loss(auxi[2], auxi[3], auxi[4], targets)
        if epoch >= 20:                                           # engine_proto.py:61-64
            loss = loss + ppc_cov_coe * cov + ppc_mean_coe * mean
    if check_finite and not math.isfinite(loss.item()):           # engine_proto.py:66-70 (host sync: opt-in)
        print("Loss is {}, stopping training".format(loss.item()))
        sys.exit(1)
    optimizer.zero_grad()
    model._grad_sync = grad_sync
    loss.backward()
    model._grad_sync = None
    if grad_sync is not None:
        optimizer.grad_scale = grad_sync.finish()
    optimizer.step()
    return loss.detach(), (cov.detach() if cov is not None else None), (mean.detach() if mean is not None else None)


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, args=None, grad_sync=None, log_every=30, logger=print):
    """Epoch loop with the reference's signature shape (engine_proto.py:24-113); data_loader yields (samples, targets)."""
    model.train(True)
    use_ppc = True if args is None else bool(getattr(args, "use_ppc_loss", True))
    cov_coe = 0.1 if args is None else getattr(args, "ppc_cov_coe", 0.1)
    mean_coe = 0.5 if args is None else getattr(args, "ppc_mean_coe", 0.5)
    total, n = 0.0, 0
    for it, (samples, targets) in enumerate(data_loader):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        loss, _, _ = train_one_step(model, criterion, samples, targets, optimizer, epoch, cov_coe, mean_coe, use_ppc, grad_sync)
        if it % log_every == 0:
            v = float(loss)
            if not math.isfinite(v):
                logger("Loss is {}, stopping training".format(v))
                sys.exit(1)
            logger(f"Epoch: [{epoch}] it {it} loss {v:.4f} lr {optimizer.param_groups[0]['lr']:.6f}")
            total += v
            n += 1
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
