"""Flat parameter / gradient / bf16-shadow storage for a model (host-side plumbing).

All trainable parameters of a module live in ONE fp32 device buffer (8-element aligned segments), their
gradients in a second one and a bf16 shadow copy (the MFMA operand format) in a third:
  * one kernel launch re-casts every weight to bf16 (or the fused AdamW writes the shadow itself),
  * backward kernels accumulate straight into the flat gradient (param.grad are views of it),
  * the data-parallel gradient exchange is a single RCCL all-reduce over the flat buffer,
  * the optimizer is one fused kernel over (p, g, m, v[, ema, bf16]).
nn.Parameter objects stay ordinary views, so state_dict()/load_state_dict() keep the reference's key layout.
"""
import torch

from . import ops

ALIGN = 8


class FlatStore:
    def __init__(self, module, groups):
        """groups: list of (name, [(param_name, param), ...]) in buffer order (= optimizer segments)."""
        self.module = module
        self.entries = []          # (name, param, offset, numel)
        self.segments = []         # (group name, begin, end)
        off = 0
        for gname, plist in groups:
            begin = off
            for name, p in plist:
                if not p.requires_grad:
                    continue
                self.entries.append((name, p, off, p.numel()))
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.segments.append((gname, begin, off))
        self.total = off
        dev = self.entries[0][1].device
        self.device = dev
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.bf16 = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self._views16 = {}
        self._gviews = {}
        with torch.no_grad():
            for name, p, o, n in self.entries:
                self.params[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.params[o:o + n].view(p.shape)
                self._views16[id(p)] = self.bf16[o:o + n].view(p.shape)
                self._gviews[id(p)] = self.grads[o:o + n].view(p.shape)
                p.grad = self._gviews[id(p)]
        self.bf16_fresh = False
        self._t16 = None           # transposed bf16 shadows (register_transposed)

    # ------------------------------------------------------------------ transposed bf16 shadows
    def register_transposed(self, params):
        """Keep W^T (bf16, [in][out]) next to the bf16 shadow of every 2-D weight in `params`: the input-gradient products
        dx = dy W then read both operands contraction-contiguous (csrc/rowgemm.hip).  One transposing launch per refresh."""
        params = [p for p in params if id(p) in self._views16]
        if not params:
            return
        off, desc, tiles, views = 0, [], 0, {}
        by_id = {id(p): o for _, p, o, _ in self.entries}
        for p in params:
            rows, cols = p.shape[0], p.numel() // p.shape[0]
            desc.append((by_id[id(p)], off, rows, cols))
            tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
            views[id(p)] = (off, rows, cols)
            off += (rows * cols + ALIGN - 1) // ALIGN * ALIGN
        buf = torch.empty(off, dtype=torch.bfloat16, device=self.device)
        self._t16 = dict(buf=buf, desc=torch.tensor(desc, dtype=torch.int64).to(self.device), n=len(desc), tiles=tiles, fresh=False,
                         views={k: buf[o:o + r * c].view(c, r) for k, (o, r, c) in views.items()})

    def refresh_t16(self):
        t = self._t16
        if t is not None and not t["fresh"]:
            ops.transpose_bf16_batched(self.bf16, t["buf"], t["desc"], t["n"], t["tiles"])
            t["fresh"] = True

    def w16t(self, p):
        """bf16 W^T ([in][out]) of a registered weight, or None."""
        return None if self._t16 is None else self._t16["views"].get(id(p))

    # ------------------------------------------------------------------ bf16 shadow
    def refresh_bf16(self, force=False):
        """Re-cast the fp32 masters to the bf16 shadow (one streaming kernel)."""
        if force or not self.bf16_fresh:
            ops.cast_bf16(self.params, self.bf16)
            self.bf16_fresh = True
            if self._t16 is not None:
                self._t16["fresh"] = False

    def w16(self, p):
        return self._views16[id(p)]

    def mark_bf16_written(self):
        """The fused optimizer kernel has just rewritten the bf16 shadow from the new fp32 masters."""
        self.bf16_fresh = True
        if self._t16 is not None:
            self._t16["fresh"] = False

    def invalidate(self):
        self.bf16_fresh = False
        if self._t16 is not None:
            self._t16["fresh"] = False

    # ------------------------------------------------------------------ gradients
    def grad_view(self, p):
        """The flat-gradient view of p, attached as p.grad (zeroed first if p.grad had been dropped)."""
        g = self._gviews[id(p)]
        if p.grad is None or p.grad.data_ptr() != g.data_ptr():
            ops.zero_(g)
            p.grad = g
        return g

    def attach_all_grads(self):
        """Called at the start of a backward pass: if every grad was set to None (zero_grad default), one memset."""
        if all(p.grad is None for _, p, _, _ in self.entries):
            ops.zero_(self.grads)
            for _, p, _, _ in self.entries:
                p.grad = self._gviews[id(p)]
        else:
            for _, p, _, _ in self.entries:
                self.grad_view(p)

    def zero_grad(self):
        ops.zero_(self.grads)
        for _, p, _, _ in self.entries:
            p.grad = self._gviews[id(p)]

    def still_flat(self):
        """False if someone re-materialised the parameters (e.g. module.to(), load with assign)."""
        n, p, o, _ = self.entries[0]
        return p.data_ptr() == self.params.data_ptr() + 4 * o and p.device == self.device
