"""Flat parameter / gradient storage for a network.

Every parameter of a network is a view into ONE contiguous fp32 buffer and its `.grad` a view into
a second one.  That is what lets the engine
  * run clip_grad_norm_ + Adam as two launches over the whole network (base_model.py:815-850,
    911-922; optimizers.py:130-132) instead of ~700 per-tensor kernels, and
  * hand contiguous gradient buckets to RCCL while the backward pass is still running
    (trainner_amd/dp.py).
state_dict() keys/shapes are untouched (views keep their own shape), so .pth checkpoints
round-trip with the reference (base_model.py:353-443).
"""
import torch

ALIGN = 64  # floats; keeps every parameter 256-B aligned


class FlatParams:
    def __init__(self, module):
        self.module = module
        self.flat = None
        self.grad = None
        self.offsets = []   # (param, offset, numel)
        self.total = 0
        self.version = 0    # bumped whenever parameter VALUES may have changed (optimizer step, load, rebuild)

    def touch(self):
        self.version += 1

    def params(self):
        return [p for p in self.module.parameters()]

    def _consistent(self):
        if self.flat is None:
            return False
        base = self.flat.data_ptr()
        gbase = self.grad.data_ptr()
        ps = self.params()
        if len(ps) != len(self.offsets):
            return False
        for p, (q, off, n) in zip(ps, self.offsets):
            if p is not q or p.data_ptr() != base + 4 * off or p.device != self.flat.device:
                return False
            if p.grad is None or p.grad.data_ptr() != gbase + 4 * off:
                return False
        return True

    def ensure(self):
        """(Re)build the flat buffers if a parameter was moved / replaced (e.g. by module.to())."""
        if self._consistent():
            return self
        ps = self.params()
        if not ps:
            raise RuntimeError("network has no parameters")
        dev = ps[0].device
        self.offsets, off = [], 0
        for p in ps:
            if p.dtype != torch.float32:
                raise RuntimeError("the HIP engine keeps fp32 master parameters")
            self.offsets.append((p, off, p.numel()))
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o, n in self.offsets:
                flat[o:o + n].copy_(p.data.reshape(-1))
                old_grad = p.grad
                p.data = flat[o:o + n].view(p.shape)
                g = grad[o:o + n].view(p.shape)
                if old_grad is not None:
                    g.copy_(old_grad.to(dev))
                p.grad = g
                p._tnr_flat = (self, o)
        self.flat, self.grad = flat, grad
        self.version += 1
        return self

    def view_of(self, p, which="grad"):
        holder, o = p._tnr_flat
        assert holder is self
        src = self.grad if which == "grad" else self.flat
        return src[o:o + p.numel()].view(p.shape)
