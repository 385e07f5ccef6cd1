"""Optimiser factory of the SR path.

`config_optimizer(train_opt, name, net)` keeps the reference's contract (codes/models/optimizers.py:
74-134: lr_<G|D> default 1e-4, beta1 0.9, beta2 0.999, weight decay 0, Adam default eps 1e-8) but the
'adam' branch (:130-132) returns FusedAdam: torch.optim.Adam's update rule as ONE HIP launch over the
network's flat parameter buffer instead of ~700 per-tensor kernels.  state_dict()/load_state_dict()
use torch.optim.Adam's layout ('step', 'exp_avg', 'exp_avg_sq' per parameter) so `.state` files
written by either engine resume in the other (base_model.py:454-500).
"""
import logging
import math
import os

import torch

from .. import ops

logger = logging.getLogger("base")


def get_optim_params(networks, only_requires_grad=True, param_filter=None):
    if isinstance(networks, torch.nn.Module):
        networks = [networks]
    if param_filter:
        raise NotImplementedError("parameter filters are not implemented by the HIP engine")
    params = []
    for net in networks:
        params += [p for p in net.parameters() if p.requires_grad or not only_requires_grad]
    return params, []


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._moments = {}     # id(FlatParams) -> (exp_avg_flat, exp_avg_sq_flat)
        self._t = {}           # group index -> optimiser steps taken (ONE host counter per group; the per-parameter `step` tensors of
                               # torch.optim.Adam's layout exist only in state_dict(), where the reference resuming from it needs them)

    def _holders(self, group):
        """The distinct flat buffers the group's parameters live in (insertion ordered)."""
        holders = {}
        for p in group["params"]:
            h = getattr(p, "_tnr_flat", None)
            if h is None:
                raise RuntimeError("FusedAdam: parameter is not part of a HIP-engine network (no flat storage)")
            holders.setdefault(id(h[0]), h[0])
        return list(holders.values())

    def _ensure_state(self, group):
        for holder in self._holders(group):
            holder.ensure()
            key = id(holder)
            mv = self._moments.get(key)
            if mv is None or mv[0].numel() != holder.total or mv[0].device != holder.flat.device:
                mv = (torch.zeros_like(holder.flat), torch.zeros_like(holder.flat))
                self._moments[key] = mv
                for p in group["params"]:
                    if p._tnr_flat[0] is not holder:
                        continue
                    st = self.state[p]
                    o, n = p._tnr_flat[1], p.numel()
                    old_m, old_v = st.get("exp_avg"), st.get("exp_avg_sq")
                    st["exp_avg"] = mv[0][o:o + n].view(p.shape)
                    st["exp_avg_sq"] = mv[1][o:o + n].view(p.shape)
                    if old_m is not None:          # state arrived through load_state_dict
                        st["exp_avg"].copy_(old_m)
                        st["exp_avg_sq"].copy_(old_v)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        for gi, group in enumerate(self.param_groups):
            self._ensure_state(group)
            b1, b2 = group["betas"]
            t = self._t.get(gi, 0) + 1
            bc1 = 1.0 - b1 ** t
            bc2 = 1.0 - b2 ** t
            for holder in self._holders(group):
                m, v = self._moments[id(holder)]
                ops.adam_step(holder.flat, holder.grad, m, v, group["lr"] / bc1, b1, b2, math.sqrt(bc2), group["eps"],
                              group["weight_decay"])
                holder.touch()                       # packed weights of this network are stale now
            self._t[gi] = t

    def group_steps(self):
        """Optimiser steps taken, per parameter group."""
        return [self._t.get(gi, 0) for gi in range(len(self.param_groups))]

    def state_dict(self):
        """torch.optim.Adam's layout.  `step` is materialised here, one tensor PER parameter: torch.optim.Adam (the reference resuming
        from this state, base_model.py:479-491) increments each entry in place and torch.save keeps aliasing -- a shared tensor would be
        bumped once per parameter per step."""
        for gi, group in enumerate(self.param_groups):
            if self._moments or gi in self._t:
                self._ensure_state(group)
            for p in group["params"]:
                if p in self.state and "exp_avg" in self.state[p]:
                    self.state[p]["step"] = torch.tensor(float(self._t.get(gi, 0)))
        return super().state_dict()

    def zero_grad(self, set_to_none=False):
        """Zero the flat gradient buffers in place (views stay attached to the parameters)."""
        for group in self.param_groups:
            for holder in self._holders(group):
                holder.ensure()
                ops.fill(holder.grad, 0.0)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._moments = {}       # re-home the loaded moments into flat buffers on next use
        self._t = {}
        for gi, group in enumerate(self.param_groups):
            steps = {int(float(self.state[p]["step"])) for p in group["params"] if p in self.state and "step" in self.state[p]}
            if len(steps) > 1:
                # a torch.optim.Adam checkpoint may legitimately hold this: a parameter whose grad was None for some steps (frozen
                # or unfrozen mid-run) lags the others.  FusedAdam keeps ONE counter per group (its bias correction is a launch
                # argument): resume from the most advanced one -- what the bulk of the parameters had -- and say so.
                # TNR_STRICT_OPTIM_STATE=1 turns this into an error.
                msg = "FusedAdam: the parameters of group %d carry different step counts %s; resuming with step %d for all of them" % (
                    gi, sorted(steps), max(steps))
                if os.environ.get("TNR_STRICT_OPTIM_STATE") == "1":
                    raise ValueError(msg)
                logger.warning(msg)
            if steps:
                self._t[gi] = max(steps)
            self._ensure_state(group)


def config_optimizer(train_opt, name, net=None, optim_params=None):
    if name not in ("G", "D"):
        raise NotImplementedError("Invalid optimizer name: {}".format(name))
    for n in ([net] if isinstance(net, torch.nn.Module) else (net or [])):
        if hasattr(n, "flat_params"):
            n.flat_params()              # parameters become views of the network's flat buffer
    if not optim_params:
        optim_params, _ = get_optim_params(net, True)
    def opt_or(key, default):
        # missing keys read as None (NoneDict); legitimate falsy values (beta1_G: 0, lr_D: 0) are kept
        v = train_opt.get(key, None)
        return default if v is None else v

    optim = opt_or("optim_" + name, "adam")
    lr = opt_or("lr_" + name, 1e-4)
    wd = opt_or("weight_decay_" + name, 0)
    beta1 = opt_or("beta1_" + name, 0.9)
    beta2 = opt_or("beta2_" + name, 0.999)
    if optim != "adam":
        raise NotImplementedError("optimizer [{}] is outside the SR hot path of the HIP engine".format(optim))
    return FusedAdam(optim_params, lr=lr, weight_decay=wd, betas=(beta1, beta2))
