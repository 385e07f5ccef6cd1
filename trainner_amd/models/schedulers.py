"""LR schedulers of the training recipes: the reference's `get_schedulers` contract (codes/models/schedulers.py:9-126)
for MultiStepLR / StepLR (:72-74,84-87; stock PyTorch) and the image-to-image recipes' `Linear` policy (:39-52: the
initial rate for `fixed_niter` iterations, then a linear decay to zero over `niter_decay`; `fixed_niter_rel` splits
`niter` instead).  They drive FusedAdam through `param_groups[...]['lr']` like any optimizer."""
from torch.optim import lr_scheduler


def _linear_rule(train_opt):
    def rule(it):
        rel = train_opt.get("fixed_niter_rel", None)
        if rel:
            assert 0 < rel <= 1.0
            fixed = train_opt["niter"] * rel
            decay = train_opt["niter"] - fixed
        else:
            fixed, decay = train_opt["fixed_niter"], train_opt["niter_decay"]
        return max(0, 1.0 - max(0, it + 1 - fixed) / max(1, decay))
    return rule


def get_schedulers(optimizers=None, schedulers=None, train_opt=None):
    schedulers = schedulers or []
    scheme = train_opt["lr_scheme"]
    for optimizer in optimizers:
        if scheme == "MultiStepLR":
            sched = lr_scheduler.MultiStepLR(optimizer, train_opt["lr_steps"], train_opt["lr_gamma"])
        elif scheme == "StepLR":
            sched = lr_scheduler.StepLR(optimizer, step_size=train_opt["lr_step_size"], gamma=train_opt["lr_gamma"])
        elif scheme == "Linear":
            sched = lr_scheduler.LambdaLR(optimizer, lr_lambda=_linear_rule(train_opt))
        else:
            raise NotImplementedError("Learning rate scheme [{}] is outside the hot path of the HIP engine".format(scheme))
        schedulers.append(sched)
    return schedulers
