"""LR schedulers of the SR recipe: the reference's `get_schedulers` contract
(codes/models/schedulers.py:9-126) for the MultiStepLR branch (:72-74), which is stock PyTorch and
drives FusedAdam through `param_groups[...]['lr']` like any optimizer."""
from torch.optim import lr_scheduler


def get_schedulers(optimizers=None, schedulers=None, train_opt=None):
    schedulers = schedulers or []
    scheme = train_opt["lr_scheme"]
    for optimizer in optimizers:
        if scheme == "MultiStepLR":
            sched = lr_scheduler.MultiStepLR(optimizer, train_opt["lr_steps"], train_opt["lr_gamma"])
        elif scheme == "StepLR":
            sched = lr_scheduler.StepLR(optimizer, step_size=train_opt["lr_step_size"], gamma=train_opt["lr_gamma"])
        else:
            raise NotImplementedError("Learning rate scheme [{}] is outside the SR hot path of the HIP engine".format(scheme))
        schedulers.append(sched)
    return schedulers
