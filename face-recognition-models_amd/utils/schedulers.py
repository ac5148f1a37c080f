"""Learning-rate schedule used by the pipeline (reference: main_code/utils/schedulers.py:3-14,16-104).
Only "customstep" is ever selected upstream (model_utils.py:558); the torch built-ins stay reachable
through the same `get_scheduler(optimizer, choice)` entry for compatibility."""
import torch

CUSTOM_STEPS = (20, 40, 60)
CUSTOM_RATIO = 0.1


class CustomStepLR(torch.optim.lr_scheduler.LRScheduler):
    """Multiply the CURRENT lr by `ratio` whenever the epoch counter lands on one of `steps`."""

    def __init__(self, optimizer, steps, ratio=0.1, last_epoch=-1):
        self.steps = frozenset(steps)
        self.ratio = ratio
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        k = self.ratio if self.last_epoch in self.steps else 1.0
        return [g["lr"] * k for g in self.optimizer.param_groups]

    def state_dict(self):
        sd = super().state_dict()
        sd["steps"] = sorted(self.steps)      # a frozenset is not weights_only-loadable
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        sd["steps"] = frozenset(sd.get("steps", self.steps))
        super().load_state_dict(sd)


def get_scheduler(optimizer, choice="customstep", num_epochs=None, steps_per_epoch=None, **overrides):
    name = {1: "step", 2: "multistep", 3: "customstep", 4: "cosine", 5: "none"}.get(choice, choice)
    name = str(name).lower()
    sched = torch.optim.lr_scheduler
    if name == "customstep":
        return CustomStepLR(optimizer, **{"steps": CUSTOM_STEPS, "ratio": CUSTOM_RATIO, **overrides})
    if name == "step":
        return sched.StepLR(optimizer, **{"step_size": 30, "gamma": 0.1, **overrides})
    if name == "multistep":
        return sched.MultiStepLR(optimizer, **{"milestones": [40, 80, 100, 150], "gamma": 0.1, **overrides})
    if name == "cosine":
        return sched.CosineAnnealingLR(optimizer, T_max=num_epochs or 1, **{"eta_min": 0, **overrides})
    if name == "none":
        return sched.LambdaLR(optimizer, lambda _: 1.0)
    raise ValueError(f"Invalid scheduler: {choice}")
