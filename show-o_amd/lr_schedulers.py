"""Learning-rate schedules of the training loop -- drop-in for the reference's `models/lr_schedulers.py`
(`get_scheduler` and the six `get_*_schedule*` builders, `SchedulerType`; used at training/train.py:246-251, 618).

Same names, arguments and the same learning-rate sequences (tests/golden/lr_schedules.npz was recorded from the reference).
One schedule object serves both optimizer kinds: a `torch.optim.Optimizer` (its param_groups' "lr" are updated, like
`LambdaLR`) or the native `showo_amd.Trainer` (`set_lr`), whose AdamW step runs in one HIP kernel per bucket.
"""
import math
from enum import Enum


class SchedulerType(Enum):
    LINEAR = "linear"
    COSINE = "cosine"
    COSINE_WITH_RESTARTS = "cosine_with_restarts"
    POLYNOMIAL = "polynomial"
    CONSTANT = "constant"
    CONSTANT_WITH_WARMUP = "constant_with_warmup"


def _warm(step, warmup):
    return float(step) / float(max(1, warmup))


def _factor(kind, warmup=0, total=0, num_cycles=0.5, power=1.0, lr_init=1.0, lr_end=1e-7):
    """multiplier of the initial learning rate at optimizer step `step`"""
    if kind == "constant":
        return lambda step: 1
    if kind == "constant_with_warmup":
        return lambda step: float(step) / float(max(1.0, warmup)) if step < warmup else 1.0
    if kind == "linear":
        return lambda step: _warm(step, warmup) if step < warmup else max(0.0, float(total - step) / float(max(1, total - warmup)))
    if kind == "cosine":
        def f(step):
            if step < warmup:
                return _warm(step, warmup)
            progress = float(step - warmup) / float(max(1, total - warmup))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
        return f
    if kind == "cosine_with_restarts":
        def f(step):
            if step < warmup:
                return _warm(step, warmup)
            progress = float(step - warmup) / float(max(1, total - warmup))
            if progress >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * progress) % 1.0))))
        return f
    if kind == "polynomial":
        if not (lr_init > lr_end):
            raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

        def f(step):
            if step < warmup:
                return _warm(step, warmup)
            if step > total:
                return lr_end / lr_init
            remaining = 1 - (step - warmup) / (total - warmup)
            return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
        return f
    raise ValueError(kind)


class LRSchedule:
    """`torch.optim.lr_scheduler.LambdaLR` semantics (the constructor applies step 0, every `step()` advances by one) for a torch
    optimizer or the native Trainer"""

    def __init__(self, optimizer, factor, last_epoch=-1):
        self.optimizer, self.factor = optimizer, factor
        self._native = not hasattr(optimizer, "param_groups")
        self.base_lrs = [float(optimizer.lr)] if self._native else [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self._last_lr = list(self.base_lrs)
        self.step()

    def _apply(self):
        self._last_lr = [b * self.factor(self.last_epoch) for b in self.base_lrs]
        if self._native:
            self.optimizer.set_lr(self._last_lr[0])
        else:
            for g, lr in zip(self.optimizer.param_groups, self._last_lr):
                g["lr"] = lr

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs), "_last_lr": list(self._last_lr)}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = int(sd["last_epoch"]), list(sd["base_lrs"])
        self._apply()


def _lr_of(optimizer):
    return float(optimizer.lr) if not hasattr(optimizer, "param_groups") else float(optimizer.defaults["lr"])


def get_constant_schedule(optimizer, last_epoch=-1):
    return LRSchedule(optimizer, _factor("constant"), last_epoch)


def get_constant_schedule_with_warmup(optimizer, num_warmup_steps, last_epoch=-1):
    return LRSchedule(optimizer, _factor("constant_with_warmup", num_warmup_steps), last_epoch)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    return LRSchedule(optimizer, _factor("linear", num_warmup_steps, num_training_steps), last_epoch)


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5, last_epoch=-1):
    return LRSchedule(optimizer, _factor("cosine", num_warmup_steps, num_training_steps, num_cycles), last_epoch)


def get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=1, last_epoch=-1):
    return LRSchedule(optimizer, _factor("cosine_with_restarts", num_warmup_steps, num_training_steps, num_cycles), last_epoch)


def get_polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, lr_end=1e-7, power=1.0, last_epoch=-1):
    return LRSchedule(optimizer, _factor("polynomial", num_warmup_steps, num_training_steps, power=power, lr_init=_lr_of(optimizer),
                                         lr_end=lr_end), last_epoch)


TYPE_TO_SCHEDULER_FUNCTION = {
    SchedulerType.LINEAR: get_linear_schedule_with_warmup,
    SchedulerType.COSINE: get_cosine_schedule_with_warmup,
    SchedulerType.COSINE_WITH_RESTARTS: get_cosine_with_hard_restarts_schedule_with_warmup,
    SchedulerType.POLYNOMIAL: get_polynomial_decay_schedule_with_warmup,
    SchedulerType.CONSTANT: get_constant_schedule,
    SchedulerType.CONSTANT_WITH_WARMUP: get_constant_schedule_with_warmup,
}


def get_scheduler(name, optimizer, num_warmup_steps=None, num_training_steps=None, num_cycles=1, power=1.0):
    name = SchedulerType(name)
    build = TYPE_TO_SCHEDULER_FUNCTION[name]
    if name == SchedulerType.CONSTANT:
        return build(optimizer)
    if num_warmup_steps is None:
        raise ValueError(f"{name} requires `num_warmup_steps`, please provide that argument.")
    if name == SchedulerType.CONSTANT_WITH_WARMUP:
        return build(optimizer, num_warmup_steps=num_warmup_steps)
    if num_training_steps is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    if name == SchedulerType.COSINE_WITH_RESTARTS:
        return build(optimizer, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps, num_cycles=num_cycles)
    if name == SchedulerType.POLYNOMIAL:
        return build(optimizer, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps, power=power)
    return build(optimizer, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps)
