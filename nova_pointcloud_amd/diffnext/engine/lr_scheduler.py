"""Learning-rate schedules with the reference's interface (diffnext/engine/lr_scheduler.py:20-83): `get_lr()` for the
current step, `step()` to advance; `_step_count` is the trainer's global step."""
import math


class ConstantLR(object):
    """lr_max after a linear warm-up from lr_max * warmup_factor; subclasses scale (lr_max - lr_min) by `get_decay()`."""

    def __init__(self, lr_max, lr_min=0, warmup_steps=0, warmup_factor=0.001, **unused):
        self._lr_max, self._lr_min = lr_max, lr_min
        self._warmup_steps, self._warmup_factor = warmup_steps, warmup_factor
        self._step_count, self._last_decay = 0, 1.0

    def step(self):
        self._step_count += 1

    def get_decay(self):
        return self._last_decay

    def get_lr(self):
        if self._step_count < self._warmup_steps:
            ramp = (self._step_count + 1.0) / self._warmup_steps
            return self._lr_max * (ramp + (1.0 - ramp) * self._warmup_factor)
        return self._lr_min + (self._lr_max - self._lr_min) * self.get_decay()


class CosineLR(ConstantLR):
    """Half-cosine from lr_max to lr_min over max_steps, re-evaluated every `decay_step` steps."""

    def __init__(self, lr_max, max_steps, lr_min=0, decay_step=1, **kwargs):
        super().__init__(lr_max=lr_max, lr_min=lr_min, **kwargs)
        self._max_steps, self._decay_step = max_steps, decay_step

    def get_decay(self):
        done = self._step_count - self._warmup_steps
        if done > 0 and done % self._decay_step == 0:
            self._last_decay = 0.5 * (1.0 + math.cos(math.pi * done / (self._max_steps - self._warmup_steps)))
        return self._last_decay


class MultiStepLR(ConstantLR):
    """lr_max * decay_gamma^k after the k-th milestone of `decay_steps`."""

    def __init__(self, lr_max, decay_steps, decay_gamma, **kwargs):
        super().__init__(lr_max=lr_max, **kwargs)
        self._decay_steps, self._decay_gamma, self._stage = list(decay_steps), decay_gamma, 0

    def get_decay(self):
        while self._stage < len(self._decay_steps) and self._step_count >= self._decay_steps[self._stage]:
            self._stage += 1
        if self._decay_steps:
            self._last_decay = self._decay_gamma ** self._stage
        return self._last_decay
