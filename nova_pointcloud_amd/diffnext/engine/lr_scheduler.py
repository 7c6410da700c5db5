"""Learning-rate schedules behind the reference's interface (diffnext/engine/lr_scheduler.py:20-83): `get_lr()` gives the
rate of the current step, `step()` advances, `_step_count` is the trainer's global step (the Trainer reads and, on resume,
sets it).

One stateless rule per schedule: the rate is a pure function of the step count, so resuming at step k or asking twice gives
what the reference's running state would hold there:
  warm-up  (step < warmup_steps)  lr_max * (r + (1 - r) * warmup_factor),  r = (step + 1) / warmup_steps
  after it                        lr_min + (lr_max - lr_min) * decay(step)
"""
import bisect
import math


class _Schedule(object):
    def __init__(self, lr_max, lr_min=0, warmup_steps=0, warmup_factor=0.001):
        self._lr_max, self._lr_min = lr_max, lr_min
        self._warmup_steps, self._warmup_factor = warmup_steps, warmup_factor
        self._step_count = 0

    def step(self):
        self._step_count += 1

    def decay(self, step) -> float:
        return 1.0

    def get_decay(self):
        return self.decay(self._step_count)

    def get_lr(self):
        k = self._step_count
        if k < self._warmup_steps:
            r = (k + 1.0) / self._warmup_steps
            return self._lr_max * (r + (1.0 - r) * self._warmup_factor)
        return self._lr_min + (self._lr_max - self._lr_min) * self.decay(k)


class ConstantLR(_Schedule):
    """lr_max after the warm-up."""

    def __init__(self, **kwargs):
        super().__init__(kwargs.pop("lr_max"), kwargs.pop("lr_min", 0), kwargs.pop("warmup_steps", 0), kwargs.pop("warmup_factor", 0.001))


class CosineLR(_Schedule):
    """Half cosine from lr_max to lr_min over max_steps, held piecewise constant over windows of `decay_step` steps."""

    def __init__(self, lr_max, max_steps, lr_min=0, decay_step=1, **kwargs):
        super().__init__(lr_max, lr_min, kwargs.pop("warmup_steps", 0), kwargs.pop("warmup_factor", 0.001))
        self._max_steps, self._decay_step = max_steps, decay_step

    def decay(self, step):
        t = step - self._warmup_steps
        t -= t % self._decay_step  # the reference refreshes its value only on multiples of decay_step
        if t <= 0:
            return 1.0
        return 0.5 * (1.0 + math.cos(math.pi * t / (self._max_steps - self._warmup_steps)))


class MultiStepLR(_Schedule):
    """lr_max * decay_gamma^k once k of the `decay_steps` milestones have been reached."""

    def __init__(self, lr_max, decay_steps, decay_gamma, **kwargs):
        super().__init__(lr_max, kwargs.pop("lr_min", 0), kwargs.pop("warmup_steps", 0), kwargs.pop("warmup_factor", 0.001))
        self._milestones, self._gamma = sorted(decay_steps), decay_gamma

    def decay(self, step):
        return self._gamma ** bisect.bisect_right(self._milestones, step)
