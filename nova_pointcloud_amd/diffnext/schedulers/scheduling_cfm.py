"""Flow-matching Euler sampler (reference diffnext/schedulers/scheduling_cfm.py:35-140).

Sampling grid: S timesteps evenly spaced (float32) between sigma_max*T and sigma_min*T, sigmas
shifted by `shift * s / (1 + (shift - 1) * s)`, a final sigma of 0, and the update
x <- x + (sigma_{i+1} - sigma_i) * v. On an MI355X the update is fused with the head projection
and CFG in `nova_head_cfg_euler`; the engine only reads `timesteps` / `sigmas` from this object.
"""
import math

import numpy as np
import torch

from .._compat import BaseOutput, ConfigMixin, SchedulerMixin, register_to_config


class FlowMatchEulerDiscreteSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor


def _shifted(sigmas, shift):
    return shift * sigmas / (1 + (shift - 1) * sigmas)


class FlowMatchEulerDiscreteScheduler(SchedulerMixin, ConfigMixin):
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False):
        sigmas = np.arange(1, num_train_timesteps + 1, dtype="float32")[::-1] / num_train_timesteps
        self._shift = shift
        if not use_dynamic_shifting:
            sigmas = _shifted(sigmas, shift)
        self.timesteps = torch.as_tensor(sigmas * num_train_timesteps)
        self.sigmas = torch.as_tensor(sigmas.copy())
        self.sigma_min, self.sigma_max = float(sigmas[-1]), float(sigmas[0])
        self.timestep = self.sigma = None  # training state
        self._begin_index = self._step_index = None  # inference counters

    @property
    def shift(self):
        return self._shift

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_shift(self, shift: float):
        self._shift = shift

    def time_shift(self, mu: float, sigma: float, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        grid = self.timesteps if schedule_timesteps is None else schedule_timesteps
        hits = (torch.as_tensor(grid) == timestep).nonzero()
        return hits[1 if len(hits) > 1 else 0].item()

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self.begin_index is None else self._begin_index

    def sample_timesteps(self, size, device=None):
        """Logit-normal training timesteps."""
        u = torch.normal(0, 1, size, device=device).sigmoid_()
        return u.mul_(self.config.num_train_timesteps).to(dtype=torch.int64)

    def set_timesteps(self, num_inference_steps, mu=None):
        n = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        grid = np.linspace(self.sigma_max * n, self.sigma_min * n, num_inference_steps, dtype="float32")
        sigmas = grid / n
        sigmas = self.time_shift(mu, 1.0, sigmas) if self.config.use_dynamic_shifting else _shifted(sigmas, self.shift)
        self.sigmas = sigmas.tolist() + [0]
        self.timesteps = sigmas * n
        self._begin_index = self._step_index = None

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = sigma * noise + (1 - sigma) * x_0 at the sampled training timesteps."""
        dtype, device = original_samples.dtype, original_samples.device
        self.timestep = self.timesteps.to(device=device)[timesteps]
        sigma = self.sigmas.to(device=device, dtype=dtype)[timesteps]
        self.sigma = sigma.view(timesteps.shape + (1,) * (noise.dim() - timesteps.dim()))
        return self.sigma * noise + (1.0 - self.sigma) * original_samples

    def scale_noise(self, sample, timestep, noise):
        if self.step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self.step_index]
        return sigma * noise + (1.0 - sigma) * sample

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        if self.step_index is None:
            self._init_step_index(timestep)
        dt = self.sigmas[self.step_index + 1] - self.sigmas[self.step_index]
        prev_sample = model_output * dt + sample
        self._step_index += 1
        return FlowMatchEulerDiscreteSchedulerOutput(prev_sample=prev_sample) if return_dict else (prev_sample,)
