"""Flow-matching Euler sampler: the surface of reference diffnext/schedulers/scheduling_cfm.py:35-140.

Sampling grid: S timesteps evenly spaced (float32) between sigma_max * T and sigma_min * T, sigmas shifted by
`shift * s / (1 + (shift - 1) * s)`, a final sigma of 0, and the update x <- x + (sigma_{i+1} - sigma_i) * v.
On an MI355X the update is fused with the head projection and CFG in `nova_head_cfg_euler`; the engine only reads
`timesteps` / `sigmas` from this object. The training side (`sample_timesteps`, `add_noise`) is the torch definition.
"""
import math

import torch

from .. import _torch_ops as ops
from .._compat import BaseOutput, ConfigMixin, SchedulerMixin, register_to_config


class FlowMatchEulerDiscreteSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor


class FlowMatchEulerDiscreteScheduler(SchedulerMixin, ConfigMixin):
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False):
        grid = ops.training_sigma_grid(num_train_timesteps, shift, use_dynamic_shifting)
        self._shift = shift
        self.sigma_max = float(grid[0])
        self.sigma_min = float(grid[-1])
        # training tables (descending); `set_timesteps` replaces both by the sampling grid
        self.sigmas = torch.as_tensor(grid.copy())
        self.timesteps = torch.as_tensor(grid * num_train_timesteps)
        # state written by add_noise (training) / step (sampling)
        self.timestep = None
        self.sigma = None
        self._step_index = None
        self._begin_index = None

    # ---- read-only views and small setters ------------------------------------------------------
    shift = property(lambda self: self._shift)
    step_index = property(lambda self: self._step_index)
    begin_index = property(lambda self: self._begin_index)

    def set_shift(self, shift: float):
        self._shift = shift

    def time_shift(self, mu: float, sigma: float, t):
        e = math.exp(mu)
        return e / (e + (1 / t - 1) ** sigma)

    # ---- sampling -------------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps, mu=None):
        cfg = self.config
        sig = ops.sampling_sigma_grid(self.sigma_max, self.sigma_min, cfg.num_train_timesteps, num_inference_steps)
        sig = self.time_shift(mu, 1.0, sig) if cfg.use_dynamic_shifting else ops.shift_sigmas(sig, self.shift)
        self.num_inference_steps = num_inference_steps
        self.timesteps = sig * cfg.num_train_timesteps
        self.sigmas = sig.tolist() + [0]
        self._step_index = self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        grid = torch.as_tensor(self.timesteps if schedule_timesteps is None else schedule_timesteps)
        hits = (grid == timestep).nonzero()
        return hits[min(1, len(hits) - 1)].item()  # a duplicated timestep resolves to its second occurrence

    def _init_step_index(self, timestep):
        start = self._begin_index
        self._step_index = self.index_for_timestep(timestep) if start is None else start

    def _current_sigma_pair(self, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        i = self._step_index
        return self.sigmas[i], self.sigmas[i + 1]

    def scale_noise(self, sample, timestep, noise):
        sigma, _ = self._current_sigma_pair(timestep)
        return sigma * noise + (1.0 - sigma) * sample

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        sigma, sigma_next = self._current_sigma_pair(timestep)
        prev_sample = model_output * (sigma_next - sigma) + sample
        self._step_index += 1
        if not return_dict:
            return (prev_sample,)
        return FlowMatchEulerDiscreteSchedulerOutput(prev_sample=prev_sample)

    # ---- training -------------------------------------------------------------------------------
    def sample_timesteps(self, size, device=None):
        """Logit-normal discrete timesteps (indices into the descending training tables)."""
        u = torch.normal(0, 1, size, device=device).sigmoid_()
        return u.mul_(self.config.num_train_timesteps).to(dtype=torch.int64)

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = sigma * noise + (1 - sigma) * x_0; remembers `timestep` / `sigma` for the loss."""
        device, dtype = original_samples.device, original_samples.dtype
        self.timestep = self.timesteps.to(device=device)[timesteps]
        sigma = self.sigmas.to(device=device, dtype=dtype)[timesteps]
        self.sigma = sigma.view(timesteps.shape + (1,) * (noise.dim() - timesteps.dim()))
        return self.sigma * noise + (1.0 - self.sigma) * original_samples
