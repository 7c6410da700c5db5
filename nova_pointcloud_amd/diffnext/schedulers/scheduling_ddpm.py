"""DDPM ancestral sampler (reference diffnext/schedulers/scheduling_ddpm.py:75-354, itself the
diffusers scheduler). Secondary sampler of the reference (used by its stand-alone point-cloud
scripts); PyTorch definition only — the MI355X engine builds the flow-matching Euler sampler and
raises NotImplementedError for this one.

RNG contract (:303-312): one fresh gaussian of the sample's shape per step with t > 0.
"""
import math

import numpy as np
import torch

from .._compat import BaseOutput, ConfigMixin, SchedulerMixin, register_to_config


class DDPMSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor
    pred_original_sample: torch.FloatTensor


def _cosine_betas(n, max_beta=0.999):
    bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - bar((i + 1) / n) / bar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


class DDPMScheduler(SchedulerMixin, ConfigMixin):
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False):
        n = num_train_timesteps
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start**0.5, beta_end**0.5, n, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            betas = _cosine_betas(n)
        elif beta_schedule == "sigmoid":
            betas = torch.sigmoid(torch.linspace(-6, 6, n)) * (beta_end - beta_start) + beta_start
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:  # zero terminal SNR (arXiv 2305.08891, algorithm 1)
            abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
            first, last = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
            abar_sqrt = (abar_sqrt - last) * first / (first - last)
            abar = abar_sqrt**2
            betas = 1 - torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        self.betas, self.alphas = betas, 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one, self.init_noise_sigma = torch.tensor(1.0), 1.0
        self.custom_timesteps, self.num_inference_steps = False, None
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy())
        self.variance_type = variance_type

    def scale_model_input(self, sample, timestep=None):
        return sample

    def sample_timesteps(self, size, device=None):
        return torch.randint(0, self.config.num_train_timesteps, size, device=device)

    def get_velocity(self, sample, noise, timesteps):
        acp = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        shape = timesteps.shape + (1,) * (noise.dim() - timesteps.dim())
        return acp[timesteps].sqrt().view(shape) * noise - (1 - acp[timesteps]).sqrt().view(shape) * sample

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        n = self.config.num_train_timesteps
        if timesteps is not None:
            self.custom_timesteps, ts = True, np.array(timesteps, dtype=np.int64)
        else:
            if num_inference_steps > n:
                raise ValueError("num_inference_steps exceeds num_train_timesteps")
            self.num_inference_steps, self.custom_timesteps = num_inference_steps, False
            spacing = self.config.timestep_spacing
            if spacing == "linspace":
                ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
            elif spacing == "leading":
                ratio = n // num_inference_steps
                ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
                ts += self.config.steps_offset
            elif spacing == "trailing":
                ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
            else:
                raise ValueError(f"unknown timestep_spacing {spacing}")
        self.timesteps = torch.from_numpy(ts).to(device)

    def previous_timestep(self, timestep):
        if self.custom_timesteps:
            idx = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            return torch.tensor(-1) if idx == self.timesteps.shape[0] - 1 else self.timesteps[idx + 1]
        steps = self.num_inference_steps or self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // steps

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_t = 1 - a_t / a_prev
        var = torch.clamp((1 - a_prev) / (1 - a_t) * beta_t, min=1e-20)
        kind = variance_type or self.config.variance_type
        if kind == "fixed_small":
            return var
        if kind == "fixed_small_log":
            return torch.exp(0.5 * torch.log(var))
        if kind == "fixed_large":
            return beta_t
        if kind == "fixed_large_log":
            return torch.log(beta_t)
        if kind == "learned":
            return predicted_variance
        if kind == "learned_range":
            frac = (predicted_variance + 1) / 2
            return frac * torch.log(beta_t) + (1 - frac) * torch.log(var)
        return var

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        t, prev_t = timestep, self.previous_timestep(timestep)
        predicted_variance = None
        if model_output.shape[1] == sample.shape[1] * 2 and self.variance_type in ("learned", "learned_range"):
            model_output, predicted_variance = torch.split(model_output, sample.shape[1], dim=1)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        kind = self.config.prediction_type
        if kind == "epsilon":
            x0 = (sample - b_t**0.5 * model_output) / a_t**0.5
        elif kind == "sample":
            x0 = model_output
        elif kind == "v_prediction":
            x0 = a_t**0.5 * sample - b_t**0.5 * model_output
        else:
            raise ValueError(f"unknown prediction_type {kind}")
        # NB: like the reference's step (:268-300) there is no clip / dynamic-threshold stage; `clip_sample`,
        # `thresholding` & co. are accepted config keys only.
        mean = (a_prev**0.5 * cur_beta / b_t) * x0 + (cur_alpha**0.5 * b_prev / b_t) * sample
        if t > 0:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            if self.variance_type == "fixed_small_log":
                mean = mean + self._get_variance(t, predicted_variance) * noise
            elif self.variance_type == "learned_range":
                mean = mean + torch.exp(0.5 * self._get_variance(t, predicted_variance)) * noise
            else:
                mean = mean + self._get_variance(t, predicted_variance) ** 0.5 * noise
        return DDPMSchedulerOutput(prev_sample=mean, pred_original_sample=x0) if return_dict else (mean,)

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        shape = timesteps.shape + (1,) * (noise.dim() - timesteps.dim())
        return acp[timesteps].sqrt().view(shape) * original_samples + (1 - acp[timesteps]).sqrt().view(shape) * noise

    def __len__(self):
        return self.config.num_train_timesteps
