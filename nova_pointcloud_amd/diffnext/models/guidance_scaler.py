"""Classifier-free guidance bookkeeping (reference diffnext/models/guidance_scaler.py:21-87).

Batch layout contract: guidance rows are stacked [cond ; uncond (; third pass)] along dim 0.
The MI355X engine folds `scale()` into the head/Euler kernel (`nova_head_cfg_euler`); this class
is the host-side mirror used by the PyTorch path and to derive the per-step guidance values.
"""
import torch


class GuidanceScaler(object):
    def __init__(self, **kwargs):
        self.guidance_scale = kwargs.get("guidance_scale", 1)
        self.guidance_trunc = kwargs.get("guidance_trunc", 0)
        self.guidance_renorm = kwargs.get("guidance_renorm", 1)
        self.image_guidance_scale = kwargs.get("image_guidance_scale", 0)
        self.spatiotemporal_guidance_scale = kwargs.get("spatiotemporal_guidance_scale", 0)
        self.min_guidance_scale = kwargs.get("min_guidance_scale", None) or self.guidance_scale
        self.inc_guidance_scale = self.guidance_scale - self.min_guidance_scale

    @property
    def extra_pass(self) -> bool:
        return self.image_guidance_scale + self.spatiotemporal_guidance_scale > 0

    @property
    def num_passes(self) -> int:
        return (3 if self.extra_pass else 2) if self.guidance_scale > 1 else 1

    def clone(self):
        return GuidanceScaler(**self.__dict__)

    def decay_guidance_scale(self, decay=0):
        self.guidance_scale = self.inc_guidance_scale * decay + self.min_guidance_scale

    def expand(self, x: torch.Tensor, padding: torch.Tensor = None) -> torch.Tensor:
        if self.guidance_scale <= 1:
            return x
        x = torch.stack([x] * self.num_passes)
        if self.image_guidance_scale and padding is not None:
            x[1] = padding
        return x.flatten(0, 1)

    def expand_text(self, c: torch.Tensor) -> torch.Tensor:
        if not self.extra_pass:
            return c
        parts = list(c.chunk(2))
        if self.image_guidance_scale:
            parts.append(parts[1])
        if self.spatiotemporal_guidance_scale:
            parts.append(parts[0])
        return torch.cat(parts)

    def maybe_disable(self, timestep, *args):
        if self.guidance_scale > 1 and self.guidance_trunc and float(timestep) < self.guidance_trunc:
            passes, self.guidance_scale = self.num_passes, 1
            return [a.chunk(passes)[0] for a in args]
        return args

    def renorm(self, x, cond):
        if self.guidance_renorm >= 1:
            return x
        dims = tuple(range(1, x.dim()))
        ratio = cond.norm(dim=dims, keepdim=True) / x.norm(dim=dims, keepdim=True)
        return x * ratio.clamp(self.guidance_renorm, 1)

    def scale(self, x: torch.Tensor) -> torch.Tensor:
        if self.guidance_scale <= 1:
            return x
        g = self.guidance_scale
        if self.image_guidance_scale:
            cond, uncond, imgcond = x.chunk(3)
            x = self.renorm(uncond + (cond - imgcond) * g, cond)
            return x + (imgcond - uncond) * self.image_guidance_scale
        if self.spatiotemporal_guidance_scale:
            cond, uncond, perturb = x.chunk(3)
            x = self.renorm(uncond + (cond - uncond) * g, cond)
            return x + (cond - perturb) * self.spatiotemporal_guidance_scale
        cond, uncond = x.chunk(2)
        return self.renorm(uncond + (cond - uncond) * g, cond)
