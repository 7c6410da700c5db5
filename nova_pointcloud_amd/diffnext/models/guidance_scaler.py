"""Classifier-free guidance bookkeeping: the surface of reference diffnext/models/guidance_scaler.py:21-87.

Row layout contract: guidance passes are stacked along dim 0 as [cond ; uncond (; third pass)].
The MI355X engine folds `scale()` into the head / Euler kernel (`nova_head_cfg_euler`); this class is the host-side
definition used by the PyTorch path and by the engine to derive the per-step guidance values.
"""
import torch

from .. import _torch_ops as ops

_FIELDS = (("guidance_scale", 1), ("guidance_trunc", 0), ("guidance_renorm", 1), ("image_guidance_scale", 0),
           ("spatiotemporal_guidance_scale", 0))


class GuidanceScaler(object):
    def __init__(self, **kwargs):
        for name, default in _FIELDS:
            setattr(self, name, kwargs.get(name, default))
        floor = kwargs.get("min_guidance_scale", None)
        self.min_guidance_scale = floor or self.guidance_scale
        self.inc_guidance_scale = self.guidance_scale - self.min_guidance_scale

    # ---- what kind of guidance is on ----------------------------------------------------------
    @property
    def extra_pass(self) -> bool:
        return (self.image_guidance_scale + self.spatiotemporal_guidance_scale) > 0

    @property
    def num_passes(self) -> int:
        if self.guidance_scale <= 1:
            return 1
        return 3 if self.extra_pass else 2

    def clone(self):
        return GuidanceScaler(**vars(self))

    def decay_guidance_scale(self, decay=0):
        """Linear schedule from `min_guidance_scale` (decay 0) to the initial scale (decay 1)."""
        self.guidance_scale = self.min_guidance_scale + decay * self.inc_guidance_scale

    # ---- batch expansion ----------------------------------------------------------------------
    def expand(self, x: torch.Tensor, padding: torch.Tensor = None) -> torch.Tensor:
        n = self.num_passes
        if n == 1:
            return x
        stacked = torch.stack([x for _ in range(n)])
        if padding is not None and self.image_guidance_scale:
            stacked[1] = padding
        return stacked.flatten(0, 1)

    def expand_text(self, c: torch.Tensor) -> torch.Tensor:
        if not self.extra_pass:
            return c
        cond, uncond = c.chunk(2)
        rows = [cond, uncond]
        if self.image_guidance_scale:
            rows.append(uncond)
        if self.spatiotemporal_guidance_scale:
            rows.append(cond)
        return torch.cat(rows)

    def maybe_disable(self, timestep, *args):
        """Below `guidance_trunc` guidance is switched off for the rest of the denoise loop: keep the cond rows."""
        active = self.guidance_scale > 1 and self.guidance_trunc
        if not (active and float(timestep) < self.guidance_trunc):
            return args
        n = self.num_passes
        self.guidance_scale = 1
        return [a.chunk(n)[0] for a in args]

    # ---- mixing -------------------------------------------------------------------------------
    def renorm(self, x, cond):
        return x if self.guidance_renorm >= 1 else ops.clamp_to_cond_norm(x, cond, self.guidance_renorm)

    def scale(self, x: torch.Tensor) -> torch.Tensor:
        n = self.num_passes
        if n == 1:
            return x
        if n == 2:
            cond, uncond = x.chunk(2)
            return self.renorm(ops.guided(cond, uncond, self.guidance_scale), cond)
        cond, uncond, third = x.chunk(3)
        if self.image_guidance_scale:  # third pass = image-conditioned rows
            mixed = self.renorm(uncond + (cond - third) * self.guidance_scale, cond)
            return mixed + (third - uncond) * self.image_guidance_scale
        mixed = self.renorm(ops.guided(cond, uncond, self.guidance_scale), cond)  # third pass = perturbed rows
        return mixed + (cond - third) * self.spatiotemporal_guidance_scale
