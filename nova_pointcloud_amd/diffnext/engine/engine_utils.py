"""Trainer helpers (reference diffnext/engine/engine_utils.py:26-120, model_ema.py:20-41)."""
import collections
import copy
import random

import numpy as np
import torch
from torch import nn


def count_params(module, trainable=True, unit="M"):
    n = sum(p.numel() for p in module.parameters() if p.requires_grad or not trainable)
    return n / {"M": 1e6, "B": 1e9}[unit]


def freeze_module(module, trainable=False):
    module.train() if trainable else module.eval()
    for p in module.parameters():
        p.requires_grad = trainable
    return module


def get_device(index):
    return torch.device("cuda", index) if torch.cuda.is_available() else torch.device("cpu")


def manual_seed(seed, device_and_seed=None):
    """Seed torch (CPU + the given GPU), numpy and python."""
    torch.manual_seed(seed)
    if device_and_seed is not None and torch.cuda.is_available():
        with torch.cuda.device(device_and_seed[0]):
            torch.cuda.manual_seed(device_and_seed[1])
    np.random.seed(seed)
    random.seed(seed)


def get_param_groups(model):
    """Optimizer groups: normalisation layers and parameters flagged `no_weight_decay` get weight_decay 0; a parameter's
    `lr_scale` attribute becomes the group's `lr_scale` (the trainer multiplies the scheduled rate by it)."""
    norms = (nn.BatchNorm2d, nn.GroupNorm, nn.SyncBatchNorm, nn.LayerNorm)
    seen, groups = set(), collections.OrderedDict()
    for module in model.modules():
        for _, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in seen:
                continue
            seen.add(p)
            attrs = collections.OrderedDict()
            if hasattr(p, "lr_scale"):
                attrs["lr_scale"] = p.lr_scale
            if getattr(p, "no_weight_decay", False) or isinstance(module, norms):
                attrs["weight_decay"] = 0
            key = "/".join(f"{k}:{v}" for k, v in attrs.items())
            groups.setdefault(key, {**attrs, "params": []})["params"].append(p)
    return list(groups.values())


class ModelEMA(nn.Module):
    """float32 exponential moving average of the trainable parameters."""

    def __init__(self, model, decay=0.99, update_every=100, device="gpu"):
        super().__init__()
        self.decay, self.update_every = decay, update_every
        self.model = copy.deepcopy(model).eval()
        for p in self.model.parameters():
            p.data = p.data.float() if p.requires_grad else p.data
            p.requires_grad = False
        if device == "cpu":
            self.model.cpu()

    @torch.no_grad()
    def update(self, model):
        for avg, cur in zip(self.model.parameters(), model.parameters()):
            if cur.requires_grad:
                new = cur.data.float()
                avg.copy_(avg.to(new.device).mul_(self.decay).add_(new, alpha=1 - self.decay))
