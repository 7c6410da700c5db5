"""Base classes the reference takes from `diffusers` (ConfigMixin, ModelMixin, SchedulerMixin,
DiffusionPipeline, BaseOutput, register_to_config).

The GPU box and the build container have no `diffusers`, so this package carries small
equivalents with the same surface the reference touches: config attr-dict +
`register_to_config`, `.device/.dtype`, `save_pretrained/from_pretrained` of a LOCAL directory
(config.json + safetensors, model_index.json for pipelines). When `diffusers` is importable it is
used instead, so an existing NOVA environment behaves exactly as before.
"""
import functools
import importlib
import inspect
import json
import os
from collections import OrderedDict

import torch
from torch import nn

try:  # pragma: no cover - exercised only where diffusers is installed
    from diffusers.configuration_utils import ConfigMixin, register_to_config
    from diffusers.models.modeling_utils import ModelMixin
    from diffusers.pipelines.pipeline_utils import DiffusionPipeline
    from diffusers.schedulers.scheduling_utils import SchedulerMixin
    from diffusers.utils import BaseOutput

    HAVE_DIFFUSERS = True
except ImportError:
    HAVE_DIFFUSERS = False

    class FrozenConfig(OrderedDict):
        """Read-only attribute dict (what `obj.config` returns)."""

        def __getattr__(self, name):
            try:
                return self[name]
            except KeyError as e:
                raise AttributeError(name) from e

    def _jsonable(v):
        if isinstance(v, (tuple, list)):
            return [_jsonable(i) for i in v]
        if isinstance(v, torch.dtype):
            return str(v)
        return v

    class ConfigMixin(object):
        config_name = "config.json"

        def register_to_config(self, **kwargs):
            cfg = dict(getattr(self, "_internal_dict", {}))
            cfg.update(kwargs)
            self._internal_dict = FrozenConfig(cfg)

        @property
        def config(self):
            return getattr(self, "_internal_dict", FrozenConfig())

        def save_config(self, save_directory):
            os.makedirs(save_directory, exist_ok=True)
            cfg = {"_class_name": type(self).__name__, **{k: _jsonable(v) for k, v in self.config.items()}}
            with open(os.path.join(save_directory, self.config_name), "w") as f:
                json.dump(cfg, f, indent=2)

        @classmethod
        def load_config(cls, directory):
            with open(os.path.join(directory, cls.config_name)) as f:
                cfg = json.load(f)
            return {k: v for k, v in cfg.items() if not k.startswith("_")}

        @classmethod
        def from_config(cls, config, **kwargs):
            accepted = inspect.signature(cls.__init__).parameters
            return cls(**{k: v for k, v in {**dict(config), **kwargs}.items() if k in accepted})

    def register_to_config(init):
        """Decorator: record the constructor's keyword values (defaults included) in `self.config`."""

        @functools.wraps(init)
        def wrapped(self, *args, **kwargs):
            sig = inspect.signature(init)
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            values = {k: v for k, v in list(bound.arguments.items())[1:] if not k.startswith("_")}
            values.pop("kwargs", None)
            init(self, *args, **kwargs)
            ConfigMixin.register_to_config(self, **values)

        return wrapped

    class ModelMixin(nn.Module):
        weights_name = "diffusion_pytorch_model.safetensors"

        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):
            return next(p for p in self.parameters() if p.is_floating_point()).dtype

        def save_pretrained(self, save_directory, **kwargs):
            from safetensors.torch import save_file

            self.save_config(save_directory)
            state = {k: v.detach().contiguous().cpu() for k, v in self.state_dict().items()}
            save_file(state, os.path.join(save_directory, self.weights_name))

        @classmethod
        def from_pretrained(cls, directory, torch_dtype=None, **kwargs):
            from safetensors.torch import load_file

            if not os.path.isdir(directory):
                raise OSError(f"{directory} is not a local directory (no network access from this build)")
            model = cls.from_config(cls.load_config(directory))
            path = os.path.join(directory, cls.weights_name)
            legacy = os.path.join(directory, "diffusion_pytorch_model.bin")  # what the reference's trainer writes (safe_serialization=False)
            if not os.path.exists(path) and os.path.exists(legacy):
                model.load_state_dict(torch.load(legacy, map_location="cpu", weights_only=True))
            else:
                model.load_state_dict(load_file(path))
            return model.to(torch_dtype).eval() if torch_dtype is not None else model.eval()

    class SchedulerMixin(object):
        config_name = "scheduler_config.json"

        def save_pretrained(self, save_directory, **kwargs):
            self.save_config(save_directory)

        @classmethod
        def from_pretrained(cls, directory, **kwargs):
            return cls.from_config(cls.load_config(directory))

    class BaseOutput(OrderedDict):
        """Output container: fields given as keywords, readable as attributes, keys and by index."""

        def __init__(self, **fields):
            super().__init__()
            for name in getattr(type(self), "__annotations__", {}):
                fields.setdefault(name, None)
            for k, v in fields.items():
                if v is not None:
                    self[k] = v
                object.__setattr__(self, k, v)

        def __getitem__(self, k):
            if isinstance(k, str):
                return dict(self.items())[k]
            return self.to_tuple()[k]

        def to_tuple(self):
            return tuple(self[k] for k in self.keys())

    class DiffusionPipeline(ConfigMixin):
        config_name = "model_index.json"

        def __init__(self):
            self._device, self._dtype = torch.device("cpu"), torch.float32

        def _modules(self):
            return [m for m in vars(self).values() if isinstance(m, nn.Module)]

        @property
        def device(self):
            for m in self._modules():
                return next(m.parameters()).device
            return self._device

        @property
        def dtype(self):
            for m in self._modules():
                return next(p for p in m.parameters() if p.is_floating_point()).dtype
            return self._dtype

        def to(self, *args, **kwargs):
            for m in self._modules():
                m.to(*args, **kwargs)
            return self

        def save_pretrained(self, save_directory, **kwargs):
            os.makedirs(save_directory, exist_ok=True)
            index = {"_class_name": type(self).__name__}
            for name, comp in vars(self).items():
                if hasattr(comp, "save_pretrained") and not name.startswith("_"):
                    comp.save_pretrained(os.path.join(save_directory, name))
                    index[name] = [type(comp).__module__, type(comp).__name__]
            with open(os.path.join(save_directory, self.config_name), "w") as f:
                json.dump(index, f, indent=2)

        @classmethod
        def from_pretrained(cls, directory, torch_dtype=None, **kwargs):
            """Load from a LOCAL pipeline directory: `model_index.json` ({component: [module, class]}, the diffusers layout
            the reference's checkpoints use) + one sub-directory per component."""
            if not os.path.isdir(directory):
                raise OSError(f"{directory} is not a local directory (no network access from this build)")
            with open(os.path.join(directory, cls.config_name)) as f:
                index = json.load(f)
            comps = {}
            optional = set(getattr(cls, "_optional_components", ()))
            for name, spec in index.items():
                if name.startswith("_") or not isinstance(spec, list) or spec[0] is None:
                    continue
                comp_dir = os.path.join(directory, name)
                try:
                    klass = getattr(importlib.import_module(spec[0]), spec[1])
                except (ImportError, AttributeError):
                    # a component of a library that is not installed here (diffusers' VAE, transformers' text encoder):
                    # optional ones are left out (the point-set path needs the transformer and the scheduler only)
                    if name in optional:
                        continue
                    raise
                args = {"torch_dtype": torch_dtype} if issubclass(klass, nn.Module) else {}
                comps[name] = klass.from_pretrained(comp_dir, **args)
            accepted = inspect.signature(cls.__init__).parameters
            return cls(**{k: v for k, v in comps.items() if k in accepted})
