"""Pipeline output container and component registration (reference pipeline_utils.py:26-64)."""
import torch

from ..._compat import BaseOutput


class NOVAPipelineOutput(BaseOutput):
    """`images` ([B, ...] 4-D results) or `frames` (5-D results: [B, C, T, H, W] latents / point sets)."""

    images: object
    frames: object


class PipelineMixin(object):
    def register_module(self, model_or_path, name) -> torch.nn.Module:
        """Record the component's class name in the pipeline config; paths are loaded with from_pretrained."""
        model = model_or_path
        if isinstance(model_or_path, str):
            cls = self.__init__.__annotations__[name]
            model = cls.from_pretrained(model_or_path, torch_dtype=self.dtype) if model_or_path else cls()
            model = model.to(self.device) if isinstance(model, torch.nn.Module) else model
        self.register_to_config(**{name: model.__class__.__name__})
        return model
