"""Samplers (reference diffnext/schedulers)."""
from .scheduling_cfm import FlowMatchEulerDiscreteScheduler  # noqa: F401
from .scheduling_ddpm import DDPMScheduler  # noqa: F401
