"""Pipelines (reference diffnext/pipelines/__init__.py:18 exports NOVAPipeline)."""
from .nova import NOVAPipeline  # noqa: F401
