"""Training engine (reference diffnext/engine): learning-rate schedules, parameter groups, EMA and the Trainer."""
from .train_engine import Trainer  # noqa: F401
