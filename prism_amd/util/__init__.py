"""Observability shims the learner loop expects (``/root/reference/prism/util``): a Logger with the
same ``log_data`` / ``report`` surface (wandb optional) and a Checkpointer.  Out of the hot path."""
from .logger import Logger
from .checkpointer import Checkpointer
from prism_amd.agents.action_selectors import LinearAnneal
