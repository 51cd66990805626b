"""``Timestep`` — the record collectors hand to ``buffer.extend`` (field names as the reference's
``/root/reference/prism/experience/timestep.py:12-28``).  In this build it only exists at the
``extend()`` seam: the row is copied into the HBM ring and the object can die immediately; the
n-step fields are never filled on the host (the gather kernel resolves them per sample)."""
from dataclasses import dataclass
from typing import Any, Optional


@dataclass
class Timestep:
    id: int
    obs: Any = None
    reward: Optional[float] = None
    done: Optional[bool] = None
    truncated: Optional[bool] = None
    action: Optional[int] = None
    n_step_return: Optional[float] = None
    n_step_gamma: Optional[float] = None
    n_step_done: Optional[bool] = None
    needs_n_step: bool = True
    episodic_reward: float = 0
    n_step_next: Any = None
    prev: Any = None      # weakref to the previous Timestep of the same env stream
    next: Any = None      # weakref to the next Timestep, or a strong ref to a truncation node
