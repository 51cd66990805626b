from .timestep import Timestep
from .hip_replay import HipReplayBuffer, Batch

# name used by the reference's call sites (prism/experience/__init__.py)
TimestepBuffer = HipReplayBuffer
