"""prism_amd — MI355X-native replay-sample -> TD-update -> priority-writeback learner path.

Drop-in for the hot path of AechPro/Prism (``prism/learner.py:95-125``): same ``Config`` /
buffer / ``Agent`` / ``Learner`` surface, compute in hand-written gfx950 HIP kernels behind a
C-ABI (``include/prism_hip.h``).  See DESIGN.md.
"""
__version__ = "0.1.0"
