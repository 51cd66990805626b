from .hip_agent import HipAgent

Agent = HipAgent   # name used by the reference's call sites (prism/agents/__init__.py)
