"""Factory seams mirrored from the reference (``/root/reference/prism/factory``): module names and
function signatures are kept so ``from prism.factory import agent_factory`` becomes
``from prism_amd.factory import agent_factory``."""
