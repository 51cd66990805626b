"""``state.pkl`` interchange with the reference (``/root/reference/prism/agents/agent.py:195-203,222-231``).

The reference pickles its action-selector OBJECTS, so the file names classes by module path
(``prism.agents.action_selectors.*``, ``prism.util.annealing_strategies.LinearAnneal``).  ``prism_amd`` keeps the same
class names and instance attributes under its own package, so:

* ``dump`` writes protocol 2 (class references are newline-terminated text there, no length prefixes) and renames the
  module paths to the reference's: a reference ``Agent.load`` unpickles the file into ITS classes;
* ``load`` resolves the reference's module paths to the local classes, whether or not the reference is importable.
"""
import io
import pickle

_OURS = "prism_amd.agents.action_selectors"
_MAP = {("prism.agents.action_selectors", n): (_OURS, n)
        for n in ("ActionSelector", "GreedyActionSelector", "EGreedyActionSelector", "IDSActionSelector")}
_MAP[("prism.util.annealing_strategies", "LinearAnneal")] = (_OURS, "LinearAnneal")
# the IDS selector keeps its unsquish FUNCTION (action_selectors.py:123): pickled by module path like a class
_SQ = "prism_amd.agents.squish_functions"
_SQ_NAMES = ("symlog", "symexp", "obs_look_further_squish_fn", "obs_look_further_squish_fn_inverse")
for _n in _SQ_NAMES:
    _MAP[("prism.agents.squish_functions", _n)] = (_SQ, _n)


# What a ``state.pkl`` legitimately names besides the selector classes: the NumPy generator of the epsilon-greedy
# selector (action_selectors.py:33), the ``torch.nn.Softmax`` of the IDS selector (:123) and the containers pickle itself
# uses.  Anything else is refused: a checkpoint directory is data, not code.
_ALLOWED = {
    ("builtins", "set"), ("__builtin__", "set"), ("builtins", "frozenset"), ("builtins", "object"), ("__builtin__", "object"),
    ("_codecs", "encode"), ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor"),
    ("collections", "OrderedDict"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.random._mt19937", "MT19937"), ("numpy.random._pickle", "__bit_generator_ctor"),
    ("numpy.random._pickle", "__randomstate_ctor"), ("numpy.random.mtrand", "RandomState"),
    ("numpy.random._pickle", "__generator_ctor"), ("numpy.random._generator", "Generator"),
    ("numpy.random.bit_generator", "SeedSequence"),
    ("torch.nn.modules.activation", "Softmax"),
    # tensors a selector kept for logging (IDSActionSelector.loggables, action_selectors.py:178-192) travel as a nested
    # torch.save blob: the outer rebuild function is harmless, the blob is opened with weights_only (_safe_storage)
    ("torch._utils", "_rebuild_tensor_v2"),
} | {(_OURS, n) for n in ("ActionSelector", "GreedyActionSelector", "EGreedyActionSelector", "IDSActionSelector", "LinearAnneal")} \
  | {(_SQ, n) for n in _SQ_NAMES}


def _safe_storage(b):
    """Stand-in for torch.storage._load_from_bytes (which unpickles the blob without restrictions)."""
    import torch
    return torch.load(io.BytesIO(b), weights_only=True)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _safe_storage
        module, name = _MAP.get((module, name), (module, name))
        if (module, name) not in _ALLOWED:
            raise pickle.UnpicklingError(f"state.pkl names {module}.{name}, which an agent checkpoint has no business naming")
        return super().find_class(module, name)


class _PlainUnpickler(pickle.Unpickler):
    """For files that hold only numbers, booleans, strings and lists / tuples / dicts of them (the replay dump
    ``timesteps.pkl``, timestep_buffer.py:259-296): no class may be named at all."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"this file may not name classes (found {module}.{name})")


def load_plain(f):
    return _PlainUnpickler(f).load()


def load(f):
    return _Unpickler(f).load()


def loads(data):
    return load(io.BytesIO(data))


_REV = {(our_mod, n): (ref_mod, n) for (ref_mod, n), (our_mod, _) in _MAP.items()}


class _RefPickler(pickle._Pickler):
    """The Python pickler with ONE change: a reference to one of the local selector / anneal classes is written under the
    reference's module path (the GLOBAL opcode of protocol 2: ``c<module>\\n<name>\\n``), so that a reference
    ``Agent.load`` unpickles the file into ITS classes.  (The earlier byte replace over the finished stream could have hit
    a string payload that happened to contain the pattern.)"""

    def save_global(self, obj, name=None):
        key = (getattr(obj, "__module__", None), getattr(obj, "__qualname__", None))
        if key in _REV:
            mod, n = _REV[key]
            self.write(pickle.GLOBAL + mod.encode("ascii") + b"\n" + n.encode("ascii") + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)

    # (plain functions reach save_global through the dispatch TABLE, which holds the base class's method: rebind it, or
    # the selector's unsquish function would be written under this package's path)
    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type(_safe_storage)] = save_global


def dumps(obj):
    f = io.BytesIO()
    _RefPickler(f, protocol=2).dump(obj)
    return f.getvalue()


def dump(obj, f):
    f.write(dumps(obj))
