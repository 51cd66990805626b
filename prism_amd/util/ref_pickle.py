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


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        module, name = _MAP.get((module, name), (module, name))
        return super().find_class(module, name)


def load(f):
    return _Unpickler(f).load()


def loads(data):
    return load(io.BytesIO(data))


def dumps(obj):
    data = pickle.dumps(obj, protocol=2)
    for (ref_mod, name), (our_mod, _) in _MAP.items():
        data = data.replace(b"c" + our_mod.encode() + b"\n" + name.encode() + b"\n",
                            b"c" + ref_mod.encode() + b"\n" + name.encode() + b"\n")
    return data


def dump(obj, f):
    f.write(dumps(obj))
