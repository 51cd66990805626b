"""Value-squish pairs of the TD target (``/root/reference/prism/agents/squish_functions.py:4-18``, selected by
``loss_squish_fn_id`` through ``model_factory.py:16-23``): the target is ``squish(r + gamma' * unsquish(z_next))``
(``iqn_model.py:141-148``, ``q_ensemble.py:77-82``) and the information-directed selector looks at unsquished estimates
(``action_selectors.py:128-130``).  The learner applies them inside the HIP loss kernels (csrc/common.h ``squish_value`` /
``unsquish_value``); these torch forms serve the selectors and keep ``state.pkl`` interchangeable (the reference pickles
the selector's function by module path)."""
import torch

_EPS = 0.01        # the epsilon of Pohlen et al., "Observe and Look Further" (arXiv:1805.11593), as the reference fixes it


def symlog(x):
    return x.sign() * (x.abs() + 1).log()


def symexp(x):
    return x.sign() * (x.abs().exp() - 1)


def obs_look_further_squish_fn(x):
    return x.sign() * ((x.abs() + 1).sqrt() - 1) + _EPS * x


def obs_look_further_squish_fn_inverse(y):
    root = (1 + 4 * _EPS * (y.abs() + 1 + _EPS)).sqrt()
    return y.sign() * (((root - 1) / (2 * _EPS)).square() - 1)


SQUISH_IDS = {"none": 0, "obs_look_further": 1, "symlog": 2}       # include/prism_hip.h PRISM_SQUISH_*


def parse(squish_fn_id):
    """(squish, unsquish) or (None, None): anything but the two known ids parses to no squish, as in the reference."""
    if squish_fn_id == "obs_look_further":
        return obs_look_further_squish_fn, obs_look_further_squish_fn_inverse
    if squish_fn_id == "symlog":
        return symlog, symexp
    return None, None


def unsquish_id(fn):
    """PRISM_SQUISH_* id of a selector's unsquish function (None -> 0), or None for a function the kernels do not know."""
    if fn is None:
        return 0
    if fn is obs_look_further_squish_fn_inverse:
        return SQUISH_IDS["obs_look_further"]
    if fn is symexp:
        return SQUISH_IDS["symlog"]
    return None
