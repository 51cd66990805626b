"""The Config mirror keeps the reference's field names, order and preset values (snapshot taken from the
live reference by tools/gen_golden.py config)."""
import dataclasses
import json
import os

from tests import helpers as H


def test_config_fields_and_presets_match_reference():
    from prism_amd.config import Config, DEFAULT_CONFIG, MINATAR_CONFIG, baseline_config, derive
    snap = json.load(open(os.path.join(H.GOLDEN, "config_presets.json")))
    assert [f.name for f in dataclasses.fields(Config)] == snap["fields"]
    assert DEFAULT_CONFIG.__dict__ == snap["DEFAULT_CONFIG"]
    assert MINATAR_CONFIG.__dict__ == snap["MINATAR_CONFIG"]
    # the reference's preset idiom and JSON round trip
    c = Config(**MINATAR_CONFIG.__dict__)
    c.batch_size = 64
    assert MINATAR_CONFIG.batch_size == 32 and Config.deserialize(c.serialize().encode()) == c
    # optional MI355X knobs ride along as plain attributes, never as dataclass fields
    d = derive(MINATAR_CONFIG, hip_graph=False, batch_size=8)
    assert d.hip_graph is False and "hip_graph" not in {f.name for f in dataclasses.fields(Config)}
    for i, (b, per, iqn, ids) in enumerate([(32, False, False, False), (256, True, False, False),
                                            (256, True, True, False), (512, True, True, True),
                                            (512, True, True, True)]):
        cfg = baseline_config(i)
        assert (cfg.batch_size, cfg.use_per, cfg.use_iqn, cfg.use_ids) == (b, per, iqn, ids)


def _listify(d):
    """JSON turns the reference's one-element tuples into lists."""
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in d.items()}


def test_ablation_presets_match_reference():
    from prism_amd.config import ADDITIVE_ABLATION_BASE_CONFIG, SUBTRACTIVE_ABLATION_BASE_CONFIG
    snap = json.load(open(os.path.join(H.GOLDEN, "config_presets.json")))
    assert _listify(ADDITIVE_ABLATION_BASE_CONFIG.__dict__) == snap["ADDITIVE_ABLATION_BASE_CONFIG"]
    assert _listify(SUBTRACTIVE_ABLATION_BASE_CONFIG.__dict__) == snap["SUBTRACTIVE_ABLATION_BASE_CONFIG"]


def test_every_ablation_experiment_stage_is_covered_by_the_kernels():
    """Every configuration the reference's two MinAtar experiment files generate
    (additive_ablation_experiment.py:31-162, subtractive_ablation_experiment.py:30-52; recorded by
    tools/gen_golden.py as the fields that differ from the preset) constructs, and its model shapes at its own
    batch size are ones prism_learner_supported accepts -- i.e. the experiments run unchanged."""
    import contextlib
    import ctypes
    import io
    from prism_amd import _native as N
    from prism_amd import config as C
    from prism_amd.agents.hip_agent import model_dims
    from prism_amd.factory.model_factory import create_model
    snap = json.load(open(os.path.join(H.GOLDEN, "config_presets.json")))
    channels = {"MinAtar/Breakout-v1": 4, "MinAtar/Asterix-v1": 4, "MinAtar/SpaceInvaders-v1": 6,
                "MinAtar/Freeway-v1": 7, "MinAtar/Seaquest-v1": 10}
    L = N.lib()
    n = 0
    for key, base in (("ADDITIVE_STAGES", C.ADDITIVE_ABLATION_BASE_CONFIG),
                      ("SUBTRACTIVE_STAGES", C.SUBTRACTIVE_ABLATION_BASE_CONFIG)):
        for group, diff in snap[key].items():
            cfg = C.derive(base, device="cpu", **diff)
            ch = channels[cfg.env_name]
            if "Breakout" in cfg.env_name:          # building every stage once is enough for the module shells
                with contextlib.redirect_stdout(io.StringIO()):
                    create_model((10, 10, ch), 6, cfg)
            d = model_dims(cfg, ch, 6)
            assert L.prism_learner_supported(ctypes.byref(d), cfg.batch_size) == N.PRISM_OK, group
            n += 1
    assert n >= 40
