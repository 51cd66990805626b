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
