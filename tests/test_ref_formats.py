"""Checkpoint interchange with the reference, CPU part: the fixtures under tests/golden/ref_checkpoint
and tests/golden/ref_buffer were WRITTEN BY THE REFERENCE (tools/gen_golden.py checkpoint / nstep: Agent.save,
TimestepBuffer.save); nothing here imports it."""
import os
import pickle
import sys

import numpy as np

from tests import helpers as H

CK = os.path.join(H.GOLDEN, "ref_checkpoint")
BUF = os.path.join(H.GOLDEN, "ref_buffer")


def test_reference_state_pkl_unpickles_without_the_reference():
    assert not any(m == "prism" or m.startswith("prism.") for m in sys.modules)      # the reference is not around
    from prism_amd.agents import action_selectors as S
    from prism_amd.util import ref_pickle
    exp = np.load(os.path.join(H.GOLDEN, "ref_checkpoint_expected.npz"))
    with open(os.path.join(CK, "agent", "state.pkl"), "rb") as f:
        st = ref_pickle.load(f)
    assert isinstance(st["action_selector"], S.EGreedyActionSelector)
    assert isinstance(st["action_selector"].epsilon, S.LinearAnneal)
    assert st["action_selector"].epsilon.get_state() == int(exp["epsilon_step"])
    assert isinstance(st["eval_action_selector"], S.GreedyActionSelector)
    assert st["n_updates"] == int(exp["n_updates"]) and st["max_grad_norm"] == float(exp["max_grad_norm"])
    # the object works: epsilon anneals from the restored step, the generator state came along
    sel = st["action_selector"]
    e0 = sel.epsilon.get_value()
    sel.epsilon.update(1000)
    assert sel.epsilon.get_value() < e0
    with open(os.path.join(CK, "state_ids.pkl"), "rb") as f:
        st2 = ref_pickle.load(f)
    ids = st2["action_selector"]
    assert isinstance(ids, S.IDSActionSelector)
    assert (ids.lmbda, ids.random_sample, ids.ids_rho_lower_bound, ids.beta) == (0.1, False, 0.25, 0.8)


def test_state_pkl_written_here_names_the_reference_classes():
    from prism_amd.agents import action_selectors as S
    from prism_amd.util import ref_pickle
    state = {"action_selector": S.EGreedyActionSelector(1.0, 0.01, 1000), "eval_action_selector": S.GreedyActionSelector(),
             "ids": S.IDSActionSelector(0.1, False, 1e-10, 0.25, 0.8), "n_updates": 3, "max_grad_norm": 10.0,
             "use_cuda_graph": False}
    state["action_selector"].epsilon.update(17)
    data = ref_pickle.dumps(state)
    assert b"prism_amd" not in data
    for name in (b"EGreedyActionSelector", b"GreedyActionSelector", b"IDSActionSelector"):
        assert b"cprism.agents.action_selectors\n" + name + b"\n" in data
    assert b"cprism.util.annealing_strategies\nLinearAnneal\n" in data
    back = ref_pickle.loads(data)
    assert back["action_selector"].epsilon.get_state() == 17 and back["n_updates"] == 3
    # and the reference's own instance attributes are all there (its methods run on this state)
    ref_state = ref_pickle.load(open(os.path.join(CK, "agent", "state.pkl"), "rb"))
    assert set(vars(ref_state["action_selector"])) == set(vars(state["action_selector"]))
    ref_ids = ref_pickle.load(open(os.path.join(CK, "state_ids.pkl"), "rb"))["action_selector"]
    assert set(vars(ref_ids)) == set(vars(state["ids"]))


def test_selector_unsquish_function_travels_by_the_reference_path():
    """An IDS selector built with a value squish keeps its unsquish FUNCTION (agent_factory.py:20-27); pickle writes a
    function by module path, so the file must name the reference's module and load back as the local function."""
    import torch
    from prism_amd.agents import action_selectors as S, squish_functions as Q
    from prism_amd.util import ref_pickle
    for sid, fn in (("symlog", Q.symexp), ("obs_look_further", Q.obs_look_further_squish_fn_inverse)):
        squish, unsquish = Q.parse(sid)
        assert unsquish is fn and Q.unsquish_id(unsquish) == Q.SQUISH_IDS[sid]
        x = torch.linspace(-7.0, 9.0, 33)
        torch.testing.assert_close(unsquish(squish(x)), x, rtol=2e-5, atol=2e-5)          # a pair of inverses
        data = ref_pickle.dumps({"action_selector": S.IDSActionSelector(0.1, False, 1e-10, 0.25, 0.8, unsquish)})
        assert b"prism_amd" not in data and b"cprism.agents.squish_functions\n" + fn.__name__.encode() + b"\n" in data
        assert ref_pickle.loads(data)["action_selector"].unsquish_function is fn
    assert Q.parse("none") == (None, None) and Q.parse("anything else") == (None, None) and Q.unsquish_id(None) == 0


def check_next_obs(next_obs, ring, recs, exp):
    """Next observations of the sampleable rows vs the reference's own post-load collate.  One artefact of the
    reference's load is NOT reproduced: a kept timestep whose cached ``n_step_next`` names a timestep the load left
    out holds a dead weak reference afterwards, and the collate then bootstraps from the timestep's OWN observation
    (timestep_buffer.py:145-159 with a non-terminal flag).  The ring keeps the true successor observation, which is
    in the file."""
    nk = int(ring["n_kept"])
    kept = set(int(i) for i in ring["ids"][:nk])
    by_id = {r["id"]: r for r in recs}
    n_art = 0
    for i in range(nk):
        r = by_id[int(ring["ids"][i])]
        want = exp["next_obs"][i].reshape(-1)
        nsn = r["n_step_next_id"]
        if nsn is not None and nsn not in kept and nsn in by_id and not np.array_equal(next_obs[i], want):
            np.testing.assert_array_equal(want, exp["obs"][i].reshape(-1))               # the artefact
            np.testing.assert_array_equal(next_obs[i], by_id[nsn]["obs"].reshape(-1))    # the true successor
            n_art += 1
        else:
            np.testing.assert_array_equal(next_obs[i], want, err_msg=f"row {i}")
    assert n_art <= nk // 2


def test_reference_buffer_file_parses_and_keeps_what_the_reference_keeps():
    from prism_amd.experience import ref_format
    flat = pickle.load(open(os.path.join(BUF, "experience_buffer", "timesteps.pkl"), "rb"))
    recs = ref_format.parse_timesteps(flat)
    g = np.load(os.path.join(H.GOLDEN, "nstep_chain.npz"))
    assert len(recs) == int(g["N"])
    np.testing.assert_array_equal(np.stack([r["obs"] for r in recs]), g["obs"])
    ring = ref_format.ring_from_timesteps(flat)
    exp = np.load(os.path.join(H.GOLDEN, "ref_buffer_expected.npz"))
    nk = int(ring["n_kept"])
    np.testing.assert_array_equal(ring["ids"][:nk], exp["ids"])     # exactly the timesteps the reference's load keeps
    assert len(ring["ids"]) > nk                                    # the others ride along as link targets only
    # the sampleable rows, walked over the restored chain, give what the reference serves from its cached values
    from oracle import per_ref
    nall = len(ring["ids"])
    rp = per_ref.ReplayOracle(128, 400, 3, 0.99, 0.5, 0.5, use_per=False)
    rp.obs[:nall], rp.succ_obs[:nall] = ring["obs"].reshape(nall, -1), ring["succ_obs"]
    rp.reward[:nall], rp.action[:nall], rp.flags[:nall], rp.link[:nall] = ring["reward"], ring["action"], ring["flags"], ring["link"]
    got = rp.gather(np.arange(nk))
    check_next_obs(got["next_obs"], ring, recs, exp)
    np.testing.assert_array_equal(got["reward"], exp["reward"].reshape(-1))
    np.testing.assert_array_equal(got["gamma"], exp["gamma"].reshape(-1))
    np.testing.assert_array_equal(got["nonterminal"].astype(bool), exp["nonterminal"].reshape(-1))
    # our writer -> our reader is the identity on what it was given
    ring = {k: (v[:nk] if isinstance(v, np.ndarray) else v) for k, v in ring.items()}
    ring["link"] = np.where(ring["link"] < nk, ring["link"], -1)
    n = nk
    back = np.full(n, -1, np.int32)
    for s, l in enumerate(ring["link"]):
        if l >= 0:
            back[l] = s
    flat2 = ref_format.serialize_ring(ring["obs"].reshape(n, -1), ring["succ_obs"], ring["reward"], ring["action"],
                                      ring["flags"], ring["link"], back, ring["ids"], ring["obs_shape"])
    ring2 = ref_format.ring_from_timesteps(flat2)
    for k in ("obs", "succ_obs", "reward", "action", "link", "ids"):
        np.testing.assert_array_equal(ring2[k], ring[k], err_msg=k)
    # flags: a stored slot whose successor exists but is not stored is written truncated, as the reference's save does
    trunc_added = (ring2["flags"] & 2) & ~(ring["flags"] & 2)
    np.testing.assert_array_equal(ring2["flags"] & ~np.uint8(2), ring["flags"] & ~np.uint8(2))
    assert np.all((trunc_added == 0) | ((ring["flags"] & 4 != 0) & (ring["link"] < 0)))
