"""The reference's replay-buffer checkpoint format: ``experience_buffer/timesteps.pkl``.

``TimestepBuffer.save`` (``/root/reference/prism/experience/timestep_buffer.py:259-296``) pickles ONE flat Python
list, the concatenation of ``Timestep.serialize()`` of every stored timestep in storage order
(``prism/experience/timestep.py:30-100``); ``load`` rebuilds the linked list from it
(``timestep.py:102-277``, ``timestep_buffer.py:298-318``).  Per timestep the list holds

    id, obs block, truncated-successor block, reward, done, truncated, action,
    n_step_return, n_step_gamma, n_step_done, needs_n_step, episodic_reward, n_step_next id, prev id, next id

where an obs block is ``n, v_0 .. v_{n-1}, rank, d_0 .. d_{rank-1}`` or the single value ``NULL_VALUE``, the
truncated-successor block is ``id`` followed by an obs block (or ``NULL_VALUE``), and every ``None`` is written as
``NULL_VALUE`` (-1313).  This module converts between that list and the structure-of-arrays ring of
``HipReplayBuffer`` (pure host code: NumPy in, NumPy out).  The torchrl sampler / writer dumps the reference writes
next to the file are third-party formats that cannot be pinned here; priorities travel in
``hip_sampler.pt`` instead.
"""
import numpy as np

NULL_VALUE = -1313          # timestep.py:9
FLAG_DONE, FLAG_TRUNC, FLAG_HAS_NEXT = 1, 2, 4


def serialize_ring(obs, succ_obs, reward, action, flags, link, back, slot_id, obs_shape):
    """Arrays of the first n ring slots -> the reference's flat list.

    Ids: the stored ``Timestep.id`` of a slot (``slot_id``; slots filled in bulk have none and get numbers above every
    real id, so they cannot collide with one).  A slot whose successor
    exists but is not stored (the collector had not finished it) is written the way the reference's ``save`` does it
    (timestep_buffer.py:276-291): truncated, its successor observation kept in an artificially numbered node."""
    n = int(obs.shape[0])
    base = max([int(v) for v in slot_id[:n] if v >= 0] + [-1]) + 1
    ids = [int(slot_id[i]) if slot_id[i] >= 0 else base + i for i in range(n)]
    fresh = max(ids + [0]) + 1                  # ids of the truncated-successor nodes written for stored truncations
    artificial = -1                             # timestep_buffer.py:267
    shape = [int(s) for s in obs_shape]
    out = []
    for i in range(n):
        f = int(flags[i])
        done, trunc, has_next = bool(f & FLAG_DONE), bool(f & FLAG_TRUNC), bool(f & FLAG_HAS_NEXT)
        out.append(ids[i])
        out.append(int(obs[i].size))
        out += obs[i].reshape(-1).tolist()
        out.append(len(shape))
        out += shape
        next_id = NULL_VALUE
        open_chain = has_next and not trunc and link[i] < 0        # successor not stored: truncate artificially
        if trunc or open_chain:
            if trunc:
                tid = fresh
                fresh += 1
            else:
                tid = artificial
                artificial -= 1
            out.append(tid)
            out.append(int(succ_obs[i].size))
            out += succ_obs[i].reshape(-1).tolist()
            out.append(len(shape))
            out += shape
            next_id = tid
        else:
            out.append(NULL_VALUE)
            if link[i] >= 0:
                next_id = ids[int(link[i])]
        out += [float(reward[i]), done, bool(trunc or open_chain), int(action[i])]
        out += [NULL_VALUE, NULL_VALUE, NULL_VALUE, True, 0]       # n-step cache empty: recomputed on first sample
        out.append(NULL_VALUE)                                      # n_step_next
        out.append(ids[int(back[i])] if back[i] >= 0 else NULL_VALUE)
        out.append(next_id)
    return out


def _obs_block(flat, idx):
    if flat[idx] == NULL_VALUE:
        return None, idx + 1
    n = int(flat[idx])
    vals = flat[idx + 1:idx + 1 + n]
    idx += 1 + n
    rank = int(flat[idx])
    shape = [int(v) for v in flat[idx + 1:idx + 1 + rank]]
    return np.asarray(vals, dtype=np.float32).reshape(shape), idx + 1 + rank


def _opt(v, conv):
    return None if v == NULL_VALUE else conv(v)


def parse_timesteps(flat):
    """The flat list -> records in file order (timestep.py:102-186)."""
    recs, idx = [], 0
    while idx < len(flat):
        r = {"id": int(flat[idx])}
        r["obs"], idx = _obs_block(flat, idx + 1)
        if flat[idx] != NULL_VALUE:
            r["trunc_id"] = int(flat[idx])
            r["trunc_obs"], idx = _obs_block(flat, idx + 1)
        else:
            r["trunc_id"], r["trunc_obs"] = None, None
            idx += 1
        r["reward"], r["done"], r["truncated"] = _opt(flat[idx], float), _opt(flat[idx + 1], bool), _opt(flat[idx + 2], bool)
        r["action"] = _opt(flat[idx + 3], int)
        r["needs_n_step"] = _opt(flat[idx + 7], bool)
        r["n_step_next_id"], r["prev_id"], r["next_id"] = (_opt(flat[idx + 9], int), _opt(flat[idx + 10], int),
                                                           _opt(flat[idx + 11], int))
        idx += 12
        recs.append(r)
    return recs


def ring_from_timesteps(flat):
    """The flat list -> arrays for ``HipReplayBuffer.load_arrays`` (+ ids and the observation shape).

    As in ``Timestep.deserialize_linked_list`` (timestep.py:206-273) a timestep whose ``prev`` / ``n_step_next`` /
    ``next`` id names a timestep that is not in the file is not put back into the storage; the kept ones fill slots
    0 .. n_kept-1 in file order (timestep_buffer.py:314-318).  The reference then serves the kept timesteps from
    the n-step values CACHED in the file (``needs_n_step`` False), which were computed over the chain as it was before
    the save.  The ring has no cache -- it walks the links at sample time -- so the left-out timesteps are loaded too,
    as unsampleable rows BEHIND the kept ones (slots n_kept ..): the walk then sees the chain as it was and arrives at
    the cached values."""
    recs = parse_timesteps(flat)
    known = {r["id"] for r in recs}

    def complete(r):
        links = [r["n_step_next_id"], r["prev_id"]] + ([] if r["trunc_id"] is not None else [r["next_id"]])
        return all(v is None or v in known for v in links)

    order = [r for r in recs if complete(r)]
    n_kept = len(order)
    order += [r for r in recs if not complete(r) and r["reward"] is not None and r["action"] is not None]
    slot_of = {r["id"]: s for s, r in enumerate(order)}
    n = len(order)
    if n_kept == 0:
        raise ValueError("no complete timestep in the file")
    obs_shape = order[0]["obs"].shape
    O = int(np.prod(obs_shape))
    obs, succ = np.zeros((n, O), np.float32), np.zeros((n, O), np.float32)
    reward, action = np.zeros(n, np.float32), np.zeros(n, np.int32)
    flags, link = np.zeros(n, np.uint8), np.full(n, -1, np.int32)
    ids = np.zeros(n, np.int64)
    by_id = {r["id"]: r for r in recs}
    for s, r in enumerate(order):
        ids[s] = r["id"]
        obs[s] = r["obs"].reshape(-1)
        reward[s], action[s] = r["reward"], r["action"]
        f = (FLAG_DONE if r["done"] else 0) | (FLAG_TRUNC if r["truncated"] else 0)
        if r["trunc_id"] is not None:               # strong truncated successor: its observation travels in the record
            f |= FLAG_HAS_NEXT
            succ[s] = r["trunc_obs"].reshape(-1)
        elif r["next_id"] is not None and not r["truncated"] and r["next_id"] in by_id:
            f |= FLAG_HAS_NEXT
            nxt = by_id[r["next_id"]]
            succ[s] = nxt["obs"].reshape(-1)
            # the link is walked by the n-step return only through timesteps that have been completed (timestep_buffer.py:210-218)
            if r["next_id"] in slot_of and nxt["reward"] is not None:
                link[s] = slot_of[r["next_id"]]
        flags[s] = f
    return dict(obs=obs.reshape((n,) + tuple(obs_shape)), succ_obs=succ, reward=reward, action=action, flags=flags, link=link,
                ids=ids, n_kept=n_kept, obs_shape=tuple(int(v) for v in obs_shape))
