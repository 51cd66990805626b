"""world_size = 2 on CPU (gloo): the data-parallel scheme keeps replicas bit-identical and equals the
single-process update on the combined minibatch (math checked with the CPU oracle; the HIP agent uses the
same prism_amd.dist helpers around its two native calls)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle.learner_ref import LearnerOracle
    from prism_amd import dist as pdist
    g = H.load_case("iqn_small")
    cfg = H.case_config(g)
    init_seed, replay_seed, tau_seed = pdist.rank_seeds(int(g["seed"]), rank)
    assert init_seed == int(g["seed"]) and (rank == 0 or replay_seed != init_seed)
    sd, tgt = H.build_init_state(cfg, init_seed)
    spec = H.spec_from_config(cfg)
    orc = LearnerOracle(sd, spec, tgt)
    names = list(orc.p.keys())
    B = int(g["B"])
    half = B // world
    out = []
    for step in range(2):
        batch, w, taus = H.case_batch(g, step)
        sl = slice(rank * half, (rank + 1) * half)
        T = 4
        local = {k: v[sl] for k, v in batch.items()}
        ltaus = [t.view(T, B, 1)[:, sl].reshape(-1, 1) for t in taus]
        orc.update(local, w[sl], ltaus, apply=False)       # local loss/grad: mean over the LOCAL batch
        flat = torch.cat([orc.last["grads"][k].reshape(-1) for k in names])
        scale = pdist.allreduce_grads(flat)                  # sum over ranks, scale = 1/world
        assert scale == 1.0 / world
        flat = flat * scale
        off = 0
        for k in names:
            n = orc.p[k].numel()
            orc.p[k].grad = flat[off:off + n].view_as(orc.p[k]).clone()
            off += n
        torch.nn.utils.clip_grad_norm_(list(orc.p.values()), spec.max_grad_norm)
        orc.opt.step()
        pflat = torch.cat([p.detach().reshape(-1) for p in orc.p.values()])
        assert pdist.assert_replicas_identical(pflat)
        out.append(pflat.clone())
    assert pdist.shard_capacity(10_000_000, 8) == 1_250_000
    if rank == 0:
        q.put([o.numpy() for o in out])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    dp = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process on the full batch: mean over B == average of the two half-batch means
    from oracle.learner_ref import LearnerOracle
    g = H.load_case("iqn_small")
    cfg = H.case_config(g)
    sd, tgt = H.build_init_state(cfg, int(g["seed"]))
    orc = LearnerOracle(sd, H.spec_from_config(cfg), tgt)
    for step in range(2):
        batch, w, taus = H.case_batch(g, step)
        orc.update(batch, w, taus)
        ref = torch.cat([p.detach().reshape(-1) for p in orc.p.values()]).numpy()
        np.testing.assert_allclose(dp[step], ref, rtol=0, atol=2e-6)
