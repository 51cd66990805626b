"""GPU: data-parallel learner path.  Two ranks share the one GPU of the test box (gloo carries the
gradient all-reduce through host memory; on a real node the backend is RCCL): replicas must stay
bit-identical while sampling different shards, and a 2-rank run on identical data must equal the
single-replica run."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, same_data, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from prism_amd import dist as pdist
    from prism_amd.agents import hip_agent
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay

    def host_allreduce(flat, group=None):          # gloo over host memory (no RCCL with two ranks on one device)
        h = flat.cpu()
        dist.all_reduce(h)
        flat.copy_(h)
        return 1.0 / world
    hip_agent.pdist.allreduce_grads = host_allreduce

    cfg = baseline_config(2, device="cuda:0", batch_size=32, experience_replay_capacity=2048)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6, process_group=dist.group.WORLD)
    buf, ag = ln.experience_buffer, ln.agent
    assert ag.world == world
    if not same_data:
        _, buf.seed, ag.seed = pdist.rank_seeds(cfg.seed, rank)
    fill_replay(buf, 2048, seed=0 if same_data else rank)
    sums = []
    for step in range(5):
        ln.step()
        torch.cuda.synchronize()
        assert pdist.assert_replicas_identical(ag.flat.cpu())
        sums.append(buf._index.cpu().numpy().copy())
    q.put((rank, ag.flat.cpu().numpy(), np.stack(sums), any(isinstance(g, tuple) and len(g) == 2 for g in ag._graphs.values())))
    dist.barrier()
    dist.destroy_process_group()


def _run(same_data):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, same_data, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240), q.get(timeout=240)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(600)
def test_two_ranks_stay_identical_on_different_shards():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    (r0, p0, i0, g0), (r1, p1, i1, g1) = _run(same_data=False)
    np.testing.assert_array_equal(p0, p1)             # replicas bit-identical
    assert not np.array_equal(i0, i1)                 # ... while sampling different transitions
    assert g0 and g1                                  # the split hipGraph path really ran


@pytest.mark.timeout(600)
def test_two_ranks_on_identical_data_equal_single_replica():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from prism_amd.config import baseline_config
    from prism_amd.learner import Learner
    from prism_amd.synthetic import fill_replay
    (r0, p0, i0, _), (r1, p1, i1, _) = _run(same_data=True)
    np.testing.assert_array_equal(i0, i1)
    cfg = baseline_config(2, device="cuda:0", batch_size=32, experience_replay_capacity=2048)
    ln = Learner()
    with contextlib.redirect_stdout(io.StringIO()):
        ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
    fill_replay(ln.experience_buffer, 2048, seed=0)
    for step in range(5):
        ln.step()
    torch.cuda.synchronize()
    # (g + g) * 0.5 == g exactly; only the clip norm is summed in a different order
    np.testing.assert_allclose(p0, ln.agent.flat.cpu().numpy(), rtol=0, atol=1e-6)
