"""``build_exp_buffer(config)`` — mirrors ``/root/reference/prism/factory/exp_buffer_factory.py:
10-36``: prioritized or uniform replay of ``experience_replay_capacity`` items, batch size, alpha
and beta from the config — here an HBM-resident ring instead of torchrl ListStorage."""
from prism_amd.experience import HipReplayBuffer


def build_exp_buffer(config, capacity=None):
    if getattr(config, "run_through_redis", False):
        raise NotImplementedError("prism_amd: the Redis-decoupled buffer (async_components) is out of scope")
    return HipReplayBuffer(capacity=capacity or config.experience_replay_capacity, batch_size=config.batch_size,
                           device=config.device, frame_stack=config.frame_stack_size,
                           n_step=config.n_step_returns_length, gamma=config.gamma, use_per=config.use_per,
                           alpha=config.per_alpha, beta=config.per_beta_start,
                           mass_rng=getattr(config, "per_mass_rng", "philox"), seed=config.seed,
                           strict=getattr(config, "per_strict", False))
