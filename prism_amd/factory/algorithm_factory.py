"""``build_algorithm(config)`` — mirrors ``/root/reference/prism/factory/algorithm_factory.py:3-37``
(seeding order, agent before buffer).  Environment processes are the producer side and out of
scope: pass a ``collector`` that implements the reference collector surface (``get_env_info``,
``signal_processes_start_collecting``, ``collect_timesteps``, ``log``, ``close``), e.g. the
reference's own ``ExperienceCollector`` — or give ``obs_shape`` / ``n_actions`` for a learner that
is fed through ``buffer.extend`` / ``load_arrays`` directly."""
import os
import random

import numpy as np
import torch


def build_algorithm(config, collector=None, obs_shape=None, n_actions=None, process_group=None):
    from prism_amd.factory import agent_factory, exp_buffer_factory
    from prism_amd.util import Checkpointer, Logger

    if collector is not None:
        obs_shape, n_actions, _ = collector.get_env_info()
    if obs_shape is None or n_actions is None:
        raise ValueError("build_algorithm needs a collector or (obs_shape, n_actions)")
    config.redis_side = "server"
    torch.manual_seed(config.seed)
    np.random.seed(config.seed)
    random.seed(config.seed)
    os.makedirs(config.checkpoint_dir, exist_ok=True)
    agent = agent_factory.build_agent(config, obs_shape, n_actions, process_group=process_group)
    if collector is not None:
        collector.signal_processes_start_collecting(agent)
    exp_buffer = exp_buffer_factory.build_exp_buffer(config)
    checkpointer = Checkpointer(os.path.join(config.checkpoint_dir, config.env_name), agent, exp_buffer,
                                config.timesteps_between_evaluations, config.hours_per_checkpoint)
    logger = Logger(config, None)
    return agent, collector, exp_buffer, logger, checkpointer
