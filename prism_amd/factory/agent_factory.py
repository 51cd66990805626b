"""``build_agent(config, obs_shape, n_actions)`` — mirrors
``/root/reference/prism/factory/agent_factory.py:7-61``: online model, optional target model
(constructed second so it advances the torch RNG exactly as the reference does, then overwritten
with the online weights), action selectors, Adam.  RMSprop / SGD are not implemented in the fused
optimizer kernel and fail loudly."""
from prism_amd.agents import action_selectors
from prism_amd.agents.hip_agent import HipAgent
from prism_amd.factory import model_factory


def build_agent(config, obs_shape, n_actions, process_group=None):
    obs_shape = [int(a) for a in obs_shape]
    n_actions = int(n_actions)
    model = model_factory.create_model(obs_shape, n_actions, config)
    target_model = None
    if config.use_target_network:
        target_model = model_factory.create_model(obs_shape, n_actions, config)
        target_model.load_state_dict(model.state_dict())

    eval_selector = action_selectors.GreedyActionSelector()
    if config.use_ids:
        from prism_amd.agents import squish_functions
        _, unsquish = squish_functions.parse(config.loss_squish_fn_id)          # (agent_factory.py:20-27)
        selector = action_selectors.IDSActionSelector(config.ids_lambda, config.ids_use_random_samples,
                                                      config.ids_epsilon, config.ids_rho_lower_bound,
                                                      config.ids_beta, unsquish)
    elif config.use_e_greedy:
        selector = action_selectors.EGreedyActionSelector(config.e_greedy_initial_epsilon,
                                                          config.e_greedy_final_epsilon,
                                                          config.e_greedy_decay_timesteps, config.seed)
    else:
        selector = action_selectors.GreedyActionSelector()
    if not config.use_adam:
        raise model_factory.UnsupportedConfig("prism_amd: only Adam is implemented in the fused optimizer kernel")
    return HipAgent(model, selector, eval_selector, target_model, config, obs_shape[-1], n_actions,
                    process_group=process_group)
