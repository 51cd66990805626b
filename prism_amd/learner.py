"""``Learner`` — same ``configure`` / ``learn`` surface as ``/root/reference/prism/learner.py``,
plus ``step()``: ONE pass of the hot section ``learner.py:95-125`` (sample -> IS weights -> TD
update -> priority writeback -> target sync) with everything resident in HBM.

Quirks kept on purpose: beta is forced to 0.5 every step (learner.py:104-107); the target network
is synced on a *timestep* period (learner.py:122-124).
"""
import time

import numpy as np

from prism_amd.factory import algorithm_factory

_TIMER_KEYS = ("Component Update Time", "Iteration Time", "Agent Update Time", "Batch Sampling Time",
               "Timestep Collection Time")


class Learner:
    def __init__(self):
        self.agent = self.timestep_collector = self.experience_buffer = self.logger = self.checkpointer = None
        self.cumulative_timesteps = 0
        self.cumulative_model_updates = 0
        self.timesteps_since_report = 0
        self.timesteps_since_target_model_update = 0
        self.collected_steps_per_second_ema = None
        self.overall_steps_per_second_ema = None
        self.loggables = {}
        self.per_beta = None
        self.time_phases = True
        self.fused = True
        self.hip_graph = True

    def configure(self, config, collector=None, obs_shape=None, n_actions=None, process_group=None):
        (self.agent, self.timestep_collector, self.experience_buffer, self.logger,
         self.checkpointer) = algorithm_factory.build_algorithm(config, collector, obs_shape, n_actions,
                                                                process_group)
        self.timestep_limit = config.timestep_limit
        self.initial_random_timesteps = config.num_initial_random_timesteps
        self.timesteps_per_iteration = config.timesteps_per_iteration
        self.timesteps_per_report = config.timesteps_per_report
        self.target_network_update_period = config.target_update_period
        self.use_target_network = config.use_target_network
        self.use_per = config.use_per
        if self.use_per:
            from prism_amd.util import LinearAnneal
            self.per_beta = LinearAnneal(config.per_beta_start, config.per_beta_end,
                                         config.per_beta_anneal_timesteps)
        self.cumulative_timesteps = self.cumulative_model_updates = 0
        self.timesteps_since_report = self.timesteps_since_target_model_update = 0
        self.reset_loggables()
        self.device = config.device
        # MI355X-only knobs (not Config fields): fuse the step into four launches / replay it from a hipGraph
        self.fused = bool(getattr(config, "fused_step", True)) and \
            getattr(config, "per_mass_rng", "philox") == "philox" and getattr(config, "tau_rng", "philox") == "philox"
        self.hip_graph = bool(getattr(config, "hip_graph", True))

    # ------------------------------------------------------------------ the hot path
    def step(self, timesteps_this_iteration=0, eager=False):
        """learner.py:95-125.  All work is enqueued on the current HIP stream; nothing here
        synchronises with the device."""
        buf, agent, tp = self.experience_buffer, self.agent, self.time_phases
        if self.fused and hasattr(agent, "step_fused") and hasattr(buf, "flush"):
            buf.flush()
            if self.use_per:
                buf.buffer._sampler._beta = 0.5
            td = agent.step_fused(buf, eager=eager, use_graph=self.hip_graph)
            self.cumulative_model_updates += 1
            self.timesteps_since_target_model_update += timesteps_this_iteration
            if self.use_target_network and \
                    self.timesteps_since_target_model_update >= self.target_network_update_period:
                agent.sync_target_model()
                self.timesteps_since_target_model_update = 0
            return td
        t0 = time.perf_counter() if tp else 0.0
        if self.cumulative_model_updates == 1 and agent.get_static_batch() is not None:
            buf.set_static_batch(agent.get_static_batch())
        batch, info = buf.sample(return_info=True)
        t1 = time.perf_counter() if tp else 0.0
        if self.use_per:
            per_weights = info["_weight"]
            buf.buffer._sampler._beta = 0.5
        else:
            per_weights = 1
        new_per_weights = agent.update(batch, per_weights=per_weights)
        self.cumulative_model_updates += 1
        t2 = time.perf_counter() if tp else 0.0
        if self.use_per:
            buf.update_priority(info["index"], new_per_weights, take_abs=True)
        self.timesteps_since_target_model_update += timesteps_this_iteration
        if self.use_target_network and self.timesteps_since_target_model_update >= self.target_network_update_period:
            agent.sync_target_model()
            self.timesteps_since_target_model_update = 0
        if tp:
            t3 = time.perf_counter()
            self.loggables["Batch Sampling Time"].append(t1 - t0)
            self.loggables["Agent Update Time"].append(t2 - t1)
            self.loggables["Component Update Time"].append(t3 - t2)
        return new_per_weights

    # ------------------------------------------------------------------ the outer loop
    def _learn(self):
        col = self.timestep_collector
        self.cumulative_timesteps = col.collect_timesteps(self.initial_random_timesteps, self.agent,
                                                          self.experience_buffer, random=True)
        self.timesteps_since_report = self.cumulative_timesteps
        self.logger.set_holdout_data(self.experience_buffer.sample(return_info=False).clone())
        self.checkpointer.checkpoint(self.cumulative_timesteps)
        while self.cumulative_timesteps < self.timestep_limit:
            loop_start = time.perf_counter()
            n = col.collect_timesteps(self.timesteps_per_iteration, self.agent, self.experience_buffer)
            dt = time.perf_counter() - loop_start
            self.loggables["Timestep Collection Time"].append(dt)
            if n > 0:
                self.cumulative_timesteps += n
                self.timesteps_since_report += n
                sps = n / dt
                self.collected_steps_per_second_ema = sps if self.collected_steps_per_second_ema is None else \
                    self.collected_steps_per_second_ema * 0.9 + 0.1 * sps
            self.step(n)
            self.checkpointer.checkpoint(self.cumulative_timesteps)
            if self.timesteps_since_report >= self.timesteps_per_report:
                self.report()
            it = time.perf_counter() - loop_start
            self.loggables["Iteration Time"].append(it)
            if n > 0:
                sps = n / it
                self.overall_steps_per_second_ema = sps if self.overall_steps_per_second_ema is None else \
                    self.overall_steps_per_second_ema * 0.9 + 0.1 * sps

    def report(self):
        self._log()
        self.logger.report()
        self.timesteps_since_report = 0
        self.reset_loggables()

    def reset_loggables(self):
        self.loggables = {k: [] for k in _TIMER_KEYS}

    def _log(self):
        self.agent.log(self.logger)
        if self.timestep_collector is not None:
            self.timestep_collector.log(self.logger)
        if not self.loggables["Iteration Time"]:
            return
        if self.use_per:
            self.logger.log_data(self.per_beta.get_value(), "Report/PER", "Beta")
        self.logger.log_data(self.cumulative_timesteps, "Report/Metrics", "Cumulative Timesteps")
        self.logger.log_data(self.cumulative_model_updates, "Report/Model", "Number of Updates")
        self.logger.log_data(self.collected_steps_per_second_ema, "Report/Metrics", "Collected Steps per Second")
        self.logger.log_data(self.overall_steps_per_second_ema, "Report/Metrics", "Overall Steps per Second")
        for key, value in self.loggables.items():
            if value:
                self.logger.log_data(float(np.mean(value)), "Report/Metrics", key)

    def learn(self):
        if self.agent is None:
            print("YOU MUST CONFIGURE THE LEARNER BEFORE CALLING LEARN!")
            return
        if self.timestep_collector is None:
            raise RuntimeError("learn() needs a collector; use step() when feeding the buffer directly")
        try:
            self._learn()
        finally:
            self.checkpointer.save_backup_checkpoint()
            self.experience_buffer.empty()
            self.timestep_collector.close()
            self.logger.close()
