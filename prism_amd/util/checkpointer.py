"""Checkpointer with the reference's interface and on-disk layout
(``/root/reference/prism/util/checkpointer.py:8-51``): an agent checkpoint
``<save_dir>/agent_checkpoint_<timesteps>/agent/{model.pt,optimizer.pt,target_model.pt,state.pkl}`` every
``timesteps_per_agent_checkpoint`` timesteps; the hourly backup checkpoint is disabled in the reference by an early
``return`` (:23-24) and is a no-op here too; ``load_checkpoint`` restores agent and experience buffer."""
import os
import time


class Checkpointer:
    def __init__(self, save_dir, agent, experience_buffer, timesteps_per_agent_checkpoint, hours_per_backup):
        self.save_dir = save_dir
        self.agent = agent
        self.experience_buffer = experience_buffer
        self.timesteps_per_agent_checkpoint = timesteps_per_agent_checkpoint
        self.last_backup_checkpoint_time = time.time()
        self.last_agent_checkpoint_timesteps = 0
        self.seconds_per_backup = hours_per_backup * 60 * 60

    def load_checkpoint(self, checkpoint_dir):
        self.agent.load(checkpoint_dir)
        self.experience_buffer.load(checkpoint_dir)

    def save_backup_checkpoint(self):
        return

    def save_agent_checkpoint(self):
        path = os.path.join(self.save_dir, f"agent_checkpoint_{self.last_agent_checkpoint_timesteps}")
        self.agent.save(path)

    def checkpoint(self, timesteps):
        now = time.time()
        if now - self.last_backup_checkpoint_time > self.seconds_per_backup:
            self.save_backup_checkpoint()
            self.last_backup_checkpoint_time = now
        if timesteps - self.last_agent_checkpoint_timesteps >= self.timesteps_per_agent_checkpoint:
            self.last_agent_checkpoint_timesteps = timesteps
            self.save_agent_checkpoint()
