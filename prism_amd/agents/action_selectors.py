"""Acting-time action selection (OUT of the HIP hot path; plain torch on whatever device the model
lives on).  Behaviour follows ``/root/reference/prism/agents/action_selectors.py``: greedy
(:70-83), epsilon-greedy with linear anneal (:24-67) and information-directed sampling (:114-192).
Kept only so ``Agent.forward`` / checkpoints keep working when the learner is swapped in.
"""
import numpy as np
import torch


class LinearAnneal:
    """prism/util/annealing_strategies.py:22-33."""

    def __init__(self, start_value, stop_value, max_steps):
        self.start, self.stop, self.max_steps, self.current_step = start_value, stop_value, max_steps, 0

    def update(self, n_steps):
        self.current_step += n_steps
        return self.get_value()

    def get_value(self):
        if self.max_steps == 0 or self.current_step >= self.max_steps:
            return self.stop
        f = min(1, self.current_step / self.max_steps)
        return self.start * (1 - f) + self.stop * f

    def get_state(self):
        return self.current_step

    def set_state(self, state):
        self.current_step = state


class ActionSelector:
    """Instance attributes carry the reference's names (action_selectors.py:4-7,24-33,114-123): ``state.pkl`` pickles
    these objects, and a checkpoint written here has to come back to life as the reference's classes (and vice versa)."""

    def __init__(self, logger=None):
        self.logger = logger
        self.loggables = {}

    def select_action(self, action_probs):
        return torch.argmax(action_probs, dim=-1).long().view(-1)

    def save(self, path):
        pass

    def load(self, path):
        pass

    def log(self, logger, action_value_distribution, q_estimates):
        pass


class GreedyActionSelector(ActionSelector):
    def generate_action_probs(self, action_value_distribution, q_estimates):
        mean_q = q_estimates.mean(dim=-1)
        return torch.nn.functional.one_hot(mean_q.argmax(dim=-1), mean_q.shape[-1])


class EGreedyActionSelector(ActionSelector):
    def __init__(self, e_start, e_stop, anneal_time, seed=123):
        super().__init__()
        self.epsilon = LinearAnneal(e_start, e_stop, anneal_time)
        self.greedy = GreedyActionSelector()
        self.rng = np.random.RandomState(seed)

    def generate_action_probs(self, action_value_distribution, q_estimates):
        n, n_actions = q_estimates.shape[0], q_estimates.shape[1]
        if self.rng.uniform(0, 1) < self.epsilon.update(n):
            a = torch.as_tensor(self.rng.randint(n_actions, size=(n,)), dtype=torch.long)
            return torch.nn.functional.one_hot(a, n_actions)
        return self.greedy.generate_action_probs(action_value_distribution, q_estimates)

    def save(self, path):          # action_selectors.py:47-52
        import os
        with open(os.path.join(path, "epsilon.txt"), "w") as f:
            f.write(str(self.epsilon.get_state()) + "\n")

    def load(self, path):          # action_selectors.py:54-62
        import os
        eps_path = os.path.join(path, "epsilon.txt")
        if os.path.exists(eps_path):
            with open(eps_path) as f:
                self.epsilon.set_state(int(f.readlines()[0]))

    def log(self, logger, action_value_distribution, q_estimates):
        logger.log_data(data=self.epsilon.get_value(), group_name="Report/Action Selector", var_name="Epsilon")


class IDSActionSelector(ActionSelector):
    """score = regret^2 / information gain; ensemble spread gives the regret, the return
    distribution's per-action variance (normalised, floored at rho) gives the gain."""

    def __init__(self, lmbda, random_sample, epsilon, ids_rho_lower_bound, beta, unsquish_function=None):
        super().__init__()
        self.lmbda, self.random_sample, self.epsilon = lmbda, random_sample, epsilon
        self.ids_rho_lower_bound, self.beta, self.unsquish_function = ids_rho_lower_bound, beta, unsquish_function
        self.softmax = torch.nn.Softmax(dim=-1)

    def generate_action_probs(self, action_value_distribution, q_estimates, for_log=False):
        if self.unsquish_function is not None:
            action_value_distribution = self.unsquish_function(action_value_distribution)
            q_estimates = self.unsquish_function(q_estimates)
        mean, spread = q_estimates.mean(dim=-1), q_estimates.std(dim=-1)
        sd = spread.sqrt()
        upper = (mean + self.lmbda * sd).max(dim=-1).values.view(-1, 1)
        regret = upper - (mean - self.lmbda * sd)
        regret_sq = regret.square()
        var_z = action_value_distribution.var(dim=0)
        normalized_var_z = var_z / (self.epsilon + var_z.mean(dim=-1, keepdim=True))
        rho = normalized_var_z.clamp(min=self.ids_rho_lower_bound)
        gain = torch.log(1 + spread / rho) + self.epsilon
        scores = regret_sq / gain
        if self.random_sample:
            probs = self.softmax(-scores).clamp(min=self.epsilon, max=1)
        else:
            probs = torch.nn.functional.one_hot(scores.argmin(dim=-1), scores.shape[-1])
        if for_log:
            self.loggables = {"Action Regret": regret, "Q Estimate Ensemble Mean": mean,
                              "Q Estimate Ensemble Variance": spread, "Return Distribution Variance": var_z,
                              "Information Gain": gain, "IDS Scores": scores, "Action Probs": probs,
                              "Normalized Return Distribution Variance": normalized_var_z}
        return probs

    def select_action(self, action_probs):
        if self.random_sample:
            return torch.multinomial(action_probs, 1, True).long().view(-1)
        return action_probs.argmax(dim=-1).long().view(-1)

    def log(self, logger, action_value_distribution, q_estimates):
        self.generate_action_probs(action_value_distribution, q_estimates, for_log=True)
        for key, value in self.loggables.items():
            logger.log_data(data=[[round(x, 4) for x in row] for row in value.tolist()], group_name="Debug/IDS",
                            var_name=key)
