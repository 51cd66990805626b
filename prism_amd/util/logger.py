"""Logger — same call surface as ``/root/reference/prism/util/logger.py:5-110``; ``wandb`` is
imported only when ``config.log_to_wandb`` is set (it is not installed on the GPU boxes)."""


class Logger:
    def __init__(self, config=None, holdout_data=None):
        self.wandb_run = None
        if config is not None and getattr(config, "log_to_wandb", False):
            import wandb
            name = config.env_name if config.wandb_run_name == "null" else config.wandb_run_name
            self.wandb_run = wandb.init(project=config.wandb_project_name, group=config.wandb_group_name,
                                        name=name, reinit=True)
            self.wandb_run.config.update(config)
        self.current_data = {}
        self.current_iteration = 0
        self.holdout_data = holdout_data
        self.enabled = True

    def set_holdout_data(self, data):
        self.holdout_data = data

    def enable(self):
        self.enabled = True

    def disable(self):
        self.enabled = False

    def log_data(self, data, group_name, var_name, override_enable=False):
        if override_enable or self.enabled:
            key = "{}-{}".format(group_name, var_name)
            self.current_data[key] = data
            if self.wandb_run is not None:
                self.wandb_run.log({key: data}, commit=False)

    log = log_data

    def override_log(self, data, group_name, var_name):
        self.log_data(data, group_name, var_name, override_enable=True)

    def report(self, iteration=None, float_sig_figs=6):
        it = self.current_iteration if iteration is None else iteration
        groups = {}
        for key, val in self.current_data.items():
            grp, _, var = key.partition("-")
            groups.setdefault(grp, []).append((var, val))
        print("\n" + "=" * 20 + f" iteration {it} " + "=" * 20)
        for grp in sorted(groups):
            print(grp)
            for var, val in groups[grp]:
                if isinstance(val, float):
                    val = f"{val:.{float_sig_figs}g}"
                print(f"    {var}: {val}")
        if self.wandb_run is not None:
            self.wandb_run.log({}, commit=True)
        self.current_iteration = it + 1
        self.current_data.clear()

    def close(self):
        if self.wandb_run is not None:
            self.wandb_run.finish()
