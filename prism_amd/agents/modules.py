"""Parameter containers with the reference's ``state_dict`` layout.

The HIP learner owns one flat fp32 parameter buffer; these ``nn.Module`` shells exist so that
(a) ``torch.manual_seed(seed)`` + construction reproduces the reference's initial weights exactly
(same layer types created in the same order: ``/root/reference/prism/factory/model_factory.py:
54-145``), (b) ``state_dict()`` keys/shapes interchange with reference checkpoints (SURVEY.md
Appendix B), and (c) the acting path (``Agent.forward``, out of scope for the HIP kernels) has a
plain torch forward to run.  The TD update never calls these ``forward`` methods.

Layout restated from: minatar_cnn_model.py:7-46, iqn_model.py:6-46, ffnn_model.py:46-81,
q_ensemble.py:6-48, composite_model.py:7-70.
"""
import numpy as np
import torch
import torch.nn as nn


def _as_tensor(x, device):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32))
    return x.float().to(device)


class LayerStack(nn.Module):
    """[LN] Linear [ReLU] ... under attribute ``model`` (an ``nn.Sequential``), so keys read
    ``<prefix>.model.<i>.weight`` like the reference's FFNNModel."""

    def __init__(self, n_in, n_out, n_layers, width, use_layer_norm, norm_first_layer=True,
                 final_relu=False, device="cpu"):
        super().__init__()
        dims = [n_in] + [width] * (n_layers - 1) + [n_out]
        seq = []
        for i in range(n_layers):
            if use_layer_norm and (i > 0 or norm_first_layer):
                seq.append(nn.LayerNorm(dims[i]))
            seq.append(nn.Linear(dims[i], dims[i + 1]))
            if i < n_layers - 1:
                seq.append(nn.ReLU())
        if final_relu:
            seq.append(nn.ReLU())
        self.model = nn.Sequential(*seq).to(device)

    def forward(self, x):
        return self.model(x.reshape(x.shape[0], -1))


class MinAtarEmbed(nn.Module):
    """Conv2d(C->16, 3x3, stride 1, no pad) + ReLU + flatten: (B,10,10,C) -> (B,1024)."""

    def __init__(self, in_channels, device="cpu"):
        super().__init__()
        self.in_channels = in_channels
        self.model = nn.Sequential(nn.Conv2d(in_channels, 16, kernel_size=3, stride=1), nn.ReLU(),
                                   nn.Flatten()).to(device)
        self.output_dim = 16 * 8 * 8

    def forward(self, x):
        return self.model(x.permute(0, 3, 1, 2).float())


class IQNHead(nn.Module):
    def __init__(self, n_in, n_actions, n_basis, use_layer_norm, n_layers, width, n_tau, n_tau_next,
                 n_tau_act, huber_k, double_q, loss_weight, propagate_grad, device="cpu"):
        super().__init__()
        self.device = device
        self.n_actions, self.n_basis = n_actions, n_basis
        self.n_current_quantile_samples, self.n_next_quantile_samples = n_tau, n_tau_next
        self.n_quantile_samples_per_action = n_tau_act
        self.huber_k, self.use_double_q_learning = huber_k, double_q
        self.distributional_loss_weight, self.propagate_grad = loss_weight, propagate_grad
        self.cos_basis_range = torch.arange(1, n_basis + 1, 1, device=device)
        self.phi = nn.Sequential(nn.Linear(n_basis, n_in), nn.ReLU()).to(device)
        self.model = None
        if n_layers > 0:
            self.model = LayerStack(n_in, width, n_layers, width, use_layer_norm, final_relu=True,
                                    device=device)
            n_in = width
        if use_layer_norm:
            self.embedding_to_quantile_layer = nn.Sequential(nn.LayerNorm(n_in),
                                                             nn.Linear(n_in, n_actions)).to(device)
        else:
            # (the reference passes device= to the constructor here, iqn_model.py:46, so on a GPU it draws this one
            # layer's initial weights from the CUDA generator; created on the host like every other layer, the
            # initial state is the same on any device and equals the reference's CPU run seed for seed)
            self.embedding_to_quantile_layer = nn.Linear(n_in, n_actions).to(device)

    @torch.no_grad()
    def forward(self, e, n_quantile_samples=None, for_action=False):
        """Acting-time forward only (torch ops; iqn_model.py:48-87)."""
        e = e.reshape(e.shape[0], -1)
        T = self.n_quantile_samples_per_action if for_action else n_quantile_samples
        taus = torch.rand([T * e.shape[0], 1], device=e.device)
        c = torch.cos(torch.tile(taus, [1, self.n_basis]) * self.cos_basis_range.to(e.device) * np.pi)
        h = self.phi(c) * torch.tile(e, [T, 1])
        if self.model is not None:
            h = self.model(h)
        z = self.embedding_to_quantile_layer(h)
        return z.view(T, -1, self.n_actions) if for_action else (z, taus)


class QHeads(nn.Module):
    def __init__(self, n_in, n_actions, n_heads, use_layer_norm, n_layers, width, double_q, loss_weight,
                 variation_coef, device="cpu"):
        super().__init__()
        self.device, self.n_heads = device, n_heads
        self.use_double_q_learning, self.q_loss_weight = double_q, loss_weight
        self.ensemble_variation_coef = variation_coef
        self.theil = torch.tensor(0.0, device=device)
        heads = []
        for _ in range(n_heads):
            if n_layers > 0:
                heads.append(LayerStack(n_in, n_actions, n_layers, width, use_layer_norm, device=device))
            elif use_layer_norm:
                heads.append(nn.Sequential(nn.LayerNorm(n_in), nn.Linear(n_in, n_actions)))
            else:
                heads.append(nn.Linear(n_in, n_actions))
        self.q_heads = nn.ModuleList(heads).to(device)

    @torch.no_grad()
    def forward(self, e):
        return torch.stack([h(e) for h in self.q_heads], dim=-1)


class CompositeModel(nn.Module):
    """embedding -> (IQN and/or Q heads).  ``forward`` serves acting only."""

    def __init__(self, embedding_model, distribution_model, q_function_model, device="cpu"):
        super().__init__()
        self.embedding_model = embedding_model
        self.distribution_model = distribution_model
        self.q_function_model = q_function_model
        self.device = device
        self.use_cuda_graph = False
        self.should_build_forward_cuda_graph = False   # written by algorithm_factory.py:24-26
        self.loggables = {}

    @torch.no_grad()
    def forward(self, x, for_action=True):
        x = _as_tensor(x, self.device)
        e = self.embedding_model(x) if self.embedding_model is not None else x
        dist = self.distribution_model(e, for_action=for_action) if self.distribution_model is not None else None
        if self.q_function_model is not None:
            q = self.q_function_model(e)
        elif for_action and dist is not None:
            q = dist.mean(dim=0).unsqueeze(-1)
        else:
            q = None
        return q, dist

    _forward_without_cuda_graph = forward

    def log(self, logger):
        if self.q_function_model is not None:
            logger.log_data(data=float(self.q_function_model.theil), group_name="Debug/Q Ensemble",
                            var_name="Variation Loss")
