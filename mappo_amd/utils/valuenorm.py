"""ValueNorm (onpolicy/utils/valuenorm.py:8-78) with its 3-float state resident in HBM.

`update` runs as HIP kernels (mappo_minibatch_moments + mappo_valuenorm_update); the fused kernels
(GAE, advantages, PPO loss) read the state directly, so `normalize` / `denormalize` are only needed by
outside callers and are plain tensor expressions."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


class ValueNorm(nn.Module):
    def __init__(self, input_shape=1, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5, device="cuda"):
        super().__init__()
        if input_shape != 1 or norm_axes != 1 or per_element_update:
            raise NotImplementedError("only the reference's usage ValueNorm(1) is built")
        self.beta, self.epsilon = beta, epsilon
        # one contiguous 3-float buffer; the reference's three parameters are views of it (state_dict keys kept)
        self._state = torch.zeros(3, dtype=torch.float32, device=device)
        self.running_mean = nn.Parameter(self._state[0:1], requires_grad=False)
        self.running_mean_sq = nn.Parameter(self._state[1:2], requires_grad=False)
        self.debiasing_term = nn.Parameter(self._state[2], requires_grad=False)
        self._mom = torch.zeros(4, dtype=torch.float64, device=device)

    @property
    def state(self):
        """Device tensor [running_mean, running_mean_sq, debiasing_term] handed to the kernels."""
        return self._state

    def running_mean_var(self):
        d = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.running_mean / d
        var = (self.running_mean_sq / d - mean ** 2).clamp(min=1e-2)
        return mean, var

    @torch.no_grad()
    def update(self, input_vector):
        x = self._dev(input_vector).reshape(-1)
        ones = torch.ones_like(x)
        ops.minibatch_moments(x, ones, None, x.numel(), self._mom)
        ops.valuenorm_update(self._state, self._mom, self.beta)

    def _dev(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(device=self._state.device, dtype=torch.float32).contiguous()

    def normalize(self, input_vector):
        mean, var = self.running_mean_var()
        return (self._dev(input_vector) - mean) / torch.sqrt(var)

    def denormalize(self, input_vector):
        """Returns NumPy like the reference (valuenorm.py:76)."""
        mean, var = self.running_mean_var()
        return (self._dev(input_vector) * torch.sqrt(var) + mean).cpu().numpy()
