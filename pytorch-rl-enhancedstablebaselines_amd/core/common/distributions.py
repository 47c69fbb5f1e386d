"""Squashed diagonal Gaussian (reference: core/common/distributions.py:125-260).

`eps_queue`: teacher-forcing hook. The reference draws eps through `Normal.rsample` on the CPU generator; a
GPU run cannot reproduce those numbers, so parity tests push the recorded eps tensors here and the next
`sample()` calls consume them instead of `randn` (SURVEY 7 "Stochastic parity")."""
import math
from typing import List, Optional

import torch as th


class SquashedDiagGaussianDistribution:
    def __init__(self, action_dim: int, epsilon: float = 1e-6):
        self.action_dim = action_dim
        self.epsilon = epsilon
        self.mean: Optional[th.Tensor] = None
        self.log_std: Optional[th.Tensor] = None
        self.gaussian_actions: Optional[th.Tensor] = None
        self.eps_queue: List[th.Tensor] = []

    def proba_distribution(self, mean_actions: th.Tensor, log_std: th.Tensor):
        self.mean, self.log_std = mean_actions, log_std  # Normal(mean, log_std.exp()) (:161-165)
        return self

    def draw_eps(self, shape, device, dtype=th.float32) -> th.Tensor:
        """Standard-normal draw of Normal.rsample (distributions.py:183), or the next teacher-forced tensor."""
        if self.eps_queue:
            return self.eps_queue.pop(0).to(device, dtype).reshape(shape)
        return th.randn(shape, dtype=dtype, device=device)

    def _eps(self) -> th.Tensor:
        return self.draw_eps(self.mean.shape, self.mean.device, self.mean.dtype)

    def sample(self) -> th.Tensor:
        """rsample then tanh (:183, :236-239)"""
        self.gaussian_actions = self.mean + self.log_std.exp() * self._eps()
        return th.tanh(self.gaussian_actions)

    def mode(self) -> th.Tensor:
        self.gaussian_actions = self.mean
        return th.tanh(self.gaussian_actions)

    def log_prob(self, actions: th.Tensor, gaussian_actions: Optional[th.Tensor] = None) -> th.Tensor:
        """sum Normal.log_prob(u) - sum log(1 - a^2 + eps) (:170-172, :226-234)"""
        if gaussian_actions is None:
            a = actions.clamp(-1.0 + 1e-6, 1.0 - 1e-6)  # TanhBijector.inverse (:699-712)
            gaussian_actions = 0.5 * (a.log1p() - (-a).log1p())
        std = self.log_std.exp()
        var = std ** 2
        lp = -((gaussian_actions - self.mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))  # torch Normal.log_prob
        lp = lp.sum(dim=1) if lp.dim() > 1 else lp.sum()
        lp = lp - th.sum(th.log(1 - actions ** 2 + self.epsilon), dim=1)
        return lp

    def actions_from_params(self, mean_actions, log_std, deterministic: bool = False) -> th.Tensor:
        self.proba_distribution(mean_actions, log_std)
        return self.mode() if deterministic else self.sample()

    def log_prob_from_params(self, mean_actions, log_std):
        action = self.actions_from_params(mean_actions, log_std)
        return action, self.log_prob(action, self.gaussian_actions)
