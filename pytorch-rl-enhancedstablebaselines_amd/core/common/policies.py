"""Policy base classes and the continuous critic (reference: core/common/policies.py:39-414, :912-987)."""
from typing import Optional

import numpy as np
import torch as th
from torch import nn

from core.common.spaces import as_box, get_action_dim
from core.common.torch_layers import FlattenExtractor, create_mlp


class BaseModel(nn.Module):
    """reference: core/common/policies.py:39-277"""

    optimizer = None

    def __init__(self, observation_space, action_space, features_extractor_class=FlattenExtractor,
                 features_extractor_kwargs: Optional[dict] = None, features_extractor: Optional[nn.Module] = None,
                 normalize_images: bool = True, optimizer_class=th.optim.Adam, optimizer_kwargs: Optional[dict] = None):
        super().__init__()
        self.observation_space = as_box(observation_space)
        self.action_space = as_box(action_space)
        self.features_extractor = features_extractor
        self.normalize_images = normalize_images
        self.optimizer_class = optimizer_class
        self.optimizer_kwargs = optimizer_kwargs or {}
        self.features_extractor_class = features_extractor_class
        self.features_extractor_kwargs = features_extractor_kwargs or {}

    def make_features_extractor(self) -> nn.Module:
        if self.features_extractor_class is not FlattenExtractor:
            raise NotImplementedError("Only FlattenExtractor is built (CNN / dict extractors are out of scope, SURVEY 2)")
        return FlattenExtractor(int(np.prod(self.observation_space.shape)))

    def extract_features(self, obs: th.Tensor, features_extractor: nn.Module) -> th.Tensor:
        return features_extractor(obs.float())  # preprocess_obs, Box branch (preprocessing.py:118-121)

    @property
    def device(self) -> th.device:
        for p in self.parameters():
            return p.device
        return th.device("cpu")

    def set_training_mode(self, mode: bool) -> None:
        self.train(mode)

    def obs_to_tensor(self, observation) -> tuple:
        """reference: policies.py:240-277 (Box branch)"""
        if isinstance(observation, th.Tensor):
            obs = observation.to(self.device, th.float32)
        else:
            obs = th.as_tensor(np.asarray(observation, dtype=np.float32), device=self.device)
        vectorized = obs.dim() == len(self.observation_space.shape) + 1
        if not vectorized:
            obs = obs.reshape((-1, *self.observation_space.shape))
        return obs, vectorized


class BasePolicy(BaseModel):
    """reference: core/common/policies.py:280-414"""

    def __init__(self, *args, squash_output: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self._squash_output = squash_output

    @property
    def squash_output(self) -> bool:
        return self._squash_output

    @staticmethod
    def _dummy_schedule(progress_remaining: float) -> float:
        return 0.0

    def _predict(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        raise NotImplementedError

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """reference: policies.py:331-386 -> (np.ndarray, None)"""
        self.set_training_mode(False)
        obs_tensor, vectorized = self.obs_to_tensor(observation)
        with th.no_grad():
            actions = self._predict(obs_tensor, deterministic=deterministic)
        actions = actions.cpu().numpy().reshape((-1, *self.action_space.shape))
        if self.squash_output:
            actions = self.unscale_action(actions)
        else:
            actions = np.clip(actions, self.action_space.low, self.action_space.high)
        if not vectorized:
            actions = actions.squeeze(axis=0)
        return actions, state

    def scale_action(self, action: np.ndarray) -> np.ndarray:
        low, high = self.action_space.low, self.action_space.high
        return 2.0 * ((action - low) / (high - low)) - 1.0

    def unscale_action(self, scaled_action: np.ndarray) -> np.ndarray:
        low, high = self.action_space.low, self.action_space.high
        return low + (0.5 * (scaled_action + 1.0) * (high - low))


class ContinuousCritic(BaseModel):
    """reference: core/common/policies.py:912-987 -- n_critics independent Q networks on cat(obs, action)."""

    def __init__(self, observation_space, action_space, net_arch: list, features_extractor: nn.Module, features_dim: int,
                 activation_fn=nn.ReLU, normalize_images: bool = True, n_critics: int = 2,
                 share_features_extractor: bool = True):
        super().__init__(observation_space, action_space, features_extractor=features_extractor,
                         normalize_images=normalize_images)
        action_dim = get_action_dim(self.action_space)
        self.share_features_extractor = share_features_extractor
        self.n_critics = n_critics
        self.q_networks: list = []
        for idx in range(n_critics):
            q_net = nn.Sequential(*create_mlp(features_dim + action_dim, 1, net_arch, activation_fn))
            self.add_module(f"qf{idx}", q_net)
            self.q_networks.append(q_net)

    def forward(self, obs: th.Tensor, actions: th.Tensor) -> tuple:
        with th.set_grad_enabled(not self.share_features_extractor):
            features = self.extract_features(obs, self.features_extractor)
        qvalue_input = th.cat([features, actions], dim=1)
        return tuple(q_net(qvalue_input) for q_net in self.q_networks)

    def q1_forward(self, obs: th.Tensor, actions: th.Tensor) -> th.Tensor:
        with th.no_grad():
            features = self.extract_features(obs, self.features_extractor)
        return self.q_networks[0](th.cat([features, actions], dim=1))
