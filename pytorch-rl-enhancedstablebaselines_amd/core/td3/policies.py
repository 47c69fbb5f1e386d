"""TD3 deterministic actor and policy (reference: core/td3/policies.py:20-83, :86-260). `MlpPolicy` only."""
from typing import Optional, Union

import torch as th
from torch import nn

from core.common import distributed as dist_util
from core.common.arena import FlatAdam, ParamArena, make_optimizer
from core.common.policies import BasePolicy, ContinuousCritic
from core.common.spaces import get_action_dim
from core.common.torch_layers import FlattenExtractor, create_mlp, get_actor_critic_arch


class Actor(BasePolicy):
    """mu(s) = tanh(MLP(s)) (reference: td3/policies.py:20-83)"""

    def __init__(self, observation_space, action_space, net_arch: list, features_extractor: nn.Module, features_dim: int,
                 activation_fn=nn.ReLU, normalize_images: bool = True):
        super().__init__(observation_space, action_space, features_extractor=features_extractor,
                         normalize_images=normalize_images, squash_output=True)
        self.net_arch, self.features_dim, self.activation_fn = net_arch, features_dim, activation_fn
        self.mu = nn.Sequential(*create_mlp(features_dim, get_action_dim(self.action_space), net_arch, activation_fn,
                                            squash_output=True))

    def forward(self, obs: th.Tensor) -> th.Tensor:
        return self.mu(self.extract_features(obs, self.features_extractor))

    def _predict(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self(observation)  # always deterministic (reference :80-83)


class TD3Policy(BasePolicy):
    """reference: td3/policies.py:86-260; default net_arch [400, 300] (:141-145)"""

    def __init__(self, observation_space, action_space, lr_schedule, net_arch: Optional[Union[list, dict]] = None,
                 activation_fn=nn.ReLU, features_extractor_class=FlattenExtractor, features_extractor_kwargs=None,
                 normalize_images: bool = True, optimizer_class=th.optim.Adam, optimizer_kwargs: Optional[dict] = None,
                 n_critics: int = 2, share_features_extractor: bool = False):
        super().__init__(observation_space, action_space, features_extractor_class, features_extractor_kwargs,
                         optimizer_class=optimizer_class, optimizer_kwargs=optimizer_kwargs, squash_output=True,
                         normalize_images=normalize_images)
        if share_features_extractor:
            raise NotImplementedError("share_features_extractor=True is not built (FlattenExtractor has no parameters)")
        if net_arch is None:
            net_arch = [400, 300]
        self.actor_arch, self.critic_arch = get_actor_critic_arch(net_arch)
        self.net_arch, self.activation_fn, self.n_critics = net_arch, activation_fn, n_critics
        self.share_features_extractor = share_features_extractor
        self._lr_schedule = lr_schedule
        self._build(lr_schedule)

    def make_actor(self) -> Actor:
        fe = self.make_features_extractor()
        return Actor(self.observation_space, self.action_space, self.actor_arch, fe, fe.features_dim, self.activation_fn)

    def make_critic(self) -> ContinuousCritic:
        fe = self.make_features_extractor()
        return ContinuousCritic(self.observation_space, self.action_space, self.critic_arch, fe, fe.features_dim,
                                self.activation_fn, n_critics=self.n_critics, share_features_extractor=False)

    def _build(self, lr_schedule) -> None:
        """Creation order of the reference (td3/policies.py:172-208): actor, actor_target, critic, critic_target."""
        self.actor = self.make_actor()
        self.actor_target = self.make_actor()
        self.actor_target.load_state_dict(self.actor.state_dict())
        self.critic = self.make_critic()
        self.critic_target = self.make_critic()
        self.critic_target.load_state_dict(self.critic.state_dict())
        self.actor_target.set_training_mode(False)
        self.critic_target.set_training_mode(False)
        self.actor.optimizer = self.critic.optimizer = None

    def to_device_arenas(self, device) -> None:
        from core.common import fused

        lr = self._lr_schedule(1)
        self.actor_arena, self.actor.optimizer = make_optimizer(self.actor.parameters(), device, lr, self.optimizer_class,
                                                                self.optimizer_kwargs)
        self.critic_arena, self.critic.optimizer = make_optimizer(self.critic.parameters(), device, lr, self.optimizer_class,
                                                                  self.optimizer_kwargs, groups=fused.twin_groups(self.critic.q_networks))
        self.actor_target_arena = ParamArena(self.actor_target.parameters(), device, with_grad=False)
        self.critic_target_arena = ParamArena(self.critic_target.parameters(), device, with_grad=False,
                                              groups=fused.twin_groups(self.critic_target.q_networks))
        self.critic_stack = fused.twin_stack(self.critic_arena)
        self.critic_target_stack = fused.twin_stack(self.critic_target_arena)
        for p in list(self.actor_target.parameters()) + list(self.critic_target.parameters()):
            p.requires_grad_(False)

    def flat_optimizers(self) -> list:
        return [o for o in (self.actor.optimizer, self.critic.optimizer) if isinstance(o, FlatAdam)]

    def broadcast_from_rank0(self) -> None:
        for arena in (self.actor_arena, self.critic_arena, self.actor_target_arena, self.critic_target_arena):
            dist_util.broadcast_(arena.flat, 0)

    def forward(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self._predict(observation, deterministic=deterministic)

    def _predict(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self.actor(observation)

    def set_training_mode(self, mode: bool) -> None:
        self.actor.set_training_mode(mode)
        self.critic.set_training_mode(mode)
        self.training = mode


MlpPolicy = TD3Policy
