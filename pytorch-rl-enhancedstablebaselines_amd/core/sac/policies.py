"""SAC actor and policy (reference: core/sac/policies.py:25-178, :181-360). `MlpPolicy` only: CNN / dict
policies and gSDE are out of scope. Construction order = the reference's, so seeded initial
weights coincide; afterwards every optimiser group is moved into a flat HBM arena (core/common/arena.py)."""
from typing import Optional, Union

import torch as th
from torch import nn

from core.common import distributed as dist_util
from core.common.arena import FlatAdam, ParamArena, make_optimizer
from core.common.distributions import SquashedDiagGaussianDistribution
from core.common.policies import BasePolicy, ContinuousCritic
from core.common.spaces import get_action_dim
from core.common.torch_layers import FlattenExtractor, create_mlp, get_actor_critic_arch

LOG_STD_MAX = 2    # reference: sac/policies.py:20-22
LOG_STD_MIN = -20


class Actor(BasePolicy):
    """Gaussian actor with tanh squashing (reference: sac/policies.py:25-178)."""

    def __init__(self, observation_space, action_space, net_arch: list, features_extractor: nn.Module, features_dim: int,
                 activation_fn=nn.ReLU, normalize_images: bool = True):
        super().__init__(observation_space, action_space, features_extractor=features_extractor,
                         normalize_images=normalize_images, squash_output=True)
        self.net_arch, self.features_dim, self.activation_fn = net_arch, features_dim, activation_fn
        action_dim = get_action_dim(self.action_space)
        self.latent_pi = nn.Sequential(*create_mlp(features_dim, -1, net_arch, activation_fn))
        last_layer_dim = net_arch[-1] if len(net_arch) > 0 else features_dim
        self.action_dist = SquashedDiagGaussianDistribution(action_dim)
        self.mu = nn.Linear(last_layer_dim, action_dim)
        self.log_std = nn.Linear(last_layer_dim, action_dim)

    def get_action_dist_params(self, obs: th.Tensor):
        latent_pi = self.latent_pi(self.extract_features(obs, self.features_extractor))
        return self.mu(latent_pi), th.clamp(self.log_std(latent_pi), LOG_STD_MIN, LOG_STD_MAX), {}

    def forward(self, obs: th.Tensor, deterministic: bool = False) -> th.Tensor:
        mean_actions, log_std, _ = self.get_action_dist_params(obs)
        return self.action_dist.actions_from_params(mean_actions, log_std, deterministic=deterministic)

    def action_log_prob(self, obs: th.Tensor):
        mean_actions, log_std, _ = self.get_action_dist_params(obs)
        return self.action_dist.log_prob_from_params(mean_actions, log_std)

    def _predict(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self(observation, deterministic)


class SACPolicy(BasePolicy):
    """reference: sac/policies.py:181-360"""

    def __init__(self, observation_space, action_space, lr_schedule, net_arch: Optional[Union[list, dict]] = None,
                 activation_fn=nn.ReLU, use_sde: bool = False, log_std_init: float = -3, use_expln: bool = False,
                 clip_mean: float = 2.0, features_extractor_class=FlattenExtractor, features_extractor_kwargs=None,
                 normalize_images: bool = True, optimizer_class=th.optim.Adam, optimizer_kwargs: Optional[dict] = None,
                 n_critics: int = 2, share_features_extractor: bool = False):
        super().__init__(observation_space, action_space, features_extractor_class, features_extractor_kwargs,
                         optimizer_class=optimizer_class, optimizer_kwargs=optimizer_kwargs, squash_output=True,
                         normalize_images=normalize_images)
        if use_sde:
            raise NotImplementedError("gSDE is out of scope for the CSTR path (SURVEY 2)")
        if share_features_extractor:
            raise NotImplementedError("share_features_extractor=True is not built (FlattenExtractor has no parameters)")
        if net_arch is None:
            net_arch = [256, 256]
        actor_arch, critic_arch = get_actor_critic_arch(net_arch)
        self.net_arch, self.activation_fn = net_arch, activation_fn
        self.actor_arch, self.critic_arch, self.n_critics = actor_arch, critic_arch, n_critics
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
        """Module creation order of the reference (sac/policies.py:280-310): actor, critic, critic_target."""
        self.actor = self.make_actor()
        self.critic = self.make_critic()
        self.critic_target = self.make_critic()
        self.critic_target.load_state_dict(self.critic.state_dict())
        self.critic_target.set_training_mode(False)
        self.actor.optimizer = self.critic.optimizer = None  # created by to_device_arenas()

    def to_device_arenas(self, device) -> None:
        """`.to(device)` of the reference (off_policy_algorithm.py:209), into flat arenas + one-launch optimisers."""
        from core.common import fused

        lr = self._lr_schedule(1)
        a = self.actor
        head_groups = [[a.mu.weight, a.log_std.weight], [a.mu.bias, a.log_std.bias]]  # one GEMM for both heads
        self.actor_arena, self.actor.optimizer = make_optimizer(self.actor.parameters(), device, lr, self.optimizer_class,
                                                                self.optimizer_kwargs, groups=head_groups)
        (hw, hwg), (hb, hbg) = self.actor_arena.stacked(0), self.actor_arena.stacked(1)
        self.actor_head = (hw, hwg, hb, hbg)
        self.critic_arena, self.critic.optimizer = make_optimizer(self.critic.parameters(), device, lr, self.optimizer_class,
                                                                  self.optimizer_kwargs, groups=fused.twin_groups(self.critic.q_networks),
                                                                  extra_grad=64)  # tail: the entropy coefficient's gradient
        self.critic_target_arena = ParamArena(self.critic_target.parameters(), device, with_grad=False,
                                              groups=fused.twin_groups(self.critic_target.q_networks))
        self.critic_stack = fused.twin_stack(self.critic_arena)
        self.critic_target_stack = fused.twin_stack(self.critic_target_arena)
        for p in self.critic_target.parameters():
            p.requires_grad_(False)

    def flat_optimizers(self) -> list:
        return [o for o in (self.actor.optimizer, self.critic.optimizer) if isinstance(o, FlatAdam)]

    def broadcast_from_rank0(self) -> None:
        for arena in (self.actor_arena, self.critic_arena, self.critic_target_arena):
            dist_util.broadcast_(arena.flat, 0)

    def forward(self, obs: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self._predict(obs, deterministic=deterministic)

    def _predict(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self.actor(observation, deterministic)

    def set_training_mode(self, mode: bool) -> None:
        self.actor.set_training_mode(mode)
        self.critic.set_training_mode(mode)
        self.training = mode


MlpPolicy = SACPolicy
