"""MADDPG networks (reference: core/maddpg/policies.py:21-505, core/common/multi_agent_policies.py:40-636).

Agents are index slices of ONE global observation / action vector (`IndexedBox.indices`); every agent has its
own deterministic actor on its observation slice and its own twin critics on cat(all features, all actions).
All actor parameters live in one flat HBM arena (agent-major), all critic parameters in another: each agent's
optimiser is a `FlatAdam` over its slice, and polyak over ALL agents' parameters is one launch per arena."""
from typing import List, Optional

import numpy as np
import torch as th
from torch import nn

from core.common import distributed as dist_util
from core.common.arena import ArenaSlice, FlatAdam, ParamArena
from core.common.spaces import as_box, get_action_dim
from core.common.torch_layers import create_mlp


def get_multi_agent_actor_critic_arch(net_arch: list) -> tuple:
    """reference: core/common/torch_layers.py:356-373"""
    actor_arch, critic_arch = [], []
    for arch in net_arch:
        if isinstance(arch, list):
            actor_arch.append(arch)
            critic_arch.append(arch)
        else:
            assert isinstance(arch, dict), "Error: the net_arch can only contain be a list of ints or a dict"
            assert "pi" in arch, "Error: no key 'pi' was provided in net_arch for the actor network"
            assert "qf" in arch, "Error: no key 'qf' was provided in net_arch for the critic network"
            actor_arch.append(arch["pi"])
            critic_arch.append(arch["qf"])
    return actor_arch, critic_arch


class _MultiAgentModule(nn.Module):
    def __init__(self, n_agents, observation_space, action_space, observation_space_list, action_space_list):
        super().__init__()
        self.n_agents = n_agents
        self.observation_space, self.action_space = as_box(observation_space), as_box(action_space)
        self.observation_space_list, self.action_space_list = list(observation_space_list), list(action_space_list)
        for sp in self.observation_space_list + self.action_space_list:
            if not hasattr(sp, "indices"):
                raise NotImplementedError("observation_space_list[agent_id] does not have 'indices' attribute. "
                                          "You need to implement a custom extraction method.")  # multi_agent_policies.py:389-393
        self.optimizer_list: list = []

    def _index(self, kind: str, agent_id: int, device) -> th.Tensor:
        """Device-resident index vector of an agent's slice (built once: indexing with a Python list would upload an
        index tensor on every call, which also breaks hipGraph capture)."""
        cache = self.__dict__.setdefault("_idx_cache", {})
        key = (kind, agent_id, str(device))
        if key not in cache:
            sp = (self.observation_space_list if kind == "obs" else self.action_space_list)[agent_id]
            cache[key] = th.as_tensor([int(i) for i in sp.indices], dtype=th.long, device=device)
        return cache[key]

    def _range(self, kind: str, agent_id: int):
        """(start, stop) when the agent's indices are one ascending run -- the slice is then a VIEW (no gather launch)"""
        cache = self.__dict__.setdefault("_range_cache", {})
        key = (kind, agent_id)
        if key not in cache:
            idx = [int(i) for i in (self.observation_space_list if kind == "obs" else self.action_space_list)[agent_id].indices]
            cache[key] = (idx[0], idx[-1] + 1) if idx and idx == list(range(idx[0], idx[0] + len(idx))) else None
        return cache[key]

    def _tiles_in_order(self, kind: str) -> bool:
        """the agents' slices, in agent order, are exactly columns 0..width-1 (so cat(slices) IS the global tensor)"""
        spaces, width = ((self.observation_space_list, self.observation_space.shape[0]) if kind == "obs"
                         else (self.action_space_list, self.action_space.shape[0]))
        flat = [int(i) for sp in spaces for i in sp.indices]
        return flat == list(range(width))

    def _agent_obs_tensor_extract(self, agent_id: int, global_observation: th.Tensor) -> th.Tensor:
        r = self._range("obs", agent_id)
        if r is not None:
            return global_observation[..., r[0]:r[1]]
        return global_observation.index_select(-1, self._index("obs", agent_id, global_observation.device))

    def _agent_action_tensor_extract(self, agent_id: int, global_action: th.Tensor) -> th.Tensor:
        r = self._range("act", agent_id)
        if r is not None:
            return global_action[..., r[0]:r[1]]
        return global_action.index_select(-1, self._index("act", agent_id, global_action.device))

    def set_training_mode(self, mode: bool) -> None:
        self.train(mode)

    @property
    def device(self) -> th.device:
        for p in self.parameters():
            return p.device
        return th.device("cpu")


class Actor(_MultiAgentModule):
    """Per-agent deterministic actors mu_i(o_i) = tanh(MLP_i(o_i)) (reference: maddpg/policies.py:21-128)."""

    def __init__(self, n_agents, observation_space, action_space, observation_space_list, action_space_list, net_arch: List[list],
                 activation_fn=nn.ReLU):
        super().__init__(n_agents, observation_space, action_space, observation_space_list, action_space_list)
        self.net_arch, self.activation_fn = net_arch, activation_fn
        self.mu_list = nn.ModuleList()
        for agent_id, aspace in enumerate(self.action_space_list):
            feat = int(np.prod(self.observation_space_list[agent_id].shape))
            self.mu_list.append(nn.Sequential(*create_mlp(feat, get_action_dim(aspace), net_arch[agent_id], activation_fn,
                                                          squash_output=True)))

    def forward(self, obs: th.Tensor) -> th.Tensor:
        return th.cat([self.mu_list[i](self._agent_obs_tensor_extract(i, obs).float()) for i in range(self.n_agents)], dim=-1)

    def _agent_predict(self, agent_id: int, agent_observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self.mu_list[agent_id](agent_observation.float())


class ContinuousCritic(_MultiAgentModule):
    """Per-agent twin Q networks. `local=False` (MADDPG, reference: maddpg/policies.py:131-272): every agent's critic
    sees the joint input cat(all features, all actions). `local=True` (IDDPG, reference: iddpg/policies.py:22-145): each
    critic sees only its own agent's observation slice and action slice."""

    def __init__(self, n_agents, observation_space, action_space, observation_space_list, action_space_list, net_arch: List[list],
                 activation_fn=nn.ReLU, n_critics: int = 2, local: bool = False):
        super().__init__(n_agents, observation_space, action_space, observation_space_list, action_space_list)
        self.n_critics, self.net_arch, self.local = n_critics, net_arch, local
        feat = [int(np.prod(sp.shape)) for sp in self.observation_space_list]
        act = [get_action_dim(sp) for sp in self.action_space_list]
        self.q_networks_list: List[List[nn.Module]] = []
        for agent_id in range(n_agents):
            width = feat[agent_id] + act[agent_id] if local else sum(feat) + sum(act)
            nets = []
            for idx in range(n_critics):
                q_net = nn.Sequential(*create_mlp(width, 1, net_arch[agent_id], activation_fn))
                self.add_module(f"agent{agent_id}_qf{idx}", q_net)
                nets.append(q_net)
            self.q_networks_list.append(nets)

    def _input(self, agent_id: int, obs: th.Tensor, actions: th.Tensor) -> th.Tensor:
        if self.local:
            return th.cat((self._agent_obs_tensor_extract(agent_id, obs).float(), self._agent_action_tensor_extract(agent_id, actions)), dim=-1)
        if self._tiles_in_order("obs"):  # cat(all agents' features) is the observation itself
            return th.cat([obs.float(), actions], dim=1)
        feats = [self._agent_obs_tensor_extract(i, obs).float() for i in range(self.n_agents)]
        return th.cat([th.cat(feats, dim=-1), actions], dim=1)

    def agent_forward(self, agent_id: int, obs: th.Tensor, actions: th.Tensor, only_first: bool = False) -> tuple:
        """Only agent `agent_id`'s Q networks: what `forward(obs, actions)[agent_id]` / `q1_forward(...)[agent_id]` return,
        without evaluating the other agents' networks (the reference computes all of them and discards n_agents - 1)."""
        x = self._input(agent_id, obs, actions)
        nets = self.q_networks_list[agent_id]
        return tuple(q(x) for q in (nets[:1] if only_first else nets))

    def forward(self, obs: th.Tensor, actions: th.Tensor) -> list:
        shared = None if self.local else self._input(0, obs, actions)
        return [tuple(q(shared if shared is not None else self._input(i, obs, actions)) for q in self.q_networks_list[i])
                for i in range(self.n_agents)]

    def q1_forward(self, obs: th.Tensor, actions: th.Tensor) -> list:
        shared = None if self.local else self._input(0, obs, actions)
        return [self.q_networks_list[i][0](shared if shared is not None else self._input(i, obs, actions)) for i in range(self.n_agents)]


class MADDPGPolicy(nn.Module):
    """reference: maddpg/policies.py:275-505. Positional order (n_agents, observation_space, action_space,
    observation_space_list, action_space_list, lr_schedule_list) is the reference's."""

    def __init__(self, n_agents: int, observation_space, action_space, observation_space_list, action_space_list,
                 lr_schedule_list, net_arch: Optional[list] = None, activation_fn=nn.ReLU, features_extractor_class_list=None,
                 features_extractor_kwargs_list=None, normalize_images: bool = True, optimizer_class_list=None,
                 optimizer_kwargs_list=None, n_critics: int = 2, share_features_extractor: bool = False):
        super().__init__()
        local_critics = getattr(type(self), "local_critics", False)
        if features_extractor_class_list is not None or optimizer_class_list is not None:
            raise NotImplementedError("custom feature extractors / optimiser classes are out of scope (SURVEY 2)")
        if share_features_extractor:
            raise NotImplementedError("share_features_extractor=True is not built (FlattenExtractor has no parameters)")
        self.n_agents = n_agents
        self.observation_space, self.action_space = as_box(observation_space), as_box(action_space)
        self.observation_space_list, self.action_space_list = list(observation_space_list), list(action_space_list)
        if net_arch is None:
            net_arch = [[400, 300] for _ in range(n_agents)]  # maddpg/policies.py:349-358
        self.net_arch = net_arch
        self.actor_arch, self.critic_arch = get_multi_agent_actor_critic_arch(net_arch)
        self.activation_fn, self.n_critics = activation_fn, n_critics
        self.squash_output = True
        self._lr_schedule_list = lr_schedule_list
        args = (n_agents, self.observation_space, self.action_space, self.observation_space_list, self.action_space_list)
        # creation order of the reference (_build, :386-430): actor, actor_target, critic, critic_target
        self.actor = Actor(*args, self.actor_arch, activation_fn)
        self.actor_target = Actor(*args, self.actor_arch, activation_fn)
        self.actor_target.load_state_dict(self.actor.state_dict())
        self.critic = ContinuousCritic(*args, self.critic_arch, activation_fn, n_critics, local_critics)
        self.critic_target = ContinuousCritic(*args, self.critic_arch, activation_fn, n_critics, local_critics)
        self.critic_target.load_state_dict(self.critic.state_dict())
        self.actor_target.set_training_mode(False)
        self.critic_target.set_training_mode(False)

    def to_device_arenas(self, device) -> None:
        from core.common import fused

        def groups(critic):  # per agent: its twin Q networks' layers back to back (batched GEMMs)
            return [g for nets in critic.q_networks_list for g in fused.twin_groups(nets)]

        self.actor_arena = ParamArena(self.actor.parameters(), device)
        self.critic_arena = ParamArena(self.critic.parameters(), device, groups=groups(self.critic))
        self.actor_target_arena = ParamArena(self.actor_target.parameters(), device, with_grad=False)
        self.critic_target_arena = ParamArena(self.critic_target.parameters(), device, with_grad=False, groups=groups(self.critic_target))
        self.critic_stacks, self.critic_target_stacks, g0 = [], [], 0
        for nets in self.critic.q_networks_list:
            n_groups = len(fused.twin_groups(nets))
            self.critic_stacks.append(fused.twin_stack(self.critic_arena, g0, n_groups // 2) if n_groups else None)
            self.critic_target_stacks.append(fused.twin_stack(self.critic_target_arena, g0, n_groups // 2) if n_groups else None)
            g0 += n_groups
        for p in list(self.actor_target.parameters()) + list(self.critic_target.parameters()):
            p.requires_grad_(False)
        self.actor_slices, self.critic_slices = [], []
        self.actor.optimizer_list, self.critic.optimizer_list = [], []
        for i in range(self.n_agents):
            lr = self._lr_schedule_list[i](1)
            sl = ArenaSlice(self.actor_arena, self.actor.mu_list[i].parameters())
            self.actor_slices.append(sl)
            self.actor.optimizer_list.append(FlatAdam(sl, lr=lr))
            cparams = [p for q in self.critic.q_networks_list[i] for p in q.parameters()]
            sl = ArenaSlice(self.critic_arena, cparams)
            self.critic_slices.append(sl)
            self.critic.optimizer_list.append(FlatAdam(sl, lr=lr))

    def flat_optimizers(self) -> list:
        return list(self.actor.optimizer_list) + list(self.critic.optimizer_list)

    def broadcast_from_rank0(self) -> None:
        for arena in (self.actor_arena, self.critic_arena, self.actor_target_arena, self.critic_target_arena):
            dist_util.broadcast_(arena.flat, 0)

    def set_training_mode(self, mode: bool) -> None:
        self.actor.set_training_mode(mode)
        self.critic.set_training_mode(mode)
        self.training = mode

    # ---- prediction (reference: multi_agent_policies.py:481-617) ----------------------------------------------------
    def _predict(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self.actor(observation)

    def forward(self, observation: th.Tensor, deterministic: bool = False) -> th.Tensor:
        return self._predict(observation, deterministic)

    def agent_scale_action(self, agent_id: int, action: np.ndarray) -> np.ndarray:
        low, high = self.action_space_list[agent_id].low, self.action_space_list[agent_id].high
        return 2.0 * ((action - low) / (high - low)) - 1.0

    def agent_unscale_action(self, agent_id: int, scaled_action: np.ndarray) -> np.ndarray:
        low, high = self.action_space_list[agent_id].low, self.action_space_list[agent_id].high
        return low + (0.5 * (scaled_action + 1.0) * (high - low))

    def scale_action(self, action: np.ndarray) -> np.ndarray:
        low, high = self.action_space.low, self.action_space.high
        return 2.0 * ((action - low) / (high - low)) - 1.0

    def unscale_action(self, scaled_action: np.ndarray) -> np.ndarray:
        low, high = self.action_space.low, self.action_space.high
        return low + (0.5 * (scaled_action + 1.0) * (high - low))

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        self.set_training_mode(False)
        dev = self.actor.device
        obs = observation.to(dev, th.float32) if isinstance(observation, th.Tensor) else th.as_tensor(np.asarray(observation, np.float32), device=dev)
        vectorized = obs.dim() == 2
        obs = obs.reshape(-1, *self.observation_space.shape)
        out = []
        for i in range(self.n_agents):
            with th.no_grad():
                a = self.actor._agent_predict(i, self.actor._agent_obs_tensor_extract(i, obs))
            a = a.cpu().numpy().reshape((-1, *self.action_space_list[i].shape))
            a = self.agent_unscale_action(i, a)  # squash_output is always True for MADDPG
            out.append(a if vectorized else a.squeeze(axis=0))
        return np.concatenate(out, axis=-1), state


MlpPolicy = MADDPGPolicy
