"""MADDPG with the reference's REAL constructor signature -- `n_agents` is the first positional and
`learning_rate_list` must be a list of length n_agents (reference: core/maddpg/maddpg.py:34-63,
core/common/base_class.py:931-934) -- and the reference's train() arithmetic (:117-191), quirks included:

  Q1  `_sample_action` applies neither action scaling nor action noise: buffer_action = action = predict()
      (core/common/multiagent_policy_algorithm.py:369, :391-392);
  Q2  the actor update evaluates EVERY agent's actor on agent `agent_id`'s observation slice (:169-171), so all
      agents' observation slices must have the same width;
  Q3  polyak runs INSIDE the per-agent loop: n_agents passes over all agents' parameters per delayed step (:181-185);
  Q4  `_update_learning_rate` is called per agent with [actor_opt_i, critic_opt_i] and indexes that 2-list by
      schedule number: actor optimisers follow schedule 0, critic optimisers schedule 1 (base_class.py:1120-1134).

Each quirk is reproduced (the golden vectors come from the unmodified reference) and can be switched off with
`faithful_quirks=False`, which gives the textbook behaviour.
"""
import os
from typing import List, Optional, Union

import torch as th
from torch.nn import functional as F

from core.common import blas, fused, hip_ops
from core.common.buffers import ReplayBuffer
from core.common.logger import DeviceMean
from core.common.off_policy_algorithm import OffPolicyAlgorithm
from core.common.spaces import split_spaces
from core.common.utils import get_schedule_fn, update_learning_rate
from core.common.vec_env import CSTRVecEnv
from core.maddpg.policies import MlpPolicy


# steps without a policy update: the agents' independent critic steps share launches (fused.twin_pair_forward_many, one deferred
# weight-gradient pass, one Adam launch); "0" = one agent after the other (A/B knob, bit-identical: tests/test_learner_parity.py)
BATCH_AGENT_CRITIC_STEPS = os.environ.get("CSTR_MADDPG_BATCH_AGENTS", "1") != "0"


class MADDPG(OffPolicyAlgorithm):
    policy_aliases = {"MlpPolicy": MlpPolicy}

    def __init__(self, n_agents: int, policy, env, observation_splits: List[List[int]], action_splits: List[List[int]],
                 learning_rate_list=1e-3, buffer_size: int = 1_000_000, learning_starts: int = 100, batch_size: int = 256,
                 tau: float = 0.005, gamma: float = 0.99, train_freq: Union[int, tuple] = 1, gradient_steps: int = 1,
                 action_noise=None, replay_buffer_class=None, replay_buffer_kwargs: Optional[dict] = None,
                 optimize_memory_usage: bool = False, policy_delay: int = 2, target_policy_noise: float = 0.2,
                 target_noise_clip: float = 0.5, stats_window_size: int = 100, tensorboard_log: Optional[str] = None,
                 policy_kwargs: Optional[dict] = None, verbose: int = 0, seed: Optional[int] = None, device="auto",
                 _init_setup_model: bool = True, faithful_quirks: bool = True):
        self.n_agents = n_agents
        if not hasattr(learning_rate_list, "__len__"):
            raise TypeError(f"object of type '{type(learning_rate_list).__name__}' has no len()")  # the declared default 1e-3 fails
        if n_agents != len(learning_rate_list):
            raise ValueError(f"In a multi-agent scenario, the number of [n_agents, len(learning_rates)]  must be consistent, "
                             f"now they are [{n_agents}, {len(learning_rate_list)}], respectively.")
        self.learning_rate_list = learning_rate_list
        super().__init__(policy, env, learning_rate_list[0], buffer_size, learning_starts, batch_size, tau, gamma, train_freq,
                         gradient_steps, action_noise=action_noise, replay_buffer_class=replay_buffer_class,
                         replay_buffer_kwargs=replay_buffer_kwargs, policy_kwargs=policy_kwargs,
                         stats_window_size=stats_window_size, tensorboard_log=tensorboard_log, verbose=verbose, device=device,
                         seed=seed, sde_support=False, optimize_memory_usage=optimize_memory_usage,
                         supported_action_spaces=(object,), support_multi_env=True)
        self.observation_splits, self.action_splits = observation_splits, action_splits
        self.observation_space_list, self.action_space_list = split_spaces(self.observation_space, self.action_space,
                                                                           observation_splits, action_splits)
        if len(self.observation_space_list) != n_agents or len(self.action_space_list) != n_agents:
            raise ValueError("observation_splits / action_splits must have one entry per agent")
        self.policy_delay, self.target_noise_clip, self.target_policy_noise = policy_delay, target_noise_clip, target_policy_noise
        self.faithful_quirks = faithful_quirks
        self.debug_capture = False
        self.last_train_tensors: dict = {}
        self.noise_queue: List[th.Tensor] = []  # teacher forcing: one tensor per agent per gradient step (maddpg.py:137)
        if _init_setup_model:
            self._setup_model()

    # ---- setup ----------------------------------------------------------------------------------------------------
    def _setup_lr_schedule(self) -> None:
        self.lr_schedule_list = [get_schedule_fn(lr) for lr in self.learning_rate_list]
        self.lr_schedule = self.lr_schedule_list[0]

    def _setup_model(self) -> None:
        """reference: multiagent_policy_algorithm.py:172-212"""
        self._setup_lr_schedule()
        blas.configure()
        if self.world_size > 1 and isinstance(self.env, CSTRVecEnv):
            self.env.seed_offset = self.rank * self.n_envs
        self.set_random_seed(self.seed)
        if self.replay_buffer_class is None:
            self.replay_buffer_class = ReplayBuffer
        if self.replay_buffer is None:
            self.replay_buffer = self.replay_buffer_class(self.buffer_size, self.observation_space, self.action_space,
                                                          device=self.device, n_envs=self.n_envs,
                                                          optimize_memory_usage=self.optimize_memory_usage, **self.replay_buffer_kwargs)
        self.policy = self.policy_class(self.n_agents, self.observation_space, self.action_space, self.observation_space_list,
                                        self.action_space_list, self.lr_schedule_list, **self.policy_kwargs)
        self.policy.to_device_arenas(self.device)
        if self.world_size > 1:
            self.policy.broadcast_from_rank0()
            for opt in self.policy.flat_optimizers():
                opt.grad_scale = 1.0 / self.world_size
            if self.seed is not None:
                th.manual_seed(self.seed + 1000003 * self.rank)
        self._convert_train_freq()
        self._ep_return = th.zeros(self.n_envs, dtype=th.float32, device=self.device)
        self._ep_stats = th.zeros(4, dtype=th.float64, device=self.device)
        self.actor, self.actor_target = self.policy.actor, self.policy.actor_target
        self.critic, self.critic_target = self.policy.critic, self.policy.critic_target
        z = lambda: th.zeros(1, dtype=th.float32, device=self.device)  # noqa: E731
        names = [f"{k}{i}" for k in ("actor", "critic") for i in range(self.n_agents)]
        self._loss_sum_buf = th.zeros(len(names), dtype=th.float32, device=self.device)  # one fill per train()
        self._loss_sums = {nm: self._loss_sum_buf[j:j + 1] for j, nm in enumerate(names)}
        self._static_batch, self._packed = None, None
        self._loss_now = z()
        pol = self.policy
        self.fused_learner = (all(fused.FastMLP.supported(m) for m in self.actor.mu_list)
                              and all(fused.FastMLP.supported(q) for nets in self.critic.q_networks_list for q in nets))
        self._actor_group = self._actor_target_group = None
        if self.fused_learner:
            self._fast_actors = [fused.FastMLP(m) for m in self.actor.mu_list]
            self._fast_actor_targets = [fused.FastMLP(m) for m in self.actor_target.mu_list]
            # all agents' actors layer by layer in ONE launch per layer (identical architectures, contiguous column slices)
            ranges_ok = all(self.actor._range(k, i) is not None for k in ("obs", "act") for i in range(self.n_agents))
            grouped = ranges_ok and self.actor._tiles_in_order("act") and fused.FastActorGroup.supported(self._fast_actors)
            self._actor_group = fused.FastActorGroup(self._fast_actors) if grouped else None
            self._actor_target_group = fused.FastActorGroup(self._fast_actor_targets) if grouped else None

            class _Nets:  # FastTwinCritic reads `.q_networks`
                def __init__(self, nets):
                    self.q_networks = nets

            self._fast_critics = [fused.FastTwinCritic(_Nets(n), st) for n, st in zip(self.critic.q_networks_list, pol.critic_stacks)]
            self._fast_critic_targets = [fused.FastTwinCritic(_Nets(n), st)
                                         for n, st in zip(self.critic_target.q_networks_list, pol.critic_target_stacks)]
        if self.faithful_quirks:
            widths = {len(s) for s in self.observation_splits}
            if len(widths) != 1:
                raise ValueError("MADDPG (reference behaviour Q2) needs equally wide observation slices for all agents; "
                                 "pass faithful_quirks=False for per-agent slices")

    # ---- acting ---------------------------------------------------------------------------------------------------
    def _policy_out_device(self, obs: th.Tensor) -> th.Tensor:
        """Joint action for the fused collect kernel: every agent's actor on its own observation slice (policies.py:87)."""
        group = getattr(self, "_actor_group", None)
        if group is None:
            return super()._policy_out_device(obs)
        A = self.actor
        out = th.empty(obs.shape[0], self.action_space.shape[0], dtype=th.float32, device=obs.device)
        group.forward([A._agent_obs_tensor_extract(j, obs) for j in range(self.n_agents)], out,
                      [A._range("act", j) for j in range(self.n_agents)])
        return out

    def _action_mode(self, warmup: bool) -> int:
        if not self.faithful_quirks:
            return super()._action_mode(warmup)
        return 2 if warmup else 3  # Q1: no scale/unscale round trip, no noise

    def _sample_action(self, learning_starts: int, action_noise=None, n_envs: int = 1):
        """reference: multiagent_policy_algorithm.py:346-396 (NumPy compatibility path), quirk Q1"""
        if not self.faithful_quirks:
            return super()._sample_action(learning_starts, action_noise, n_envs)
        if self.num_timesteps < learning_starts:
            unscaled_action = self.action_space.sample_batch(n_envs)
        else:
            unscaled_action, _ = self.predict(self._last_obs, deterministic=False)
        return unscaled_action, unscaled_action

    # ---- training -------------------------------------------------------------------------------------------------
    def _update_agent_learning_rates(self, agent_id: int) -> None:
        """reference: base_class.py:1109-1140 as called at maddpg.py:122 (quirk Q4)."""
        optimizers = [self.actor.optimizer_list[agent_id], self.critic.optimizer_list[agent_id]]
        for i, sched in enumerate(self.lr_schedule_list):
            lr = sched(self._current_progress_remaining)
            self.logger.record(f"train/agent_{i}_learning_rate", lr)
            if self.faithful_quirks:
                opt = optimizers[i] if i < len(optimizers) else None
                targets = [] if opt is None else [opt]
            else:
                targets = optimizers if i == agent_id else []
            for opt in targets:
                update_learning_rate(opt, lr)
                opt.sync_lr()

    def _use_packed_batch(self) -> bool:
        """Joint critics whose input is exactly (obs | actions): sample straight into the critic-input rows and let the
        target-smoothing kernel write every agent's next action into its columns (no gathers, no torch.cat)."""
        from core.common.buffers import ReplayBuffer

        rb, C = self.replay_buffer, self.critic
        return (self.fused_learner and not C.local and type(rb) is ReplayBuffer and rb.normalizer is None
                and C._tiles_in_order("obs") and C._tiles_in_order("act")
                and all(C._range("act", i) is not None and C._range("obs", i) is not None for i in range(self.n_agents)))

    def _packed_batch(self, batch_size: int):
        if self._packed is None or self._packed.x_data.shape[0] != batch_size:
            self._packed = self.replay_buffer.alloc_packed_batch(batch_size, with_pi=True)  # x_pi = (obs | .) for the actor step
            self._static_batch = self._packed.samples
            self._target_q = [th.empty(batch_size, 1, dtype=th.float32, device=self.device) for _ in range(self.n_agents)]
        return self._packed

    def _batch(self, batch_size: int):
        if self._static_batch is None or self._static_batch.observations.shape[0] != batch_size or self._packed is not None:
            self._packed = None
            self._static_batch = self.replay_buffer.alloc_batch(batch_size)
            self._target_q = [th.empty(batch_size, 1, dtype=th.float32, device=self.device) for _ in range(self.n_agents)]
        return self._static_batch

    def train(self, gradient_steps: int, batch_size: int) -> None:
        """reference: maddpg.py:117-191"""
        self.policy.set_training_mode(True)
        self._train_host_pre()
        self._train_device_only(gradient_steps, batch_size)
        self._train_host_only(gradient_steps)

    def _train_host_pre(self) -> None:
        for agent_id in range(self.n_agents):
            self._update_agent_learning_rates(agent_id)

    def _graph_eligible(self, callback) -> bool:
        return super()._graph_eligible(callback) and not self.noise_queue

    def _graph_phase(self) -> int:
        return self._n_updates % self.policy_delay

    def _train_host_only(self, gradient_steps: int) -> None:
        n_actor = (self._n_updates + gradient_steps) // self.policy_delay - self._n_updates // self.policy_delay
        self._n_updates += gradient_steps
        self.logger.record("train/n_updates", self._n_updates, exclude="tensorboard")
        for i in range(self.n_agents):
            if n_actor > 0:
                self.logger.record(f"train/agent_{i}_actor_loss", DeviceMean(self._loss_sums[f"actor{i}"], n_actor))
            self.logger.record(f"train/agent_{i}_critic_loss", DeviceMean(self._loss_sums[f"critic{i}"], gradient_steps))

    def _train_device_only(self, gradient_steps: int, batch_size: int) -> None:
        # a call without a policy update keeps the actors' last loss sums (slots [0, n_agents)), like maddpg.py:183-191 keeps
        # the last recorded value until the next actor update; the critics' slots are zeroed every call
        n_actor = (self._n_updates + gradient_steps) // self.policy_delay - self._n_updates // self.policy_delay
        (self._loss_sum_buf if n_actor > 0 else self._loss_sum_buf[self.n_agents:]).zero_()
        n_updates = self._n_updates
        A, C = self.actor, self.critic
        for _ in range(gradient_steps):
            n_updates += 1
            if self.fused_learner:
                self._gradient_step_fused(batch_size, n_updates)
                continue
            rd = self.replay_buffer.sample_into(self._batch(batch_size))
            next_actions_list = []
            with th.no_grad():
                for i in range(self.n_agents):  # :131-142
                    agent_next_obs = A._agent_obs_tensor_extract(i, rd.next_observations)
                    agent_action = A._agent_action_tensor_extract(i, rd.actions)
                    if self.noise_queue:
                        noise = self.noise_queue.pop(0).to(self.device)
                    else:
                        noise = agent_action.clone().normal_(0, self.target_policy_noise)
                    noise = noise.clamp(-self.target_noise_clip, self.target_noise_clip)
                    next_actions_list.append((self.actor_target.mu_list[i](agent_next_obs) + noise).clamp(-1, 1))
                next_actions = th.cat(next_actions_list, dim=-1)  # :144
            captured = []
            for i in range(self.n_agents):
                agent_obs = A._agent_obs_tensor_extract(i, rd.observations)
                with th.no_grad():  # :148-151
                    qs = self.critic_target.agent_forward(i, rd.next_observations, next_actions)
                    hip_ops.td_target_min(qs[0].contiguous(), qs[-1].contiguous(), None, rd.rewards, rd.dones, None, self.gamma,
                                          self._target_q[i])
                    target_q = self._target_q[i]
                current_q = C.agent_forward(i, rd.observations, rd.actions)  # :154 (only agent i's networks are needed)
                critic_loss = sum(F.mse_loss(q, target_q) for q in current_q)  # :157
                self._loss_sums[f"critic{i}"] += critic_loss.detach()
                C.optimizer_list[i].zero_grad()  # :162-164
                critic_loss.backward()
                self._allreduce_grads(self.policy.critic_slices[i])
                C.optimizer_list[i].step()
                actor_loss = None
                if n_updates % self.policy_delay == 0:  # :167-185
                    if self.faithful_quirks:  # Q2: every agent's actor sees agent i's observation slice
                        actions = th.cat([A.mu_list[j](agent_obs) for j in range(self.n_agents)], dim=-1)
                    else:
                        actions = A(rd.observations)
                    actor_loss = -C.agent_forward(i, rd.observations, actions, only_first=True)[0].mean()
                    self._loss_sums[f"actor{i}"] += actor_loss.detach()
                    A.optimizer_list[i].zero_grad()
                    actor_loss.backward()
                    self._allreduce_grads(self.policy.actor_slices[i])
                    A.optimizer_list[i].step()
                    if self.faithful_quirks:  # Q3: polyak inside the agent loop
                        self.policy.critic_target_arena.polyak_from(self.policy.critic_arena, self.tau)
                        self.policy.actor_target_arena.polyak_from(self.policy.actor_arena, self.tau)
                if self.debug_capture:
                    captured.append(dict(target_q=target_q.clone(), current_q=[q.detach().clone() for q in current_q],
                                         critic_loss=critic_loss.detach().clone(),
                                         actor_loss=None if actor_loss is None else actor_loss.detach().clone()))
            if not self.faithful_quirks and n_updates % self.policy_delay == 0:
                self.policy.critic_target_arena.polyak_from(self.policy.critic_arena, self.tau)
                self.policy.actor_target_arena.polyak_from(self.policy.actor_arena, self.tau)
            if self.debug_capture:
                self.last_train_tensors = dict(agents=captured)

    def _gradient_step_fused(self, batch_size: int, n_updates: int) -> None:
        """maddpg.py:127-185 on the fused path (core/common/fused.py): the same statements and quirks, GEMMs in rocBLAS,
        epilogues / loss roots in HIP, gradients written into the arenas, only the updating agent's networks evaluated."""
        pol, A, C = self.policy, self.actor, self.critic
        pb = None
        if self._use_packed_batch():
            pb = self.replay_buffer.sample_packed_into(self._packed_batch(batch_size))
            rd = pb.samples
        else:
            rd = self.replay_buffer.sample_into(self._batch(batch_size))
        B = rd.observations.shape[0]
        if not hasattr(self, "_g_bufs") or self._g_bufs.shape[1] != B:
            self._g_bufs = th.empty(2, B, 1, device=self.device)
        gq = self._g_bufs
        shared_next = shared_cur = None
        if pb is not None and self._actor_target_group is not None:
            with th.no_grad():  # :131-144: every target actor per layer in one launch, ONE smoothing launch for the joint action
                a_t = th.empty(B, pb.act_dim, dtype=th.float32, device=self.device)
                self._actor_target_group.forward([A._agent_obs_tensor_extract(i, rd.next_observations) for i in range(self.n_agents)],
                                                 a_t, [C._range("act", i) for i in range(self.n_agents)])
                queued = None
                if self.noise_queue:  # teacher-forced: one tensor per agent, in agent order
                    queued = th.cat([self.noise_queue.pop(0).to(self.device, th.float32) for _ in range(self.n_agents)], dim=1).contiguous()
                hip_ops.target_smooth(a_t, queued, None if queued is not None else self._device_rng(), self.target_policy_noise,
                                      self.target_noise_clip, pb.x_next[:, pb.obs_dim:])
            shared_next, shared_cur = pb.x_next, pb.x_data
        elif pb is not None:
            with th.no_grad():  # :131-144, one smoothing launch per agent, written into x_next's action columns
                for i in range(self.n_agents):
                    a_t = self._fast_actor_targets[i](A._agent_obs_tensor_extract(i, rd.next_observations), train_params=False)
                    queued = self.noise_queue.pop(0).to(self.device, th.float32).contiguous() if self.noise_queue else None
                    lo, hi = C._range("act", i)
                    hip_ops.target_smooth(a_t, queued, None if queued is not None else self._device_rng(), self.target_policy_noise,
                                          self.target_noise_clip, pb.x_next[:, pb.obs_dim + lo:pb.obs_dim + hi])
            shared_next, shared_cur = pb.x_next, pb.x_data
        with th.no_grad():  # :131-144
            nxt = []
            for i in range(self.n_agents if pb is None else 0):
                agent_next_obs = A._agent_obs_tensor_extract(i, rd.next_observations)
                if self.noise_queue:
                    noise = self.noise_queue.pop(0).to(self.device)
                else:
                    noise = th.empty(B, len(self.action_splits[i]), device=self.device).normal_(0, self.target_policy_noise)
                noise = noise.clamp(-self.target_noise_clip, self.target_noise_clip)
                nxt.append((self._fast_actor_targets[i](agent_next_obs, train_params=False) + noise).clamp(-1, 1))
            if pb is None:
                next_actions = th.cat(nxt, dim=-1)
                shared_next = None if C.local else self.critic_target._input(0, rd.next_observations, next_actions)
        if pb is None:
            shared_cur = None if C.local else C._input(0, rd.observations, rd.actions)
        cchain = self._critic_chain_for(B) if (pb is not None and shared_next is not None) else None
        if cchain is not None and BATCH_AGENT_CRITIC_STEPS and n_updates % self.policy_delay != 0:
            # a step WITHOUT a policy update on the row-chain kernels (core/common/chain.py:MaddpgCriticChain): one forward launch for
            # every agent's critic and target critic, a backward launch per agent, one dW / db + Adam launch per two agents
            captured_many = [] if self.debug_capture else None
            cchain.critic_steps_all(self, shared_cur, shared_next, rd, captured_many)
            if self.debug_capture:
                self.last_train_tensors = dict(agents=captured_many, batched_critic_steps=True)
            return
        if (BATCH_AGENT_CRITIC_STEPS and n_updates % self.policy_delay != 0 and shared_next is not None and B <= fused.LOSS_ROOT_MAX_ROWS
                and self.n_agents <= hip_ops.nv.MAX_ADAM_SEGS
                and all(fused.twin_pair_supported(c, t) and fused.loss_root_supported(c) for c, t in zip(self._fast_critics, self._fast_critic_targets))
                and len({c.acts[0] for c in self._fast_critics}) == 1):
            # A step WITHOUT a policy update: the agents' critic steps (:146-164) do not depend on each other (no soft update in
            # between), so their forward chains share pointer-table launches, their weight gradients one deferred launch pass and
            # their Adam steps one launch. (With a policy update, quirk Q3's soft updates inside the agent loop order the agents.)
            outs = fused.twin_pair_forward_many(self._fast_critics, self._fast_critic_targets, shared_cur, shared_next)
            captured_many = []
            with fused.deferred_weight_grads():
                for i, (qs, qs_t) in enumerate(outs):
                    td_root = dict(mode="td", q1_t=qs_t[0], q2_t=qs_t[1], next_logp=None, rew=rd.rewards, done=rd.dones, ent_coef=None,
                                   gamma=self.gamma, scale=1.0, q1=qs[0].detach(), q2=qs[1].detach(), target_out=self._target_q[i],
                                   loss_out=self._loss_now, loss_sum=self._loss_sums[f"critic{i}"], alpha=None)
                    with fused.loss_root(td_root):
                        fused.backward_q(qs, gq)
                    if self.debug_capture:  # the loss-root launch of agent i has run (only the weight gradients are deferred)
                        captured_many.append(dict(target_q=self._target_q[i].clone(), current_q=[q.detach().clone() for q in qs],
                                                  critic_loss=self._loss_now.clone(), actor_loss=None))
            for i in range(self.n_agents):
                self._allreduce_grads(pol.critic_slices[i])
            C.optimizer_list[0].step_with(*C.optimizer_list[1:])
            if self.debug_capture:
                self.last_train_tensors = dict(agents=captured_many, batched_critic_steps=True)
            return
        captured = []
        for i in range(self.n_agents):
            x_next = shared_next if shared_next is not None else self.critic_target._input(i, rd.next_observations, next_actions)
            x_cur = shared_cur if shared_cur is not None else C._input(i, rd.observations, rd.actions)
            if cchain is not None:  # agent i's critic step on the row-chain kernels: 3 launches instead of 7
                cchain.critic_step(self, i, x_cur, x_next, rd)
                qs = None
            else:
                if fused.twin_pair_supported(self._fast_critics[i], self._fast_critic_targets[i]):
                    # :154 and :148-151 as ONE four-network chain (three launches instead of six)
                    qs, qs_t = fused.twin_pair_forward(self._fast_critics[i], self._fast_critic_targets[i], x_cur, x_next)
                else:
                    with th.no_grad():  # :148-151
                        qs_t = self._fast_critic_targets[i].forward_input(x_next, train_params=False)
                    qs = self._fast_critics[i].forward_input(x_cur)  # :154
                scale = 1.0 if len(qs) == 2 else 0.5
                root = len(qs) == 2 and qs.stacked is not None and B <= fused.LOSS_ROOT_MAX_ROWS and fused.loss_root_supported(self._fast_critics[i])
                td_root = None
                if root:  # TD target (:148-151) + critic loss (:157-159) inside the critic backward's first launch
                    td_root = dict(mode="td", q1_t=qs_t[0], q2_t=qs_t[-1], next_logp=None, rew=rd.rewards, done=rd.dones, ent_coef=None,
                                   gamma=self.gamma, scale=scale, q1=qs[0].detach(), q2=qs[-1].detach(), target_out=self._target_q[i],
                                   loss_out=self._loss_now, loss_sum=self._loss_sums[f"critic{i}"], alpha=None)
                else:  # TD target + critic loss in one launch
                    hip_ops.td_twin_q_loss(qs_t[0], qs_t[-1], None, rd.rewards, rd.dones, None, self.gamma, qs[0], qs[-1], scale,
                                           self._target_q[i], gq[0], gq[1], self._loss_now, self._loss_sums[f"critic{i}"])
                if len(qs) == 2:
                    with fused.loss_root(td_root):
                        fused.backward_q(qs, gq)  # :162-164
                else:
                    with fused.deferred_weight_grads():
                        th.autograd.backward([qs[0]], [gq[0] + gq[1]])
                self._allreduce_grads(pol.critic_slices[i])
                C.optimizer_list[i].step()
            critic_loss_now = self._loss_now.clone() if self.debug_capture else None
            actor_loss_now = None
            if n_updates % self.policy_delay == 0:  # :167-185
                agent_obs = A._agent_obs_tensor_extract(i, rd.observations)
                if pb is not None and self._actor_group is not None:
                    # every agent's actor per layer in one launch, actions written into the critic input's columns (no cat);
                    # Q2: every actor sees agent i's observation slice in the reference
                    ins = [agent_obs if self.faithful_quirks else A._agent_obs_tensor_extract(j, rd.observations) for j in range(self.n_agents)]
                    cols = [(pb.obs_dim + lo, pb.obs_dim + hi) for lo, hi in (C._range("act", j) for j in range(self.n_agents))]
                    x_pi = self._actor_group.forward(ins, pb.x_pi.detach(), cols, grad_agent=i)
                else:
                    if self.faithful_quirks:  # Q2: every agent's actor sees agent i's observation slice
                        acts = [self._fast_actors[j](agent_obs, train_params=(j == i)) for j in range(self.n_agents)]
                    else:
                        acts = [self._fast_actors[j](A._agent_obs_tensor_extract(j, rd.observations), train_params=(j == i))
                                for j in range(self.n_agents)]
                    if pb is not None:  # (obs | all agents' actions) in ONE cat
                        x_pi = th.cat([rd.observations] + acts, dim=1)
                    else:
                        x_pi = C._input(i, rd.observations, th.cat(acts, dim=-1))
                if cchain is not None and tuple(x_pi.shape) == (B, cchain.W) and x_pi.is_contiguous() and x_pi.requires_grad:
                    # -mean(Q1) (:177) and its gradient down to the critic input on the chain kernels (3 launches instead of 6); the actors'
                    # backward (per-layer kernels) reads the action columns
                    g_x = cchain.actor_loss_grad(self, i, x_pi.detach())
                    with fused.deferred_weight_grads():
                        th.autograd.backward([x_pi], [g_x])
                    qs_pi = None
                else:
                    qs_pi = self._fast_critics[i].forward_input(x_pi, train_params=False, only_first=True)
                if qs_pi is None:
                    pass
                elif qs_pi.stacked is not None and B <= fused.LOSS_ROOT_MAX_ROWS and fused.loss_root_supported(self._fast_critics[i]):
                    # -mean(Q1) (:177) rides in the first launch of the backward through the (frozen) first Q network
                    with fused.loss_root(dict(mode="neg_mean", q1=qs_pi[0].detach(), loss_out=self._loss_now,
                                              loss_sum=self._loss_sums[f"actor{i}"])):
                        fused.backward_q(qs_pi, gq)
                else:
                    hip_ops.neg_mean_loss(qs_pi[0], gq[0], self._loss_now, self._loss_sums[f"actor{i}"])
                    fused.backward_q(qs_pi, gq)
                self._allreduce_grads(pol.actor_slices[i])
                if self.faithful_quirks:
                    # Q3: polyak of BOTH whole arenas inside the agent loop (:183-185), in the launch of agent i's actor step: the
                    # critics' and the other agents' actor parameters do not change in it, agent i's target follows its new weights
                    src, tgt, sl = pol.actor_arena.flat, pol.actor_target_arena.flat, pol.actor_slices[i]
                    if not (pol.critic_target_arena.same_layout(pol.critic_arena) and pol.actor_target_arena.same_layout(pol.actor_arena)):
                        raise ValueError("Iterables have different lengths")  # zip_strict's error (utils.py:447)
                    lo = sl.flat.storage_offset() - src.storage_offset()
                    hi = lo + sl.numel
                    A.optimizer_list[i].step_with(polyak=[(pol.critic_arena.flat, pol.critic_target_arena.flat, self.tau),
                                                          (src[:lo], tgt[:lo], self.tau), (src[hi:], tgt[hi:], self.tau)],
                                                  own_target=(tgt[lo:hi], self.tau))
                else:
                    A.optimizer_list[i].step()
                if self.debug_capture:
                    actor_loss_now = self._loss_now.clone()
            if self.debug_capture:
                current = ([cchain.q_out[i, 0].clone().view(B, 1), cchain.q_out[i, 1].clone().view(B, 1)] if qs is None
                           else [q.detach().clone() for q in qs])
                captured.append(dict(target_q=self._target_q[i].clone(), current_q=current, critic_loss=critic_loss_now, actor_loss=actor_loss_now))
        if not self.faithful_quirks and n_updates % self.policy_delay == 0:
            pol.critic_target_arena.polyak_from(pol.critic_arena, self.tau)
            pol.actor_target_arena.polyak_from(pol.actor_arena, self.tau)
        if self.debug_capture:
            self.last_train_tensors = dict(agents=captured)

    def _critic_chain_for(self, batch_size: int):
        """The critic steps on the row-chain kernels for this batch size (core/common/chain.py:MaddpgCriticChain), or None."""
        from core.common import chain

        cache = self.__dict__.setdefault("_chain_cache", {})
        key = (batch_size, chain.USE_CHAIN, fused.USE_FUSED_LINEAR)
        if key not in cache:
            cache[key] = chain.MaddpgCriticChain(self, batch_size) if chain.MaddpgCriticChain.supported(self, batch_size) else None
        return cache[key]

    def learn(self, total_timesteps: int, callback=None, log_interval: int = 4, tb_log_name: str = "MADDPG",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        return super().learn(total_timesteps=total_timesteps, callback=callback, log_interval=log_interval,
                             tb_log_name=tb_log_name, reset_num_timesteps=reset_num_timesteps, progress_bar=progress_bar)
