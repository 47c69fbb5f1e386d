"""TD3 (and DDPG as its special case) with the reference's constructor and `train()` arithmetic
(reference: core/td3/td3.py:19-240, core/ddpg/ddpg.py:14-130)."""
from typing import List, Optional, Union

import torch as th
from torch.nn import functional as F

from core.common import fused, hip_ops
from core.common.logger import DeviceMean
from core.common.off_policy_algorithm import OffPolicyAlgorithm
from core.td3.policies import MlpPolicy


class TD3(OffPolicyAlgorithm):
    policy_aliases = {"MlpPolicy": MlpPolicy}

    def __init__(self, policy, env, learning_rate=1e-3, buffer_size: int = 1_000_000, learning_starts: int = 100,
                 batch_size: int = 256, tau: float = 0.005, gamma: float = 0.99, train_freq: Union[int, tuple] = 1,
                 gradient_steps: int = 1, action_noise=None, replay_buffer_class=None, replay_buffer_kwargs: Optional[dict] = None,
                 optimize_memory_usage: bool = False, policy_delay: int = 2, target_policy_noise: float = 0.2,
                 target_noise_clip: float = 0.5, stats_window_size: int = 100, tensorboard_log: Optional[str] = None,
                 policy_kwargs: Optional[dict] = None, verbose: int = 0, seed: Optional[int] = None, device="auto",
                 _init_setup_model: bool = True):
        super().__init__(policy, env, learning_rate, buffer_size, learning_starts, batch_size, tau, gamma, train_freq,
                         gradient_steps, action_noise=action_noise, replay_buffer_class=replay_buffer_class,
                         replay_buffer_kwargs=replay_buffer_kwargs, policy_kwargs=policy_kwargs,
                         stats_window_size=stats_window_size, tensorboard_log=tensorboard_log, verbose=verbose, device=device,
                         seed=seed, sde_support=False, optimize_memory_usage=optimize_memory_usage,
                         supported_action_spaces=(object,), support_multi_env=True)
        self.policy_delay = policy_delay
        self.target_noise_clip = target_noise_clip
        self.target_policy_noise = target_policy_noise
        self.debug_capture = False
        self.last_train_tensors: dict = {}
        self.noise_queue: List[th.Tensor] = []  # teacher-forcing hook for the target-smoothing noise (td3.py:169)
        if _init_setup_model:
            self._setup_model()

    def _setup_model(self) -> None:
        super()._setup_model()
        self.actor, self.actor_target = self.policy.actor, self.policy.actor_target
        self.critic, self.critic_target = self.policy.critic, self.policy.critic_target
        z = lambda: th.zeros(1, dtype=th.float32, device=self.device)  # noqa: E731
        self._loss_sum_buf = th.zeros(2, dtype=th.float32, device=self.device)  # one fill per train() instead of two
        self._loss_sums = dict(actor=self._loss_sum_buf[0:1], critic=self._loss_sum_buf[1:2])
        self._loss_now = dict(actor=z(), critic=z())
        self._static_batch, self._packed = None, None
        from core.common.arena import FlatAdam

        self.fused_learner = (isinstance(self.actor.optimizer, FlatAdam) and isinstance(self.critic.optimizer, FlatAdam)
                              and fused.FastMLP.supported(self.actor.mu) and all(fused.FastMLP.supported(q) for q in self.critic.q_networks))
        if self.fused_learner:
            self._fast_actor, self._fast_actor_target = fused.FastMLP(self.actor.mu, self.actor.optimizer), fused.FastMLP(self.actor_target.mu)
            self._fast_critic, self._fast_critic_target = fused.FastTwinCritic(self.critic, self.policy.critic_stack), fused.FastTwinCritic(self.critic_target, self.policy.critic_target_stack)

    def _policy_out_device(self, obs: th.Tensor) -> th.Tensor:
        if not self.fused_learner:
            return super()._policy_out_device(obs)
        with th.no_grad():
            return self._fast_actor(obs, train_params=False)

    def _rollout_net(self):
        return self._fast_actor.rollout_operands(self._denv.obs) if self.fused_learner else None

    def _batch(self, batch_size: int):
        if self._static_batch is None or self._static_batch.observations.shape[0] != batch_size or self._packed is not None:
            self._static_batch, self._packed = self.replay_buffer.alloc_batch(batch_size), None
            self._target_q = th.empty(batch_size, 1, dtype=th.float32, device=self.device)
        return self._static_batch

    def _use_packed_batch(self) -> bool:
        """Sample straight into the critics' input rows (no torch.cat launches) on the fused path with the stock buffer."""
        from core.common.buffers import ReplayBuffer

        rb = self.replay_buffer
        return self.fused_learner and type(rb) is ReplayBuffer and rb.normalizer is None

    def _packed_batch(self, batch_size: int):
        if self._packed is None or self._packed.x_data.shape[0] != batch_size:
            self._packed = self.replay_buffer.alloc_packed_batch(batch_size)  # x_pi: the actor writes its action into the critic input
            self._static_batch = self._packed.samples
            self._target_q = th.empty(batch_size, 1, dtype=th.float32, device=self.device)
            self._act_t = th.empty(batch_size, self._packed.act_dim, dtype=th.float32, device=self.device)
        return self._packed

    def train(self, gradient_steps: int, batch_size: int = 100) -> None:
        """reference: td3.py:154-211"""
        self.policy.set_training_mode(True)
        self._train_host_pre()
        self._train_device_only(gradient_steps, batch_size)
        self._train_host_only(gradient_steps)

    def _train_host_pre(self) -> None:
        self._update_learning_rate([self.actor.optimizer, self.critic.optimizer])

    def _graph_eligible(self, callback) -> bool:
        return super()._graph_eligible(callback) and not self.noise_queue

    def _graph_phase(self) -> int:
        # the delayed policy update makes iterations differ: one captured graph per residue of the update counter
        return self._n_updates % self.policy_delay

    def _train_host_only(self, gradient_steps: int) -> None:
        n_actor = (self._n_updates + gradient_steps) // self.policy_delay - self._n_updates // self.policy_delay
        self._n_updates += gradient_steps
        self.logger.record("train/n_updates", self._n_updates, exclude="tensorboard")
        if n_actor > 0:
            self.logger.record("train/actor_loss", DeviceMean(self._loss_sums["actor"], n_actor))
        self.logger.record("train/critic_loss", DeviceMean(self._loss_sums["critic"], gradient_steps))

    def _train_device_only(self, gradient_steps: int, batch_size: int) -> None:
        # the reference keeps the last np.mean(actor_losses) until the next actor update (td3.py:207-211): a call without a
        # policy update zeroes the critic slot only, so a log dump after it still resolves the lazily-read actor loss
        n_actor = (self._n_updates + gradient_steps) // self.policy_delay - self._n_updates // self.policy_delay
        # one gradient step per train() call (the default) on the fused path: the loss kernels STORE the logged values into the
        # sums (an actor-less call leaves the actor's slot alone), no zero-fill launch
        self._single_step = gradient_steps == 1 and self.fused_learner
        if not self._single_step:
            (self._loss_sum_buf if n_actor > 0 else self._loss_sum_buf[1:2]).zero_()
        n_updates = self._n_updates  # host counter advances in _train_host_only
        for _ in range(gradient_steps):
            n_updates += 1
            if self.fused_learner:
                self._gradient_step_fused(batch_size, n_updates)
                continue
            replay_data = self.replay_buffer.sample_into(self._batch(batch_size))
            with th.no_grad():
                if self.noise_queue:
                    noise = self.noise_queue.pop(0).to(self.device)
                else:
                    noise = replay_data.actions.clone().normal_(0, self.target_policy_noise)  # :169
                noise = noise.clamp(-self.target_noise_clip, self.target_noise_clip)
                next_actions = (self.actor_target(replay_data.next_observations) + noise).clamp(-1, 1)
                qs = self.critic_target(replay_data.next_observations, next_actions)
                q1_t, q2_t = qs[0], qs[-1]  # n_critics == 1 (DDPG): min over one network
                hip_ops.td_target_min(q1_t.contiguous(), q2_t.contiguous(), None, replay_data.rewards, replay_data.dones,
                                      None, self.gamma, self._target_q)
                target_q_values = self._target_q
            current_q_values = self.critic(replay_data.observations, replay_data.actions)
            critic_loss = sum(F.mse_loss(current_q, target_q_values) for current_q in current_q_values)  # no 0.5 (:182)
            self._loss_sums["critic"] += critic_loss.detach()
            self.critic.optimizer.zero_grad()
            critic_loss.backward()
            self._allreduce_grads(self.policy.critic_arena)
            self.critic.optimizer.step()
            actor_loss = None
            if n_updates % self.policy_delay == 0:  # :192-206
                actor_loss = -self.critic.q1_forward(replay_data.observations, self.actor(replay_data.observations)).mean()
                self._loss_sums["actor"] += actor_loss.detach()
                self.actor.optimizer.zero_grad()
                actor_loss.backward()
                self._allreduce_grads(self.policy.actor_arena)
                self.actor.optimizer.step()
                self.policy.critic_target_arena.polyak_from(self.policy.critic_arena, self.tau)
                self.policy.actor_target_arena.polyak_from(self.policy.actor_arena, self.tau)
            if self.debug_capture:
                self.last_train_tensors = dict(target_q=target_q_values.clone(), current_q=[q.detach().clone() for q in current_q_values],
                                               critic_loss=critic_loss.detach().clone(),
                                               actor_loss=None if actor_loss is None else actor_loss.detach().clone())

    def _gradient_step_fused(self, batch_size: int, n_updates: int) -> None:
        """td3.py:161-206 on the fused path (core/common/fused.py)."""
        s, pol = self._loss_sums, self.policy
        pb, gather = None, None
        chain = self._chain_for(batch_size)
        if chain is not None:  # the row-chain kernels (core/common/chain.py): 4 launches (9 with the policy step) instead of 13 (22)
            pb = self._packed_batch(batch_size)
            gather = self.replay_buffer.take_predrawn(pb) if fused.USE_GATHER_IN_FIRST_LAYER else None
            if gather is None:
                self.replay_buffer.sample_packed_into(pb)  # :161 + the critics' cat([obs, act])
            return chain.step(self, pb, gather, n_updates)
        if self._use_packed_batch():
            pb = self._packed_batch(batch_size)
            if fused.USE_GATHER_IN_FIRST_LAYER and self._fast_actor_target.gather_supported(pb.samples.next_observations):
                # indices drawn by the rollout launch: the target actor's first layer below gathers the rows itself
                gather = self.replay_buffer.take_predrawn(pb)
            if gather is None:
                self.replay_buffer.sample_packed_into(pb)  # :161 + the critics' cat([obs, act])
            rd = pb.samples
        else:
            rd = self.replay_buffer.sample_into(self._batch(batch_size))
        B = rd.observations.shape[0]
        if not hasattr(self, "_g_bufs") or self._g_bufs[0].shape[0] != B:
            self._g_bufs = th.empty(2, B, 1, device=self.device)
        gq = self._g_bufs
        gq1, gq2 = gq[0], gq[1]
        with th.no_grad():  # :167-176
            if pb is not None:
                # target smoothing in ONE launch (clone + normal_ + clamp + add + clamp), written into x_next's action columns
                queued = self.noise_queue.pop(0).to(self.device, th.float32).contiguous() if self.noise_queue else None
                rng = None if queued is not None else self._device_rng()
                if fused.USE_SMOOTH_IN_LAST_LAYER and self._fast_actor_target.smooth_supported(rd.next_observations):
                    # ... inside the target actor's last layer (one launch less)
                    self._fast_actor_target(rd.next_observations, train_params=False, gather=gather,
                                            smooth=dict(noise=queued, rng_ctl=rng, sigma=self.target_policy_noise, clip=self.target_noise_clip,
                                                        out=pb.x_next[:, pb.obs_dim:]))
                else:
                    a_t = self._fast_actor_target(rd.next_observations, train_params=False, gather=gather)
                    hip_ops.target_smooth(a_t, queued, rng, self.target_policy_noise, self.target_noise_clip, pb.x_next[:, pb.obs_dim:])
                twin_pair = fused.twin_pair_supported(self._fast_critic, self._fast_critic_target)
                if not twin_pair:
                    qs = self._fast_critic_target.forward_input(pb.x_next, train_params=False)
            else:
                noise = self.noise_queue.pop(0).to(self.device) if self.noise_queue else rd.actions.clone().normal_(0, self.target_policy_noise)
                noise = noise.clamp(-self.target_noise_clip, self.target_noise_clip)
                next_actions = (self._fast_actor_target(rd.next_observations, train_params=False) + noise).clamp(-1, 1)
                qs = self._fast_critic_target(rd.next_observations, next_actions, train_params=False)
            if pb is None or not twin_pair:
                q1_t, q2_t = qs[0], qs[-1]
        if pb is not None and twin_pair:  # :179 and :173 as ONE four-network chain (three launches instead of six)
            qs, (q1_t, q2_t) = fused.twin_pair_forward(self._fast_critic, self._fast_critic_target, pb.x_data, pb.x_next)
        else:
            qs = self._fast_critic.forward_input(pb.x_data) if pb is not None else self._fast_critic(rd.observations, rd.actions)  # :179
        q1, q2 = qs[0], qs[-1]
        # TD target (:174-176) + critic loss (:182) in one launch; n_critics == 1 (DDPG): loss = mse(q1, t) -> scale 0.5 of
        # the doubled term. Twin critics: inside the critic backward's first launch (cstr_hidden_head_bwd_root_f32).
        single = getattr(self, "_single_step", False)
        c_out, c_sum = (s["critic"], None) if single else (self._loss_now["critic"], s["critic"])
        root = len(qs) == 2 and qs.stacked is not None and B <= fused.LOSS_ROOT_MAX_ROWS and fused.loss_root_supported(self._fast_critic)
        td_root = None
        if root:
            td_root = dict(mode="td", q1_t=q1_t, q2_t=q2_t, next_logp=None, rew=rd.rewards, done=rd.dones, ent_coef=None,
                           gamma=self.gamma, scale=1.0, q1=q1.detach(), q2=q2.detach(), target_out=self._target_q,
                           loss_out=c_out, loss_sum=c_sum, alpha=None)
        else:
            hip_ops.td_twin_q_loss(q1_t, q2_t, None, rd.rewards, rd.dones, None, self.gamma, q1, q2, 1.0 if len(qs) == 2 else 0.5,
                                   self._target_q, gq1, gq2, c_out, c_sum)
        if len(qs) == 2:
            with fused.loss_root(td_root):
                fused.backward_q(qs, gq)
        else:
            with fused.deferred_weight_grads():
                th.autograd.backward([q1], [gq1 + gq2])
        self._allreduce_grads(pol.critic_arena)
        self.critic.optimizer.step()
        actor_done = False
        if n_updates % self.policy_delay == 0:  # :192-206
            if pb is not None and fused.USE_FUSED_LINEAR and self._fast_actor.layers[-1][0].out_features > 1:
                # the actor's last layer writes into x_pi = (obs | .): no torch.cat, and its backward reads the action columns
                # of the critic's input gradient in place
                x_pi = self._fast_actor(rd.observations, xbuf=pb.x_pi.detach())
                qs_pi = self._fast_critic.forward_input(x_pi, train_params=False, only_first=True)
            else:
                a = self._fast_actor(rd.observations)
                qs_pi = self._fast_critic(rd.observations, a, train_params=False, only_first=True)
            a_out, a_sum = (s["actor"], None) if single else (self._loss_now["actor"], s["actor"])
            if qs_pi.stacked is not None and B <= fused.LOSS_ROOT_MAX_ROWS and fused.loss_root_supported(self._fast_critic):
                # -mean(Q1) (:194) rides in the first launch of the backward through the (frozen) first Q network
                with fused.loss_root(dict(mode="neg_mean", q1=qs_pi[0].detach(), loss_out=a_out, loss_sum=a_sum)):
                    fused.backward_q(qs_pi, gq)
            else:
                hip_ops.neg_mean_loss(qs_pi[0], gq1, a_out, a_sum)
                fused.backward_q(qs_pi, gq)
            self._allreduce_grads(pol.actor_arena)
            # the actor's step, the critics' soft update (disjoint arenas) and the actor target's soft update (by the threads that
            # have just computed the new actor weights) in ONE launch (:199, :204, :205)
            if not pol.actor_target_arena.same_layout(pol.actor_arena):
                raise ValueError("Iterables have different lengths")  # zip_strict's error (utils.py:447)
            self.actor.optimizer.step_with(polyak=(pol.critic_arena, pol.critic_target_arena, self.tau),
                                           own_target=(pol.actor_target_arena.flat, self.tau))
            actor_done = True
        if self.debug_capture:
            self.last_train_tensors = dict(target_q=self._target_q.clone(), current_q=[q.detach().clone() for q in qs],
                                           critic_loss=c_out.clone(),
                                           actor_loss=a_out.clone() if actor_done else None)

    def _chain_for(self, batch_size: int):
        """The row-chain form of the gradient step for this batch size (core/common/chain.py), or None: per-layer fused path."""
        from core.common import chain

        cache = self.__dict__.setdefault("_chain_cache", {})
        key = (batch_size, chain.USE_CHAIN, fused.USE_FUSED_LINEAR)
        if key not in cache:
            cache[key] = chain.Td3Chain(self, batch_size) if chain.Td3Chain.supported(self, batch_size) else None
        return cache[key]

    def _get_torch_save_params(self) -> tuple:
        """reference: td3.py:234-240"""
        return ["policy", "actor.optimizer", "critic.optimizer"], []

    def _extra_save_data(self) -> dict:
        return dict(policy_delay=self.policy_delay, target_policy_noise=self.target_policy_noise, target_noise_clip=self.target_noise_clip)

    @classmethod
    def _ctor_keys(cls) -> tuple:
        return super()._ctor_keys() + ("policy_delay", "target_policy_noise", "target_noise_clip")

    def learn(self, total_timesteps: int, callback=None, log_interval: int = 4, tb_log_name: str = "TD3",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        return super().learn(total_timesteps=total_timesteps, callback=callback, log_interval=log_interval,
                             tb_log_name=tb_log_name, reset_num_timesteps=reset_num_timesteps, progress_bar=progress_bar)


class DDPG(TD3):
    """reference: core/ddpg/ddpg.py:14-130 -- TD3 with policy_delay 1, one critic and a smoothing draw clamped to [-0, 0]
    (the reference passes target_policy_noise=0.1, target_noise_clip=0.0: the attribute values are kept, the noise is zero)."""

    @classmethod
    def _ctor_keys(cls) -> tuple:
        return OffPolicyAlgorithm._ctor_keys()

    def __init__(self, policy, env, learning_rate=1e-3, buffer_size: int = 1_000_000, learning_starts: int = 100,
                 batch_size: int = 256, tau: float = 0.005, gamma: float = 0.99, train_freq: Union[int, tuple] = 1,
                 gradient_steps: int = 1, action_noise=None, replay_buffer_class=None, replay_buffer_kwargs=None,
                 optimize_memory_usage: bool = False, tensorboard_log: Optional[str] = None, policy_kwargs: Optional[dict] = None,
                 verbose: int = 0, seed: Optional[int] = None, device="auto", _init_setup_model: bool = True):
        policy_kwargs = dict(policy_kwargs or {})
        policy_kwargs.setdefault("n_critics", 1)
        super().__init__(policy, env, learning_rate, buffer_size, learning_starts, batch_size, tau, gamma, train_freq,
                         gradient_steps, action_noise, replay_buffer_class, replay_buffer_kwargs, optimize_memory_usage,
                         policy_delay=1, target_policy_noise=0.1, target_noise_clip=0.0, tensorboard_log=tensorboard_log,
                         policy_kwargs=policy_kwargs, verbose=verbose, seed=seed, device=device,
                         _init_setup_model=_init_setup_model)
