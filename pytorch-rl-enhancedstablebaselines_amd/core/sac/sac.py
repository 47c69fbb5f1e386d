"""Soft Actor-Critic with the reference's constructor and `train()` arithmetic (reference: core/sac/sac.py:21-340).

One gradient step = HIP sampler (bit-exact indices + gather) -> actor/critic MLP forward/backward on
PyTorch-ROCm -> HIP `td_target_min` -> three one-launch Adam steps on flat arenas (RCCL all-reduce of each
arena's gradient first when data-parallel) -> one-launch polyak. Losses stay on the device: the reference's
four `.item()` syncs per step (:232, :236, :263, :276) are replaced by device-side running sums that the
logger resolves only when it dumps.
"""
from typing import Optional, Union

import numpy as np
import torch as th
from torch.nn import functional as F

from core.common import fused, hip_ops
from core.common.arena import FlatAdam, ParamArena
from core.common.logger import DeviceMean
from core.common.off_policy_algorithm import OffPolicyAlgorithm
from core.sac.policies import MlpPolicy


class SAC(OffPolicyAlgorithm):
    policy_aliases = {"MlpPolicy": MlpPolicy}

    def __init__(self, policy, env, learning_rate=3e-4, buffer_size: int = 1_000_000, learning_starts: int = 100,
                 batch_size: int = 256, tau: float = 0.005, gamma: float = 0.99, train_freq: Union[int, tuple] = 1,
                 gradient_steps: int = 1, action_noise=None, replay_buffer_class=None, replay_buffer_kwargs: Optional[dict] = None,
                 optimize_memory_usage: bool = False, ent_coef: Union[str, float] = "auto", target_update_interval: int = 1,
                 target_entropy: Union[str, float] = "auto", use_sde: bool = False, sde_sample_freq: int = -1,
                 use_sde_at_warmup: bool = False, stats_window_size: int = 100, tensorboard_log: Optional[str] = None,
                 policy_kwargs: Optional[dict] = None, verbose: int = 0, seed: Optional[int] = None, device="auto",
                 _init_setup_model: bool = True):
        super().__init__(policy, env, learning_rate, buffer_size, learning_starts, batch_size, tau, gamma, train_freq,
                         gradient_steps, action_noise, replay_buffer_class=replay_buffer_class,
                         replay_buffer_kwargs=replay_buffer_kwargs, policy_kwargs=policy_kwargs,
                         stats_window_size=stats_window_size, tensorboard_log=tensorboard_log, verbose=verbose, device=device,
                         seed=seed, use_sde=use_sde, sde_sample_freq=sde_sample_freq, use_sde_at_warmup=use_sde_at_warmup,
                         optimize_memory_usage=optimize_memory_usage, supported_action_spaces=(object,), support_multi_env=True)
        self.target_entropy = target_entropy
        self.log_ent_coef: Optional[th.Tensor] = None
        self.ent_coef = ent_coef
        self.target_update_interval = target_update_interval
        self.ent_coef_optimizer = None
        self.debug_capture = False   # tests: keep target_q / current_q of the last gradient step
        self.last_train_tensors: dict = {}
        if _init_setup_model:
            self._setup_model()

    def _setup_model(self) -> None:
        """reference: sac.py:159-192"""
        super()._setup_model()
        self._create_aliases()
        if self.target_entropy == "auto":
            self.target_entropy = float(-np.prod(self.env.action_space.shape).astype(np.float32))
        else:
            self.target_entropy = float(self.target_entropy)
        if isinstance(self.ent_coef, str) and self.ent_coef.startswith("auto"):
            init_value = 1.0
            if "_" in self.ent_coef:
                init_value = float(self.ent_coef.split("_")[1])
                assert init_value > 0.0, "The initial value of ent_coef must be greater than 0"
            p = th.nn.Parameter(th.log(th.ones(1) * init_value))
            # its gradient lives in the tail of the critic's gradient buffer: one all-reduce serves both (data-parallel)
            tail = getattr(self.policy.critic_arena, "grad_tail", None)
            self._ent_arena = ParamArena([p], self.device, grad_storage=tail)
            self._ent_rides_critic = tail is not None
            self.log_ent_coef = p
            self.ent_coef_optimizer = FlatAdam(self._ent_arena, lr=self.lr_schedule(1))
            if self.world_size > 1:
                self.ent_coef_optimizer.grad_scale = 1.0 / self.world_size
        else:
            self.ent_coef_tensor = th.tensor(float(self.ent_coef), device=self.device)
        z = lambda: th.zeros(1, dtype=th.float32, device=self.device)  # noqa: E731
        self._loss_sum_buf = th.zeros(4, dtype=th.float32, device=self.device)  # one fill per train() instead of four
        sb = self._loss_sum_buf
        self._loss_sums = dict(actor=sb[0:1], critic=sb[1:2], ent_coef_loss=sb[2:3], ent_coef=sb[3:4])
        self._loss_now = dict(actor=z(), critic=z())
        self._ent_coef_buf = z()
        self._static_batch, self._packed = None, None
        # fused learner path (core/common/fused.py): GEMMs in rocBLAS, everything else hand-written HIP
        self.fused_learner = self._fused_supported()
        if self.fused_learner:
            self._fast_actor = fused.FastSacActor(self.actor, self.policy.actor_head)
            self._fast_critic = fused.FastTwinCritic(self.critic, self.policy.critic_stack)
            self._fast_critic_target = fused.FastTwinCritic(self.critic_target, self.policy.critic_target_stack)

    def _fused_supported(self) -> bool:
        from core.common.arena import FlatAdam

        return (len(self.critic.q_networks) == 2 and isinstance(self.actor.optimizer, FlatAdam)
                and isinstance(self.critic.optimizer, FlatAdam) and fused.FastMLP.supported(self.actor.latent_pi)
                and all(fused.FastMLP.supported(q) for q in self.critic.q_networks))

    def _policy_out_device(self, obs: th.Tensor) -> th.Tensor:
        if not self.fused_learner:
            return super()._policy_out_device(obs)
        fa = self._fast_actor
        with th.no_grad():
            out = fa.action_log_prob(obs, train_params=False, want_logp=False, defer_rng=True)[0]
        self._rng_advance, fa.deferred_rng = fa.deferred_rng, None  # the fused collect launch advances the Philox offset
        return out

    def _rollout_net(self):
        return self._fast_actor.rollout_operands(self._denv.obs) if self.fused_learner else None

    def _create_aliases(self) -> None:
        self.actor = self.policy.actor
        self.critic = self.policy.critic
        self.critic_target = self.policy.critic_target

    def _batch(self, batch_size: int):
        if self._static_batch is None or self._static_batch.observations.shape[0] != batch_size or self._packed is not None:
            self._static_batch, self._packed = self.replay_buffer.alloc_batch(batch_size), None
            self._target_q = th.empty(batch_size, 1, dtype=th.float32, device=self.device)
        return self._static_batch

    def _use_packed_batch(self) -> bool:
        """Sample straight into the critics' input rows (no torch.cat launches): the fused path with the merged actor head and
        the stock ReplayBuffer without a VecNormalize normaliser."""
        from core.common.buffers import ReplayBuffer

        rb = self.replay_buffer
        return (self.fused_learner and self._fast_actor.head is not None and type(rb) is ReplayBuffer and rb.normalizer is None
                and self._fast_actor.act_dim <= hip_ops.nv.MAX_HEAD_ACT)

    def _packed_batch(self, batch_size: int):
        if self._packed is None or self._packed.x_data.shape[0] != batch_size:
            self._packed = self.replay_buffer.alloc_packed_batch(batch_size)
            self._static_batch = self._packed.samples
            self._target_q = th.empty(batch_size, 1, dtype=th.float32, device=self.device)
        return self._packed

    def train(self, gradient_steps: int, batch_size: int = 64) -> None:
        """reference: sac.py:199-296 = host prologue (lr schedule) + device work + host epilogue (logger)."""
        self.policy.set_training_mode(True)
        self._train_host_pre()
        self._train_device_only(gradient_steps, batch_size)
        self._train_host_only(gradient_steps)

    def _train_host_pre(self) -> None:
        optimizers = [self.actor.optimizer, self.critic.optimizer]
        if self.ent_coef_optimizer is not None:
            optimizers += [self.ent_coef_optimizer]
        self._update_learning_rate(optimizers)  # :208

    def _train_device_only(self, gradient_steps: int, batch_size: int) -> None:
        # one gradient step per train() call (the default): the loss kernels STORE the logged values, no zero-fill launch
        self._single_step = gradient_steps == 1 and self.fused_learner and self.ent_coef_optimizer is not None
        if not self._single_step:
            self._loss_sum_buf.zero_()
        for gradient_step in range(gradient_steps):
            self._gradient_step(batch_size, gradient_step)

    def _train_host_only(self, gradient_steps: int) -> None:
        self._n_updates += gradient_steps
        s = self._loss_sums  # device-side sums of this train() call; resolved lazily when the logger dumps
        self.logger.record("train/n_updates", self._n_updates, exclude="tensorboard")
        self.logger.record("train/ent_coef", DeviceMean(s["ent_coef"], gradient_steps))
        self.logger.record("train/actor_loss", DeviceMean(s["actor"], gradient_steps))
        self.logger.record("train/critic_loss", DeviceMean(s["critic"], gradient_steps))
        if self.ent_coef_optimizer is not None:
            self.logger.record("train/ent_coef_loss", DeviceMean(s["ent_coef_loss"], gradient_steps))

    def _gradient_step(self, batch_size: int, gradient_step: int) -> None:
        if self.fused_learner:
            return self._gradient_step_fused(batch_size, gradient_step)
        return self._gradient_step_aten(batch_size, gradient_step)

    def _gradient_step_fused(self, batch_size: int, gradient_step: int) -> None:
        """The same statements as `_gradient_step_aten` (sac.py:215-287), evaluated on the fused path: losses are
        backward roots whose kernels emit d(loss)/d(inputs) directly; parameter gradients land in the arenas."""
        s, pol = self._loss_sums, self.policy
        single = getattr(self, "_single_step", False)
        acc = (lambda k: None) if single else (lambda k: s[k])  # accumulate into the sums ...
        sto = (lambda k, other: s[k]) if single else (lambda k, other: other)  # ... or store straight into them
        pb, gather = None, None
        chain = self._chain_for(batch_size)
        if chain is not None:  # the row-chain kernels (core/common/chain.py): 10 launches instead of 20
            pb = self._packed_batch(batch_size)
            gather = self.replay_buffer.take_predrawn(pb) if fused.USE_GATHER_IN_FIRST_LAYER else None
            if gather is None:
                self.replay_buffer.sample_packed_into(pb)  # :215 + the critics' cat([obs, act])
            return chain.step(self, pb, gather, gradient_step)
        if self._use_packed_batch():
            pb = self._packed_batch(batch_size)
            if fused.USE_GATHER_IN_FIRST_LAYER and self._fast_actor.pair_supported(pb):
                # indices drawn by the rollout launch: the 2B-row actor pass below gathers the rows in its first layer
                gather = self.replay_buffer.take_predrawn(pb)
            if gather is None:
                self.replay_buffer.sample_packed_into(pb)  # :215 + the critics' cat([obs, act])
            rd = pb.samples
        else:
            rd = self.replay_buffer.sample_into(self._batch(batch_size))  # :215
        B = rd.observations.shape[0]
        if not hasattr(self, "_g_bufs") or self._g_bufs[0].shape[0] != B:
            e = lambda *sh: th.empty(*sh, dtype=th.float32, device=self.device)  # noqa: E731
            self._g_bufs = (e(2, B, 1), e(B))
        gq, g_lp = self._g_bufs
        gq1, gq2 = gq[0], gq[1]

        pair = pb is not None and self._fast_actor.pair_supported(pb)
        assert gather is None or pair
        if pair:  # :222 and :247 in ONE 2B-row actor pass (three launches instead of six)
            x_pi, log_prob, x_next, next_log_prob = self._fast_actor.action_log_prob_pair(pb, gather=gather)
        elif pb is not None:  # x_pi = (obs | pi(obs)): the actor head writes the action columns of the critic input itself
            x_pi, log_prob = self._fast_actor.action_log_prob(rd.observations, xbuf=pb.x_pi.detach())  # :222
        else:
            actions_pi, log_prob = self._fast_actor.action_log_prob(rd.observations)  # :222

        # :230-261. ONE launch for the three batch reductions between the forward passes and the critic backward: the entropy
        # coefficient's loss (ent_coef = exp(log_ent_coef) BEFORE the update, :230; the updated value is first used by the
        # next gradient step, so its optimiser step may wait for the critic's all-reduce: one collective instead of two),
        # the TD target (:245-254) and the critic loss (:258-261)
        twin_pair = pb is not None and fused.twin_pair_supported(self._fast_critic, self._fast_critic_target)
        with th.no_grad():
            if not pair and pb is not None:
                x_next, next_log_prob = self._fast_actor.action_log_prob(rd.next_observations, train_params=False, xbuf=pb.x_next)
            if pb is None:
                next_actions, next_log_prob = self._fast_actor.action_log_prob(rd.next_observations, train_params=False)
                q1_t, q2_t = self._fast_critic_target(rd.next_observations, next_actions, train_params=False)
            elif not twin_pair:
                q1_t, q2_t = self._fast_critic_target.forward_input(x_next, train_params=False)
        if twin_pair:  # :258 and :250 as ONE four-network chain (three launches instead of six)
            qs, (q1_t, q2_t) = fused.twin_pair_forward(self._fast_critic, self._fast_critic_target, pb.x_data, x_next)
        else:
            qs = self._fast_critic.forward_input(pb.x_data) if pb is not None else self._fast_critic(rd.observations, rd.actions)  # :258
        q1, q2 = qs
        if self.ent_coef_optimizer is not None:
            ent_coef = s["ent_coef"] if single else self._ent_coef_buf
            alpha = dict(log_alpha=self.log_ent_coef.detach(), logp_pi=log_prob.detach(), target_entropy=self.target_entropy,
                         grad_out=self._ent_arena.grad[0:1], ent_coef_out=ent_coef, loss_out=s["ent_coef_loss"] if single else None,
                         loss_sum=acc("ent_coef_loss"), ent_coef_sum=acc("ent_coef"))
        else:
            ent_coef, alpha = self.ent_coef_tensor.reshape(1), None
            s["ent_coef"] += ent_coef
        root = qs.stacked is not None and B <= fused.LOSS_ROOT_MAX_ROWS and fused.loss_root_supported(self._fast_critic)
        td_root = None
        if root:  # the three reductions ride in the critic backward's first launch (cstr_hidden_head_bwd_root_f32)
            td_root = dict(mode="td", q1_t=q1_t, q2_t=q2_t, next_logp=next_log_prob, rew=rd.rewards, done=rd.dones,
                           ent_coef=ent_coef, gamma=self.gamma, scale=0.5, q1=q1.detach(), q2=q2.detach(), target_out=self._target_q,
                           loss_out=sto("critic", self._loss_now["critic"]), loss_sum=acc("critic"), alpha=alpha)
        else:
            hip_ops.td_twin_q_loss(q1_t, q2_t, next_log_prob, rd.rewards, rd.dones, ent_coef, self.gamma, q1, q2, 0.5, self._target_q,
                                   gq1, gq2, sto("critic", self._loss_now["critic"]), acc("critic"), alpha=alpha)
            if self.ent_coef_optimizer is not None and not self._ent_rides_critic:
                self._allreduce_grads(self._ent_arena)
                self.ent_coef_optimizer.step()
        with fused.loss_root(td_root):
            fused.backward_q(qs, gq)  # :266-268
        if root:
            if self.ent_coef_optimizer is not None and not self._ent_rides_critic:
                self._allreduce_grads(self._ent_arena)
                self.ent_coef_optimizer.step()
        self._allreduce_grads(pol.critic_arena)
        if self.ent_coef_optimizer is not None and self._ent_rides_critic:
            # :240-243 (gradient averaged by the critic's collective) and :266-268 in one launch
            self.critic.optimizer.step_with(self.ent_coef_optimizer)
        else:
            self.critic.optimizer.step()

        # :273-275 (critic weights frozen)
        qs_pi = (self._fast_critic.forward_input(x_pi, train_params=False) if pb is not None
                 else self._fast_critic(rd.observations, actions_pi, train_params=False))
        q1_pi, q2_pi = qs_pi
        root = qs_pi.stacked is not None and B <= fused.LOSS_ROOT_MAX_ROWS and fused.loss_root_supported(self._fast_critic)
        actor_root = None
        if root:  # the actor loss rides in the first launch of the backward through the (frozen) critic
            actor_root = dict(mode="sac_actor", logp=log_prob.detach(), q1=q1_pi.detach(), q2=q2_pi.detach(), ent_coef=ent_coef, g_logp=g_lp,
                              loss_out=sto("actor", self._loss_now["actor"]), loss_sum=acc("actor"))
        else:
            hip_ops.sac_actor_loss(log_prob, q1_pi, q2_pi, ent_coef, g_lp, gq1, gq2, sto("actor", self._loss_now["actor"]), acc("actor"))
        with fused.loss_root(actor_root), fused.deferred_weight_grads():  # :279-281; the actor's dW / db of all layers in one launch
            if qs_pi.stacked is not None:
                th.autograd.backward([log_prob, qs_pi.stacked], [g_lp, gq])
            else:
                th.autograd.backward([log_prob, q1_pi, q2_pi], [g_lp, gq1, gq2])
        self._allreduce_grads(pol.actor_arena)
        if gradient_step % self.target_update_interval == 0:  # :281 and :284-287 (disjoint arenas) in one launch
            self.actor.optimizer.step_with(polyak=(pol.critic_arena, pol.critic_target_arena, self.tau))
        else:
            self.actor.optimizer.step()

        if self.debug_capture:
            self.last_train_tensors = dict(target_q=self._target_q.clone(), current_q=[q1.detach().clone(), q2.detach().clone()],
                                           critic_loss=sto("critic", self._loss_now["critic"]).clone(),
                                           actor_loss=sto("actor", self._loss_now["actor"]).clone(),
                                           ent_coef=ent_coef.detach().clone(), log_prob=log_prob.detach().clone())

    def _chain_for(self, batch_size: int):
        """The row-chain form of the gradient step for this batch size (core/common/chain.py), or None: per-layer fused path."""
        from core.common import chain

        cache = self.__dict__.setdefault("_chain_cache", {})
        key = (batch_size, chain.USE_CHAIN, fused.USE_FUSED_LINEAR)
        if key not in cache:
            cache[key] = chain.SacChain(self, batch_size) if chain.SacChain.supported(self, batch_size) else None
        if cache[key] is not None and len(self.actor.action_dist.eps_queue) not in (0, 2):
            return None
        return cache[key]

    def _gradient_step_aten(self, batch_size: int, gradient_step: int) -> None:
        """Stock-ATen evaluation of the step (nn.Module forwards, autograd losses): the fallback for configurations the
        fused path does not cover (custom activations / optimisers / n_critics) and the A/B reference in tests."""
        s = self._loss_sums
        replay_data = self.replay_buffer.sample_into(self._batch(batch_size))  # :215

        actions_pi, log_prob = self.actor.action_log_prob(replay_data.observations)  # :222
        log_prob = log_prob.reshape(-1, 1)

        if self.ent_coef_optimizer is not None and self.log_ent_coef is not None:
            ent_coef = th.exp(self.log_ent_coef.detach())  # :230
            ent_coef_loss = -(self.log_ent_coef * (log_prob + self.target_entropy).detach()).mean()
            s["ent_coef_loss"] += ent_coef_loss.detach()
            self.ent_coef_optimizer.zero_grad()  # :240-243
            ent_coef_loss.backward()
            self._allreduce_grads(self._ent_arena)
            self.ent_coef_optimizer.step()
        else:
            ent_coef = self.ent_coef_tensor.reshape(1)
        s["ent_coef"] += ent_coef.reshape(1)

        with th.no_grad():  # :245-254
            next_actions, next_log_prob = self.actor.action_log_prob(replay_data.next_observations)
            q1_t, q2_t = self.critic_target(replay_data.next_observations, next_actions)[:2]
            if len(self.critic_target.q_networks) != 2:
                raise NotImplementedError("td_target_min kernel is built for n_critics=2 (the reference default)")
            target_q_values = self._target_q
            hip_ops.td_target_min(q1_t.contiguous(), q2_t.contiguous(), next_log_prob.reshape(-1, 1).contiguous(),
                                  replay_data.rewards, replay_data.dones, ent_coef.contiguous(), self.gamma, target_q_values)

        current_q_values = self.critic(replay_data.observations, replay_data.actions)  # :258
        critic_loss = 0.5 * sum(F.mse_loss(current_q, target_q_values) for current_q in current_q_values)  # :261
        s["critic"] += critic_loss.detach()
        self.critic.optimizer.zero_grad()  # :266-268
        critic_loss.backward()
        self._allreduce_grads(self.policy.critic_arena)
        self.critic.optimizer.step()

        q_values_pi = th.cat(self.critic(replay_data.observations, actions_pi), dim=1)  # :273-275
        min_qf_pi, _ = th.min(q_values_pi, dim=1, keepdim=True)
        actor_loss = (ent_coef * log_prob - min_qf_pi).mean()
        s["actor"] += actor_loss.detach()
        self.actor.optimizer.zero_grad()  # :279-281
        actor_loss.backward()
        self._allreduce_grads(self.policy.actor_arena)
        self.actor.optimizer.step()

        if gradient_step % self.target_update_interval == 0:  # :284-287 (no batch-norm stats in MLPs)
            self.policy.critic_target_arena.polyak_from(self.policy.critic_arena, self.tau)

        if self.debug_capture:
            self.last_train_tensors = dict(target_q=target_q_values.clone(), current_q=[q.detach().clone() for q in current_q_values],
                                           critic_loss=critic_loss.detach().clone(), actor_loss=actor_loss.detach().clone(),
                                           ent_coef=ent_coef.detach().clone(), log_prob=log_prob.detach().clone())

    def _get_torch_save_params(self) -> tuple:
        """reference: sac.py:319-326"""
        state_dicts = ["policy", "actor.optimizer", "critic.optimizer"]
        if self.ent_coef_optimizer is not None:
            return state_dicts + ["ent_coef_optimizer"], ["log_ent_coef"]
        return state_dicts, ["ent_coef_tensor"]

    def _extra_save_data(self) -> dict:
        return dict(ent_coef=self.ent_coef, target_update_interval=self.target_update_interval, target_entropy=self.target_entropy)

    @classmethod
    def _ctor_keys(cls) -> tuple:
        return super()._ctor_keys() + ("ent_coef", "target_update_interval", "target_entropy")

    def learn(self, total_timesteps: int, callback=None, log_interval: int = 4, tb_log_name: str = "SAC",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        return super().learn(total_timesteps=total_timesteps, callback=callback, log_interval=log_interval,
                             tb_log_name=tb_log_name, reset_num_timesteps=reset_num_timesteps, progress_bar=progress_bar)
