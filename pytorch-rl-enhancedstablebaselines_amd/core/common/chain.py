"""SAC.train's gradient step on the row-chain kernels (csrc/cstr_chain.hip): 10 launches instead of 20.

    reference statement (core/sac/sac.py)                              launch
    :215 sample (gather) + :222 pi(obs) + :247 pi(next_obs)             cstr_sac_actor_chain_fwd_f32      (head left as partial sums)
    :250 target critics + :258 critics (actor head finalised inside)    cstr_q_chain_fwd_f32, 4 networks  (Q heads left as partial sums)
    :230-261 entropy-coefficient loss, TD target, critic loss, critic
             backward to dz                                             cstr_q_chain_bwd_f32, mode 1
    :266-267 critic dW / db (6 layers)                                  cstr_linear_bwd_weight_sets_f32
    :240-243, :268 entropy-coefficient + critic Adam steps              cstr_adam_multi_f32
    :273 critics on (obs, pi(obs))                                      cstr_q_chain_fwd_f32, 2 networks
    :275 actor loss + backward through the frozen critics to the action cstr_q_chain_bwd_f32, mode 2      (action gradient as partial sums)
    :279-280 backward of the squashed-Gaussian head and the actor       cstr_sac_actor_chain_bwd_f32
    :280 actor dW / db (3 layers)                                       cstr_linear_bwd_weight_sets_f32
    :281 actor Adam step + :284-287 soft update of the target critics   cstr_adam_multi_f32

The data-parallel all-reduces keep their places (between the dW / db launch and the Adam launch of each arena). Shapes the chain
kernels do not cover (other activations, widths that are not multiples of 4 or above 512, batches that are not multiples of 16, a
VecNormalize normaliser, n_critics != 2) stay on the per-layer fused path (core/common/fused.py). CSTR_CHAIN=0 turns this path off.
"""
import os
from typing import Optional

import torch as th
from torch import nn

from core import _native as nv
from core.common import fused, hip_ops

USE_CHAIN = os.environ.get("CSTR_CHAIN", "1") != "0"
# single GPU: the dW / db launch applies the Adam step to the tiles it has reduced (cstr_linear_bwd_weight_adam_sets_f32): 8 launches
USE_WGRAD_ADAM = os.environ.get("CSTR_WGRAD_ADAM", "1") != "0"
# 16-column MFMA tiles per workgroup: actor forward, Q forward (4 networks), Q forward (2 networks), Q backward, actor backward
TILES = tuple(int(v) for v in os.environ.get("CSTR_CHAIN_TILES", "2,2,2,2,1").split(","))


# TD3's class-default nets are [400, 300]; A/B on MI355X (bench --algo td3, exact-width kernels, profiles/r03_chain_tiles_sweep_td3.txt):
# 2,2,2,2,2 0.0705 ms, 2,2,2,2,1 0.0712, 2,4,2,2,1 0.0781, one column group everywhere 0.1037
TD3_TILES = tuple(int(v) for v in os.environ.get("CSTR_CHAIN_TILES_TD3", "2,2,2,2,2").split(","))


def _q_layers(qnet: nn.Sequential):
    lin = [m for m in qnet if isinstance(m, nn.Linear)]
    return tuple((m.weight, m.bias) for m in lin)


class SacChain:
    """Buffers + launch sequence of one SAC gradient step on the chain kernels. Built once per (model, batch size)."""

    @staticmethod
    def supported(model, batch_size: int) -> bool:
        if not (USE_CHAIN and fused.USE_FUSED_LINEAR and model.fused_learner and model._use_packed_batch()):
            return False
        fa = model._fast_actor
        layers = fa.latent.layers
        if len(layers) != 2 or any(act != fused.ACT_RELU for _, act in layers) or fa.head is None or not fa._hw.is_contiguous():
            return False
        for q in list(model.critic.q_networks) + list(model.critic_target.q_networks):
            mods = list(q)
            if len(mods) != 5 or not all(isinstance(mods[i], nn.Linear) for i in (0, 2, 4)) or not all(isinstance(mods[i], nn.ReLU) for i in (1, 3)):
                return False
            if mods[4].out_features != 1:
                return False
        (l1, _), (l2, _) = layers
        c1, c2 = model.critic.q_networks[0][0], model.critic.q_networks[0][2]
        d, a = l1.in_features, fa.act_dim
        if (d, a) not in hip_ops.LAYOUTS or c1.in_features != d + a:
            return False
        return (hip_ops.chain_supported(l1.out_features, l2.out_features, batch_size) and hip_ops.chain_supported(c1.out_features, c2.out_features, batch_size)
                and all(p.grad is not None for p in list(model.actor.parameters()) + list(model.critic.parameters())))

    def __init__(self, model, batch_size: int):
        fa, dev, B = model._fast_actor, model.device, batch_size
        (l1, _), (l2, _) = fa.latent.layers
        self.D, self.A, self.B = l1.in_features, fa.act_dim, B
        self.W = self.D + self.A
        self.aH1, self.aH2 = l1.out_features, l2.out_features
        c = model.critic.q_networks[0]
        self.cH1, self.cH2 = c[0].out_features, c[2].out_features
        t_act, t_q4, t_q2, t_qb, t_ab = TILES  # (a wave's share of the reduction must fit in registers: fewer tiles for wide layers)
        self.t_act, self.t_q4, self.t_q2 = _pick_tiles(self.aH1, t_act, True), _pick_tiles(self.cH1, t_q4, True), _pick_tiles(self.cH1, t_q2, True)
        self.t_qb, self.t_ab = _pick_tiles(self.cH2, t_qb), _pick_tiles(self.aH2, t_ab)
        self.actor = hip_ops.sac_actor_desc(self.D, self.A, l1.weight, l1.bias, l2.weight, l2.bias, fa._hw, fa._hb)
        self.actor_layers = (l1, l2)
        self.crit = [_q_layers(q) for q in model.critic.q_networks]
        self.targ = [_q_layers(q) for q in model.critic_target.q_networks]
        e = lambda *sh: th.empty(*sh, dtype=th.float32, device=dev)  # noqa: E731
        A, H1, H2 = self.A, self.aH1, self.aH2
        self.n_head_parts = hip_ops.chain_colgroups(H2, self.t_act)
        self.a_h1, self.a_h2, self.head_part = e(B, H1), e(B, H2), e(self.n_head_parts, 2 * B, 2 * A)
        self.params, self.eps_all, self.logp_pi, self.logp_next = e(B, 2 * A), e(2 * B, A), e(B), e(B)
        self.c_h1, self.c_h2 = e(2, B, self.cH1), e(2, B, self.cH2)
        self.n_q4, self.n_q2 = hip_ops.chain_colgroups(self.cH2, self.t_q4), hip_ops.chain_colgroups(self.cH2, self.t_q2)
        self.q_part4, self.q_part2 = e(4, self.n_q4, B), e(2, self.n_q2, B)
        self.q_out, self.qpi_out, self.gq = e(2, B), e(2, B), e(2, B)
        self.dz2c, self.dz1c = e(2, B, self.cH2), e(2, B, self.cH1)
        self.n_gact = hip_ops.chain_colgroups(self.cH1, self.t_qb)
        self.gact_part = e(2, self.n_gact, B, A)
        self.g_params, self.dz2a, self.dz1a = e(B, 2 * A), e(B, H2), e(B, H1)
        # the merged (mu | log_std) head as (values, gradient, exp_avg, exp_avg_sq) views for the fused dW + Adam launch
        self._head_params = None
        aopt = model.actor.optimizer
        off = getattr(getattr(aopt, "arena", None), "offset_of", {})
        mu, ls = fa.actor.mu, fa.actor.log_std
        if id(mu.weight) in off and id(mu.bias) in off and hasattr(aopt, "exp_avg"):
            ow, ob, nw, nb = off[id(mu.weight)], off[id(mu.bias)], 2 * A * H2, 2 * A
            if off.get(id(ls.weight)) == ow + A * H2 and off.get(id(ls.bias)) == ob + A:
                self._head_params = ((fa._hw, fa._hwg, aopt.exp_avg[ow:ow + nw], aopt.exp_avg_sq[ow:ow + nw]),
                                     (fa._hb, fa._hbg, aopt.exp_avg[ob:ob + nb], aopt.exp_avg_sq[ob:ob + nb]))

    def step(self, model, pb, gather, gradient_step: int) -> None:
        s, pol, B, W, D, A = model._loss_sums, model.policy, self.B, self.W, self.D, self.A
        fa = model._fast_actor
        single = getattr(model, "_single_step", False)
        acc = (lambda k: None) if single else (lambda k: s[k])
        sto = (lambda k, other: s[k]) if single else (lambda k, other: other)
        rd = pb.samples
        dist = fa.actor.action_dist
        eps2 = None
        if dist.eps_queue:  # teacher-forced draws (tests): the pi(obs) tensor was queued first
            eps2 = th.cat((dist.draw_eps((B, A), model.device), dist.draw_eps((B, A), model.device)), dim=0).contiguous()
        elif fa.rng_ctl is None:
            fa.rng_ctl = hip_ops.new_rng_ctl(th.initial_seed(), model.device)
        # -- pi(obs) and pi(next_obs): gather + layers 1, 2 + head partials
        # Philox offset bookkeeping: the rollout launch in front of this step leaves its advance (n_envs draws) to the launches behind it;
        # the actor launch only READS the offset (its noise lanes add what is pending), the Q backward launch's loss workgroup advances it
        pending_ctl, pending = (None, 0)
        if gather is not None and gather[2] is not None:
            pending_ctl, pending = gather[2]
        if pending_ctl is not None and eps2 is None and pending_ctl.data_ptr() != fa.rng_ctl.data_ptr():
            raise RuntimeError("the rollout launch and the sampling head must share one Philox stream")
        noise = {} if eps2 is not None else dict(head_rng_ctl=fa.rng_ctl, head_rng_offset=pending, eps_all=self.eps_all)
        eps = self.eps_all if eps2 is None else eps2
        rng_total = None
        if eps2 is None:
            rng_total = (fa.rng_ctl, pending + 2 * B)
        elif pending_ctl is not None:
            rng_total = (pending_ctl, pending)
        if gather is not None:
            ring, idx, _, _ = gather
            hip_ops.sac_actor_chain_fwd(self.actor, B, pb.x_data, pb.x_pi, pb.x_next, rd.dones, rd.rewards, self.a_h1, self.a_h2, self.head_part,
                                        self.t_act, ring=ring, sample_idx=idx, advance_ring=True, **noise)
        else:
            hip_ops.sac_actor_chain_fwd(self.actor, B, None, pb.x_pi, pb.x_next, None, None, self.a_h1, self.a_h2, self.head_part, self.t_act, **noise)
        # -- critics on x_data, target critics on x_next (its action columns finalised inside the launch)
        fin = nv.SacHeadFin(self.head_part.data_ptr(), fa._hb.data_ptr(), eps.data_ptr(), self.n_head_parts, A, D, nv.CHAIN_HEAD_GAUSSIAN, 2 * B, B, 0.0, 0.0,
                            pb.x_pi.data_ptr(), pb.x_next.data_ptr(), self.params.data_ptr(), self.logp_pi.data_ptr(), self.logp_next.data_ptr())
        self._keep = eps2  # alive until the launches that read it have been issued (and recorded)
        nets4 = [hip_ops.chain_net(self.crit[0], pb.x_data, self.c_h1[0], self.c_h2[0], self.q_part4[0], nv.CHAIN_ROLE_STORE_PI),
                 hip_ops.chain_net(self.crit[1], pb.x_data, self.c_h1[1], self.c_h2[1], self.q_part4[1], nv.CHAIN_ROLE_PLAIN),
                 hip_ops.chain_net(self.targ[0], pb.x_next, None, None, self.q_part4[2], nv.CHAIN_ROLE_NEXT_STORE),
                 hip_ops.chain_net(self.targ[1], pb.x_next, None, None, self.q_part4[3], nv.CHAIN_ROLE_NEXT)]
        hip_ops.q_chain_fwd(nets4, W, D, self.cH1, self.cH2, B, self.t_q4, fin)
        # -- entropy-coefficient loss, TD target, critic loss and the critic backward down to dz1
        if model.ent_coef_optimizer is not None:
            ent_coef = s["ent_coef"] if single else model._ent_coef_buf
            alpha = dict(log_alpha=model.log_ent_coef.detach(), logp_pi=self.logp_pi, target_entropy=model.target_entropy,
                         grad_out=model._ent_arena.grad[0:1], ent_coef_out=ent_coef, loss_out=s["ent_coef_loss"] if single else None,
                         loss_sum=acc("ent_coef_loss"), ent_coef_sum=acc("ent_coef"))
        else:
            ent_coef, alpha = model.ent_coef_tensor.reshape(1), None
            s["ent_coef"] += ent_coef
        b3s = [self.crit[0][2][1], self.crit[1][2][1], self.targ[0][2][1], self.targ[1][2][1]]
        # one GPU: no collective between a gradient and its optimiser step -> the dW / db launch applies Adam to its tiles; the step
        # counters are advanced by the loss workgroup of the launch in front of it
        fuse_opt = USE_WGRAD_ADAM and model.world_size == 1 and B > 32 and not getattr(model, "_force_segment_boundaries", False)
        ent_opt = model.ent_coef_optimizer
        root = hip_ops.chain_root("td", B, [self.q_part4[g] for g in range(4)], b3s, self.n_q4, gamma=model.gamma, scale=0.5,
                                  next_logp=self.logp_next, rew=rd.rewards, done=rd.dones, ent_coef=ent_coef, target_out=model._target_q,
                                  q_out=self.q_out, gq_out=self.gq, loss_out=sto("critic", model._loss_now["critic"]), loss_sum=acc("critic"),
                                  alpha=alpha, rng_advance=rng_total,
                                  adam_advance=([model.critic.optimizer] + ([ent_opt] if ent_opt is not None else [])) if fuse_opt else ())
        back = [hip_ops.chain_net(self.crit[g], None, self.c_h1[g], self.c_h2[g]) for g in range(2)]
        hip_ops.q_chain_bwd(back, root, W, D, self.cH1, self.cH2, self.t_qb, dz2=self.dz2c, dz1=self.dz1c)
        if fuse_opt:
            sets = []
            for g in range(2):
                (w1, b1), (w2, b2), (w3, b3) = self.crit[g]
                sets += [(self.dz1c[g], pb.x_data, w1, b1, 0, None), (self.dz2c[g], self.c_h1[g], w2, b2, 0, None),
                         (self.gq[g].view(B, 1), self.c_h2[g], w3, b3, 0, None)]
            hip_ops.linear_bwd_weight_adam_sets(sets, [model.critic.optimizer], [ent_opt._segment()] if ent_opt is not None else [])
        else:
            sets = []
            for g in range(2):
                (w1, b1), (w2, b2), (w3, b3) = self.crit[g]
                sets += [(self.dz1c[g], pb.x_data, w1.grad, b1.grad), (self.dz2c[g], self.c_h1[g], w2.grad, b2.grad),
                         (self.gq[g].view(B, 1), self.c_h2[g], w3.grad, b3.grad)]
            hip_ops.linear_bwd_weight_sets(sets)
            if model.ent_coef_optimizer is not None and not model._ent_rides_critic:
                model._allreduce_grads(model._ent_arena)
                model.ent_coef_optimizer.step()
            model._allreduce_grads(pol.critic_arena)
            if model.ent_coef_optimizer is not None and model._ent_rides_critic:
                model.critic.optimizer.step_with(model.ent_coef_optimizer)
            else:
                model.critic.optimizer.step()
        # -- actor loss through the (updated, frozen) critics
        nets2 = [hip_ops.chain_net(self.crit[g], pb.x_pi, self.c_h1[g], self.c_h2[g], self.q_part2[g]) for g in range(2)]
        hip_ops.q_chain_fwd(nets2, W, D, self.cH1, self.cH2, B, self.t_q2)
        aroot = hip_ops.chain_root("sac_actor", B, [self.q_part2[0], self.q_part2[1]], b3s[:2], self.n_q2, ent_coef=ent_coef, logp=self.logp_pi,
                                   q_out=self.qpi_out, loss_out=sto("actor", model._loss_now["actor"]), loss_sum=acc("actor"),
                                   adam_advance=[model.actor.optimizer] if fuse_opt else ())
        hip_ops.q_chain_bwd(back, aroot, W, D, self.cH1, self.cH2, self.t_qb, gact_part=self.gact_part)
        hip_ops.sac_actor_chain_bwd(self.actor, self.gact_part, 2, self.n_gact, ent_coef, pb.x_pi, self.params, eps, self.a_h1, self.a_h2,
                                    self.g_params, self.dz2a, self.dz1a, B, self.t_ab)
        l1, l2 = self.actor_layers
        soft = gradient_step % model.target_update_interval == 0
        if fuse_opt and self._head_params is not None:
            aopt = model.actor.optimizer
            sh = aopt.shadow
            hw_p, hb_p = self._head_params
            sets = [(self.dz1a, pb.x_pi[:, :D], l1.weight, l1.bias, 0, sh[0] if sh is not None and sh[4] is l1.weight else None),
                    (self.dz2a, self.a_h1, l2.weight, l2.bias, 0, sh[0] if sh is not None and sh[4] is l2.weight else None),
                    (self.g_params, self.a_h2, hw_p, hb_p, 0, None)]
            if not pol.critic_target_arena.same_layout(pol.critic_arena):
                raise ValueError("Iterables have different lengths")  # zip_strict's error (utils.py:447)
            flat = [("polyak", pol.critic_arena.flat, pol.critic_target_arena.flat, model.tau)] if soft else []
            hip_ops.linear_bwd_weight_adam_sets(sets, [aopt], flat)  # :279-281 and :284-287
        else:
            hip_ops.linear_bwd_weight_sets([(self.dz1a, pb.x_pi[:, :D], l1.weight.grad, l1.bias.grad), (self.dz2a, self.a_h1, l2.weight.grad, l2.bias.grad),
                                            (self.g_params, self.a_h2, fa._hwg, fa._hbg)])
            model._allreduce_grads(pol.actor_arena)
            if soft:  # :281 and :284-287 (disjoint arenas) in one launch
                model.actor.optimizer.step_with(polyak=(pol.critic_arena, pol.critic_target_arena, model.tau))
            else:
                model.actor.optimizer.step()
        if model.debug_capture:
            model.last_train_tensors = dict(target_q=model._target_q.clone(), current_q=[self.q_out[0].clone().view(B, 1), self.q_out[1].clone().view(B, 1)],
                                            critic_loss=sto("critic", model._loss_now["critic"]).clone(),
                                            actor_loss=sto("actor", model._loss_now["actor"]).clone(),
                                            ent_coef=ent_coef.detach().clone(), log_prob=self.logp_pi.clone())


def _pick_tiles(kdim: int, want: int, forward: bool = False) -> int:
    """The largest tile count <= `want` whose per-wave share of a K = kdim reduction fits in registers."""
    for t in (4, 2, 1):
        if t <= want and hip_ops.chain_tiles_ok(kdim, t, forward):
            return t
    return 1


class Td3Chain:
    """TD3.train's gradient step (core/td3/td3.py:154-211) on the chain kernels:

        critic step (every update)     target actor chain (next_obs rows, head as partial sums) -> Q chain, 4 networks (target actions +
                                       smoothing noise finalised inside) -> Q backward chain (TD target + critic loss inside) -> dW / db +
                                       Adam: 4 launches
        policy step (every 2nd update) actor chain (obs rows) -> Q chain, first critic on (obs, pi(obs)) -> Q backward chain (-mean(Q1))
                                       to the action -> actor backward chain -> dW / db + Adam + the soft updates of both targets: 5 launches
    Twin critics and deterministic 3-Linear actors only (DDPG's single critic stays on the per-layer path)."""

    @staticmethod
    def supported(model, batch_size: int) -> bool:
        if not (USE_CHAIN and fused.USE_FUSED_LINEAR and model.fused_learner and model._use_packed_batch()):
            return False
        if len(model.critic.q_networks) != 2:
            return False
        for q in list(model.critic.q_networks) + list(model.critic_target.q_networks):
            mods = list(q)
            if len(mods) != 5 or not all(isinstance(mods[i], nn.Linear) for i in (0, 2, 4)) or not all(isinstance(mods[i], nn.ReLU) for i in (1, 3)):
                return False
            if mods[4].out_features != 1:
                return False
        for actor in (model.actor, model.actor_target):
            mods = list(actor.mu)
            if (len(mods) != 6 or not all(isinstance(mods[i], nn.Linear) for i in (0, 2, 4)) or not all(isinstance(mods[i], nn.ReLU) for i in (1, 3))
                    or not isinstance(mods[5], nn.Tanh)):
                return False
        a1, a2, a3 = (model.actor.mu[i] for i in (0, 2, 4))
        c1, c2 = model.critic.q_networks[0][0], model.critic.q_networks[0][2]
        d, a = a1.in_features, a3.out_features
        if (d, a) not in hip_ops.LAYOUTS or c1.in_features != d + a:
            return False
        return (hip_ops.chain_supported(a1.out_features, a2.out_features, batch_size) and hip_ops.chain_supported(c1.out_features, c2.out_features, batch_size)
                and all(p.grad is not None for p in list(model.actor.parameters()) + list(model.critic.parameters())))

    def __init__(self, model, batch_size: int):
        dev, B = model.device, batch_size
        a1, a2, a3 = (model.actor.mu[i] for i in (0, 2, 4))
        t1, t2, t3 = (model.actor_target.mu[i] for i in (0, 2, 4))
        self.D, self.A, self.B = a1.in_features, a3.out_features, B
        self.W = self.D + self.A
        self.aH1, self.aH2 = a1.out_features, a2.out_features
        c = model.critic.q_networks[0]
        self.cH1, self.cH2 = c[0].out_features, c[2].out_features
        t_act, t_q4, t_q2, t_qb, t_ab = TD3_TILES
        self.t_act, self.t_q4, self.t_q1 = _pick_tiles(self.aH1, t_act, True), _pick_tiles(self.cH1, t_q4, True), _pick_tiles(self.cH1, t_q2, True)
        self.t_qb, self.t_ab = _pick_tiles(self.cH2, t_qb), _pick_tiles(self.aH2, t_ab)
        self.actor = hip_ops.sac_actor_desc(self.D, self.A, a1.weight, a1.bias, a2.weight, a2.bias, a3.weight, a3.bias)
        self.tactor = hip_ops.sac_actor_desc(self.D, self.A, t1.weight, t1.bias, t2.weight, t2.bias, t3.weight, t3.bias)
        self.actor_layers, self.tactor_layers = (a1, a2, a3), (t1, t2, t3)
        self.crit = [_q_layers(q) for q in model.critic.q_networks]
        self.targ = [_q_layers(q) for q in model.critic_target.q_networks]
        e = lambda *sh: th.empty(*sh, dtype=th.float32, device=dev)  # noqa: E731
        A, H1, H2 = self.A, self.aH1, self.aH2
        self.n_head = hip_ops.chain_colgroups(H2, self.t_act)
        self.head_part = e(self.n_head, B, A)
        self.eps = e(B, A)
        self.a_h1, self.a_h2 = e(B, H1), e(B, H2)
        self.c_h1, self.c_h2 = e(2, B, self.cH1), e(2, B, self.cH2)
        self.n_q4, self.n_q1 = hip_ops.chain_colgroups(self.cH2, self.t_q4), hip_ops.chain_colgroups(self.cH2, self.t_q1)
        self.q_part4, self.q_part1 = e(4, self.n_q4, B), e(1, self.n_q1, B)
        self.q_out, self.qpi_out, self.gq = e(2, B), e(1, B), e(2, B)
        self.dz2c, self.dz1c = e(2, B, self.cH2), e(2, B, self.cH1)
        self.n_gact = hip_ops.chain_colgroups(self.cH1, self.t_qb)
        self.gact_part = e(1, self.n_gact, B, A)
        self.g_params, self.dz2a, self.dz1a = e(B, A), e(B, H2), e(B, H1)

    def step(self, model, pb, gather, n_updates: int) -> None:
        s, pol, B, W, D, A = model._loss_sums, model.policy, self.B, self.W, self.D, self.A
        rd = pb.samples
        single = getattr(model, "_single_step", False)
        c_out, c_sum = (s["critic"], None) if single else (model._loss_now["critic"], s["critic"])
        queued = model.noise_queue.pop(0).to(model.device, th.float32).contiguous() if model.noise_queue else None  # teacher-forced, scaled
        rng = None if queued is not None else model._device_rng()
        noise = {} if queued is not None else dict(head_rng_ctl=rng, eps_all=self.eps)
        eps = self.eps if queued is None else queued
        sigma = model.target_policy_noise if queued is None else 1.0
        fuse_opt = USE_WGRAD_ADAM and model.world_size == 1 and B > 32 and not getattr(model, "_force_segment_boundaries", False)
        # -- critic step: target actor on next_obs (:171), four Q networks (:173, :179), TD target + loss + backward (:174-186)
        kw = dict(rows_mode=nv.CHAIN_ROWS_NEXT, head_n=A, **noise)
        if gather is not None:
            ring, idx, pending, _ = gather
            if pending is not None:
                raise RuntimeError("a deterministic actor's rollout launch draws nothing: no Philox advance can be pending")
            hip_ops.sac_actor_chain_fwd(self.tactor, B, pb.x_data, pb.x_pi, pb.x_next, rd.dones, rd.rewards, None, None, self.head_part, self.t_act,
                                        ring=ring, sample_idx=idx, advance_ring=True, **kw)
        else:
            hip_ops.sac_actor_chain_fwd(self.tactor, B, None, None, pb.x_next, None, None, None, None, self.head_part, self.t_act, **kw)
        t3 = self.tactor_layers[2]
        fin = nv.SacHeadFin(self.head_part.data_ptr(), t3.bias.data_ptr(), eps.data_ptr(), self.n_head, A, D, nv.CHAIN_HEAD_DETERMINISTIC, B, 0,
                            float(sigma), float(model.target_noise_clip), pb.x_pi.data_ptr(), pb.x_next.data_ptr(), None, None, None)
        self._keep = queued
        nets4 = [hip_ops.chain_net(self.crit[0], pb.x_data, self.c_h1[0], self.c_h2[0], self.q_part4[0], nv.CHAIN_ROLE_PLAIN),
                 hip_ops.chain_net(self.crit[1], pb.x_data, self.c_h1[1], self.c_h2[1], self.q_part4[1], nv.CHAIN_ROLE_PLAIN),
                 hip_ops.chain_net(self.targ[0], pb.x_next, None, None, self.q_part4[2], nv.CHAIN_ROLE_NEXT_STORE),
                 hip_ops.chain_net(self.targ[1], pb.x_next, None, None, self.q_part4[3], nv.CHAIN_ROLE_NEXT)]
        hip_ops.q_chain_fwd(nets4, W, D, self.cH1, self.cH2, B, self.t_q4, fin)
        b3s = [self.crit[0][2][1], self.crit[1][2][1], self.targ[0][2][1], self.targ[1][2][1]]
        root = hip_ops.chain_root("td", B, [self.q_part4[g] for g in range(4)], b3s, self.n_q4, gamma=model.gamma, scale=1.0, rew=rd.rewards,
                                  done=rd.dones, target_out=model._target_q, q_out=self.q_out, gq_out=self.gq, loss_out=c_out, loss_sum=c_sum,
                                  rng_advance=None if queued is not None else (rng, B), adam_advance=[model.critic.optimizer] if fuse_opt else ())
        back = [hip_ops.chain_net(self.crit[g], None, self.c_h1[g], self.c_h2[g]) for g in range(2)]
        hip_ops.q_chain_bwd(back, root, W, D, self.cH1, self.cH2, self.t_qb, dz2=self.dz2c, dz1=self.dz1c)
        sets = []
        for g in range(2):
            (w1, b1), (w2, b2), (w3, b3) = self.crit[g]
            if fuse_opt:
                sets += [(self.dz1c[g], pb.x_data, w1, b1, 0, None), (self.dz2c[g], self.c_h1[g], w2, b2, 0, None),
                         (self.gq[g].view(B, 1), self.c_h2[g], w3, b3, 0, None)]
            else:
                sets += [(self.dz1c[g], pb.x_data, w1.grad, b1.grad), (self.dz2c[g], self.c_h1[g], w2.grad, b2.grad),
                         (self.gq[g].view(B, 1), self.c_h2[g], w3.grad, b3.grad)]
        if fuse_opt:
            hip_ops.linear_bwd_weight_adam_sets(sets, [model.critic.optimizer])
        else:
            hip_ops.linear_bwd_weight_sets(sets)
            model._allreduce_grads(pol.critic_arena)
            model.critic.optimizer.step()
        actor_done = False
        a_out = None
        if n_updates % model.policy_delay == 0:  # :192-206
            a_out, a_sum = (s["actor"], None) if single else (model._loss_now["actor"], s["actor"])
            a1, a2, a3 = self.actor_layers
            hip_ops.sac_actor_chain_fwd(self.actor, B, None, pb.x_pi, None, None, None, self.a_h1, self.a_h2, self.head_part, self.t_act,
                                        rows_mode=nv.CHAIN_ROWS_OBS, head_n=A)
            fin2 = nv.SacHeadFin(self.head_part.data_ptr(), a3.bias.data_ptr(), None, self.n_head, A, D, nv.CHAIN_HEAD_DETERMINISTIC, B, 0, 0.0, 0.0,
                                 pb.x_pi.data_ptr(), None, None, None, None)
            net1 = [hip_ops.chain_net(self.crit[0], pb.x_pi, self.c_h1[0], self.c_h2[0], self.q_part1[0], nv.CHAIN_ROLE_PI)]
            hip_ops.q_chain_fwd(net1, W, D, self.cH1, self.cH2, B, self.t_q1, fin2)
            aroot = hip_ops.chain_root("neg_mean", B, [self.q_part1[0]], b3s[:1], self.n_q1, q_out=self.qpi_out, loss_out=a_out, loss_sum=a_sum,
                                       adam_advance=[model.actor.optimizer] if fuse_opt else ())
            hip_ops.q_chain_bwd(back[:1], aroot, W, D, self.cH1, self.cH2, self.t_qb, gact_part=self.gact_part)
            hip_ops.sac_actor_chain_bwd(self.actor, self.gact_part, 1, self.n_gact, None, pb.x_pi, None, None, self.a_h1, self.a_h2, self.g_params,
                                        self.dz2a, self.dz1a, B, self.t_ab, kind=nv.CHAIN_HEAD_DETERMINISTIC)
            if not pol.actor_target_arena.same_layout(pol.actor_arena) or not pol.critic_target_arena.same_layout(pol.critic_arena):
                raise ValueError("Iterables have different lengths")  # zip_strict's error (utils.py:447)
            if fuse_opt:
                aopt, tau = model.actor.optimizer, model.tau
                sh = aopt.shadow
                shadow_of = lambda lin: sh[0] if sh is not None and sh[4] is lin.weight else None  # noqa: E731
                t1, t2, t3 = self.tactor_layers
                sets = [(self.dz1a, pb.x_pi[:, :D], a1.weight, a1.bias, 0, shadow_of(a1), (t1.weight.detach(), t1.bias.detach(), tau)),
                        (self.dz2a, self.a_h1, a2.weight, a2.bias, 0, shadow_of(a2), (t2.weight.detach(), t2.bias.detach(), tau)),
                        (self.g_params, self.a_h2, a3.weight, a3.bias, 0, shadow_of(a3), (t3.weight.detach(), t3.bias.detach(), tau))]
                hip_ops.linear_bwd_weight_adam_sets(sets, [aopt], [("polyak", pol.critic_arena.flat, pol.critic_target_arena.flat, tau)])  # :199-205
            else:
                hip_ops.linear_bwd_weight_sets([(self.dz1a, pb.x_pi[:, :D], a1.weight.grad, a1.bias.grad), (self.dz2a, self.a_h1, a2.weight.grad, a2.bias.grad),
                                                (self.g_params, self.a_h2, a3.weight.grad, a3.bias.grad)])
                model._allreduce_grads(pol.actor_arena)
                model.actor.optimizer.step_with(polyak=(pol.critic_arena, pol.critic_target_arena, model.tau),
                                                own_target=(pol.actor_target_arena.flat, model.tau))
            actor_done = True
        if model.debug_capture:
            model.last_train_tensors = dict(target_q=model._target_q.clone(), current_q=[self.q_out[0].clone().view(B, 1), self.q_out[1].clone().view(B, 1)],
                                            critic_loss=c_out.clone(), actor_loss=a_out.clone() if actor_done else None)


class MaddpgCriticChain:
    """The critic steps of MADDPG.train (core/maddpg/maddpg.py:146-164) on the chain kernels. Every agent's twin critic reads the SAME joint
    input (cat(all observations, all actions)) and its target the same next input, so
      * one agent's critic step = Q chain forward (critic + target, 4 networks) -> Q backward chain (TD target + loss inside) -> dW / db +
        Adam: 3 launches instead of 7;
      * a step WITHOUT a policy update = ONE forward launch for all agents' 4 x n_agents networks, a backward launch per agent and one
        dW / db + Adam launch per two agents.
    Centralised critics only (IDDPG's local critics read per-agent inputs and stay on the per-layer path)."""

    @staticmethod
    def supported(model, batch_size: int) -> bool:
        C = model.critic
        if not (USE_CHAIN and fused.USE_FUSED_LINEAR and model.fused_learner and not C.local and C.n_critics == 2 and 4 * model.n_agents <= nv.CHAIN_MAX_NETS):
            return False
        from core.common.arena import FlatAdam

        if not all(isinstance(o, FlatAdam) for o in C.optimizer_list):
            return False
        for nets in list(C.q_networks_list) + list(model.critic_target.q_networks_list):
            for q in nets:
                mods = list(q)
                if len(mods) != 5 or not all(isinstance(mods[i], nn.Linear) for i in (0, 2, 4)) or not all(isinstance(mods[i], nn.ReLU) for i in (1, 3)):
                    return False
                if mods[4].out_features != 1:
                    return False
        c1, c2 = C.q_networks_list[0][0][0], C.q_networks_list[0][0][2]
        d, a = model.observation_space.shape[0], model.action_space.shape[0]
        if (d, a) not in hip_ops.LAYOUTS or c1.in_features != d + a:
            return False
        return hip_ops.chain_supported(c1.out_features, c2.out_features, batch_size) and all(p.grad is not None for p in C.parameters())

    def __init__(self, model, batch_size: int):
        dev, B, n = model.device, batch_size, model.n_agents
        self.B, self.n = B, n
        self.D, self.A = model.observation_space.shape[0], model.action_space.shape[0]
        self.W = self.D + self.A
        c = model.critic.q_networks_list[0][0]
        self.H1, self.H2 = c[0].out_features, c[2].out_features
        t_act, t_q4, t_q2, t_qb, t_ab = TD3_TILES
        self.t_q, self.t_qb = _pick_tiles(self.H1, t_q4, True), _pick_tiles(self.H2, t_qb)
        self.crit = [[_q_layers(q) for q in nets] for nets in model.critic.q_networks_list]
        self.targ = [[_q_layers(q) for q in nets] for nets in model.critic_target.q_networks_list]
        e = lambda *sh: th.empty(*sh, dtype=th.float32, device=dev)  # noqa: E731
        self.n_q = hip_ops.chain_colgroups(self.H2, self.t_q)
        self.c_h1, self.c_h2 = e(n, 2, B, self.H1), e(n, 2, B, self.H2)
        self.q_part = e(n, 4, self.n_q, B)
        self.q_out, self.gq = e(n, 2, B), e(n, 2, B)
        self.dz2, self.dz1 = e(n, 2, B, self.H2), e(n, 2, B, self.H1)
        # the policy step's pass through agent i's FIRST critic (:174-177): forward partials, action-gradient partials and the critic-input
        # gradient the per-layer actor backward reads (observation columns stay zero)
        self.t_q1 = _pick_tiles(self.H1, t_q2, True)
        self.n_q1, self.n_gact = hip_ops.chain_colgroups(self.H2, self.t_q1), hip_ops.chain_colgroups(self.H1, self.t_qb)
        self.q_part1, self.qpi_out = e(1, self.n_q1, B), e(1, B)
        self.gact_part = e(1, self.n_gact, B, self.A)
        self.g_x = th.zeros(B, self.W, dtype=th.float32, device=dev)

    def actor_loss_grad(self, model, i: int, x_pi) -> th.Tensor:
        """-mean(Q1_i(obs, pi(obs))) (:177) and its gradient w.r.t. the critic input: Q chain forward (first critic, frozen) -> Q backward
        chain to the action (partials) -> the partials' sum into the action columns of g_x: 3 launches instead of 6."""
        B, W, D = self.B, self.W, self.D
        net = [hip_ops.chain_net(self.crit[i][0], x_pi, self.c_h1[i, 0], self.c_h2[i, 0], self.q_part1[0])]
        hip_ops.q_chain_fwd(net, W, D, self.H1, self.H2, B, self.t_q1)
        aroot = hip_ops.chain_root("neg_mean", B, [self.q_part1[0]], [self.crit[i][0][2][1]], self.n_q1, q_out=self.qpi_out, loss_out=model._loss_now,
                                   loss_sum=model._loss_sums[f"actor{i}"])
        back = [hip_ops.chain_net(self.crit[i][0], None, self.c_h1[i, 0], self.c_h2[i, 0])]
        hip_ops.q_chain_bwd(back, aroot, W, D, self.H1, self.H2, self.t_qb, gact_part=self.gact_part)
        hip_ops.chain_sum_parts(self.gact_part, self.g_x[:, D:])
        return self.g_x

    def _nets4(self, i: int, x_cur, x_next):
        return [hip_ops.chain_net(self.crit[i][0], x_cur, self.c_h1[i, 0], self.c_h2[i, 0], self.q_part[i, 0]),
                hip_ops.chain_net(self.crit[i][1], x_cur, self.c_h1[i, 1], self.c_h2[i, 1], self.q_part[i, 1]),
                hip_ops.chain_net(self.targ[i][0], x_next, None, None, self.q_part[i, 2]),
                hip_ops.chain_net(self.targ[i][1], x_next, None, None, self.q_part[i, 3])]

    def _backward(self, model, i: int, rd, fuse_opt: bool) -> None:
        B = self.B
        b3s = [self.crit[i][0][2][1], self.crit[i][1][2][1], self.targ[i][0][2][1], self.targ[i][1][2][1]]
        root = hip_ops.chain_root("td", B, [self.q_part[i, g] for g in range(4)], b3s, self.n_q, gamma=model.gamma, scale=1.0, rew=rd.rewards,
                                  done=rd.dones, target_out=model._target_q[i], q_out=self.q_out[i], gq_out=self.gq[i], loss_out=model._loss_now,
                                  loss_sum=model._loss_sums[f"critic{i}"], adam_advance=[model.critic.optimizer_list[i]] if fuse_opt else ())
        back = [hip_ops.chain_net(self.crit[i][g], None, self.c_h1[i, g], self.c_h2[i, g]) for g in range(2)]
        hip_ops.q_chain_bwd(back, root, self.W, self.D, self.H1, self.H2, self.t_qb, dz2=self.dz2[i], dz1=self.dz1[i])

    def _sets(self, i: int, x_cur, fuse_opt: bool, opt_index: int = 0) -> list:
        B, sets = self.B, []
        for g in range(2):
            (w1, b1), (w2, b2), (w3, b3) = self.crit[i][g]
            if fuse_opt:
                sets += [(self.dz1[i, g], x_cur, w1, b1, opt_index, None), (self.dz2[i, g], self.c_h1[i, g], w2, b2, opt_index, None),
                         (self.gq[i, g].view(B, 1), self.c_h2[i, g], w3, b3, opt_index, None)]
            else:
                sets += [(self.dz1[i, g], x_cur, w1.grad, b1.grad), (self.dz2[i, g], self.c_h1[i, g], w2.grad, b2.grad),
                         (self.gq[i, g].view(B, 1), self.c_h2[i, g], w3.grad, b3.grad)]
        return sets

    @staticmethod
    def _fuse_opt(model, B: int) -> bool:
        return USE_WGRAD_ADAM and model.world_size == 1 and B > 32 and not getattr(model, "_force_segment_boundaries", False)

    def captured(self, model, i: int) -> dict:
        B = self.B
        return dict(target_q=model._target_q[i].clone(), current_q=[self.q_out[i, 0].clone().view(B, 1), self.q_out[i, 1].clone().view(B, 1)],
                    critic_loss=model._loss_now.clone(), actor_loss=None)

    def critic_step(self, model, i: int, x_cur, x_next, rd) -> None:
        """Agent i's critic step (:146-164): 3 launches."""
        fuse_opt = self._fuse_opt(model, self.B)
        hip_ops.q_chain_fwd(self._nets4(i, x_cur, x_next), self.W, self.D, self.H1, self.H2, self.B, self.t_q)
        self._backward(model, i, rd, fuse_opt)
        opt = model.critic.optimizer_list[i]
        if fuse_opt:
            hip_ops.linear_bwd_weight_adam_sets(self._sets(i, x_cur, True), [opt])
        else:
            hip_ops.linear_bwd_weight_sets(self._sets(i, x_cur, False))
            model._allreduce_grads(model.policy.critic_slices[i])
            opt.step()

    def critic_steps_all(self, model, x_cur, x_next, rd, capture=None) -> None:
        """Every agent's critic step of an update WITHOUT a policy step: the agents' steps do not depend on each other (no soft update in
        between), so their 4 x n_agents networks share ONE forward launch; a backward launch per agent; a dW / db + Adam launch per two
        agents. The same kernels on the same operands as `critic_step`: bit-identical."""
        fuse_opt = self._fuse_opt(model, self.B)
        nets = [net for i in range(self.n) for net in self._nets4(i, x_cur, x_next)]
        hip_ops.q_chain_fwd(nets, self.W, self.D, self.H1, self.H2, self.B, self.t_q)
        for i in range(self.n):
            self._backward(model, i, rd, fuse_opt)
            if capture is not None:
                capture.append(self.captured(model, i))
        opts = model.critic.optimizer_list
        if fuse_opt:
            for i0 in range(0, self.n, 2):
                group = list(range(i0, min(i0 + 2, self.n)))
                sets = [st for k, i in enumerate(group) for st in self._sets(i, x_cur, True, opt_index=k)]
                hip_ops.linear_bwd_weight_adam_sets(sets, [opts[i] for i in group])
        else:
            for i0 in range(0, self.n, 2):
                group = list(range(i0, min(i0 + 2, self.n)))
                hip_ops.linear_bwd_weight_sets([st for i in group for st in self._sets(i, x_cur, False)])
            for i in range(self.n):
                model._allreduce_grads(model.policy.critic_slices[i])
            opts[0].step_with(*opts[1:])
