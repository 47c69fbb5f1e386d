"""oracle/sac_cpu.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Functional torch-fp32 CPU restatement of the reference's SAC networks and ONE `SAC.train` gradient step
(core/sac/sac.py:213-287, core/sac/policies.py:147-175, core/common/policies.py:960-987,
core/common/distributions.py:161-260, core/common/utils.py:457-481). It serves two purposes:
  * the checker for the learner's floating-point path: pinned against tests/golden/sac_train_kat_*.npz
    (outputs of the unmodified reference) in tests/test_oracle_learner.py;
  * the learner half of bench.py's `cpu_baseline` ("port": C oracle env/ring/sampler + this step on the
    host cores).
Weights are plain tensors keyed like the reference's state_dict; no product code is imported.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch as th
import torch.nn.functional as F

LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0


def init_params(obs_dim: int, act_dim: int, net_arch: List[int], seed: int) -> Dict[str, Dict[str, th.Tensor]]:
    """nn.Linear default init in the reference's construction order: actor (latent_pi.*, mu, log_std), critic
    (qf0.*, qf1.*), critic_target (fresh init, then overwritten) -- so `seed` reproduces SAC(seed=seed)'s weights."""
    th.manual_seed(seed)

    def mlp(prefix, sizes, out):
        p, last = {}, sizes[0]
        for li, w in enumerate(sizes[1:]):
            lin = th.nn.Linear(last, w)
            p[f"{prefix}{2 * li}.weight"], p[f"{prefix}{2 * li}.bias"] = lin.weight.detach().clone(), lin.bias.detach().clone()
            last = w
        if out is not None:
            li = len(sizes) - 1
            lin = th.nn.Linear(last, out)
            p[f"{prefix}{2 * li}.weight"], p[f"{prefix}{2 * li}.bias"] = lin.weight.detach().clone(), lin.bias.detach().clone()
        return p

    actor = mlp("latent_pi.", [obs_dim] + net_arch, None)
    for head in ("mu", "log_std"):
        lin = th.nn.Linear(net_arch[-1], act_dim)
        actor[f"{head}.weight"], actor[f"{head}.bias"] = lin.weight.detach().clone(), lin.bias.detach().clone()
    critic = {}
    for q in ("qf0.", "qf1."):
        critic.update(mlp(q, [obs_dim + act_dim] + net_arch, 1))
    for q in ("qf0.", "qf1."):  # critic_target's own init consumes RNG too (then load_state_dict)
        mlp(q, [obs_dim + act_dim] + net_arch, 1)
    return {"actor": actor, "critic": critic, "critic_target": {k: v.clone() for k, v in critic.items()}}


def _mlp(p: Dict[str, th.Tensor], prefix: str, x: th.Tensor, n_hidden: int, last_linear: bool) -> th.Tensor:
    for li in range(n_hidden):
        x = F.relu(F.linear(x, p[f"{prefix}{2 * li}.weight"], p[f"{prefix}{2 * li}.bias"]))
    if last_linear:
        x = F.linear(x, p[f"{prefix}{2 * n_hidden}.weight"], p[f"{prefix}{2 * n_hidden}.bias"])
    return x


def _n_hidden(p, prefix):
    return sum(1 for k in p if k.startswith(prefix) and k.endswith(".weight")) - (0 if prefix == "latent_pi." else 1)


def actor_action_log_prob(actor: Dict[str, th.Tensor], obs: th.Tensor, eps: th.Tensor):
    """sac/policies.py:147-175 + distributions.py:161-260 with the Normal eps given."""
    h = _mlp(actor, "latent_pi.", obs, _n_hidden(actor, "latent_pi."), False)
    mean = F.linear(h, actor["mu.weight"], actor["mu.bias"])
    log_std = th.clamp(F.linear(h, actor["log_std.weight"], actor["log_std.bias"]), LOG_STD_MIN, LOG_STD_MAX)
    std = log_std.exp()  # Normal(mean, log_std.exp())
    u = mean + std * eps  # rsample
    a = th.tanh(u)
    lp = (-((u - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(dim=1)
    lp = lp - th.sum(th.log(1 - a ** 2 + 1e-6), dim=1)
    return a, lp


def critic_forward(critic: Dict[str, th.Tensor], obs: th.Tensor, act: th.Tensor):
    x = th.cat([obs, act], dim=1)
    n = _n_hidden(critic, "qf0.")
    return _mlp(critic, "qf0.", x, n, True), _mlp(critic, "qf1.", x, n, True)


class SacCpu:
    """State of one reference SAC learner: parameters, three torch.optim.Adam, log_ent_coef."""

    def __init__(self, params, lr=3e-4, gamma=0.99, tau=0.005, target_entropy=-2.0, log_ent_coef=0.0):
        self.actor = {k: v.clone().requires_grad_(True) for k, v in params["actor"].items()}
        self.critic = {k: v.clone().requires_grad_(True) for k, v in params["critic"].items()}
        self.critic_target = {k: v.clone() for k, v in params["critic_target"].items()}
        self.log_ent_coef = th.tensor([log_ent_coef], dtype=th.float32).requires_grad_(True)
        self.opt_actor = th.optim.Adam(list(self.actor.values()), lr=lr)
        self.opt_critic = th.optim.Adam(list(self.critic.values()), lr=lr)
        self.opt_ent = th.optim.Adam([self.log_ent_coef], lr=lr)
        self.gamma, self.tau, self.target_entropy = gamma, tau, target_entropy

    def act(self, obs: th.Tensor) -> th.Tensor:
        with th.no_grad():
            return actor_action_log_prob(self.actor, obs, th.randn(obs.shape[0], self.actor["mu.bias"].shape[0]))[0]

    def train_step(self, obs, act, next_obs, done, rew, eps_pi: Optional[th.Tensor] = None, eps_next: Optional[th.Tensor] = None):
        """One gradient step, statement order of sac.py:215-287. Returns the tensors the golden KATs record."""
        B, A = obs.shape[0], act.shape[1]
        eps_pi = th.randn(B, A) if eps_pi is None else eps_pi
        eps_next = th.randn(B, A) if eps_next is None else eps_next
        actions_pi, log_prob = actor_action_log_prob(self.actor, obs, eps_pi)
        log_prob = log_prob.reshape(-1, 1)
        ent_coef = th.exp(self.log_ent_coef.detach())
        ent_coef_loss = -(self.log_ent_coef * (log_prob + self.target_entropy).detach()).mean()
        self.opt_ent.zero_grad()
        ent_coef_loss.backward()
        self.opt_ent.step()
        with th.no_grad():
            next_actions, next_log_prob = actor_action_log_prob(self.actor, next_obs, eps_next)
            nq = th.cat(critic_forward(self.critic_target, next_obs, next_actions), dim=1)
            nq, _ = th.min(nq, dim=1, keepdim=True)
            nq = nq - ent_coef * next_log_prob.reshape(-1, 1)
            target_q = rew + (1 - done) * self.gamma * nq
        q1, q2 = critic_forward(self.critic, obs, act)
        critic_loss = 0.5 * (F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q))
        self.opt_critic.zero_grad()
        critic_loss.backward()
        self.opt_critic.step()
        q_pi = th.cat(critic_forward(self.critic, obs, actions_pi), dim=1)
        min_q, _ = th.min(q_pi, dim=1, keepdim=True)
        actor_loss = (ent_coef * log_prob - min_q).mean()
        self.opt_actor.zero_grad()
        actor_loss.backward()
        self.opt_actor.step()
        with th.no_grad():  # polyak_update, utils.py:478-481
            for k, t in self.critic_target.items():
                t.mul_(1 - self.tau)
                th.add(t, self.critic[k].detach(), alpha=self.tau, out=t)
        return dict(target_q=target_q, current_q1=q1.detach(), current_q2=q2.detach(), critic_loss=float(critic_loss.detach()),
                    actor_loss=float(actor_loss.detach()), ent_coef_loss=float(ent_coef_loss.detach()), ent_coef=float(ent_coef))


def params_from_golden(g, prefix="before") -> Dict[str, Dict[str, th.Tensor]]:
    out: Dict[str, Dict[str, th.Tensor]] = {"actor": {}, "critic": {}, "critic_target": {}}
    for key in g.files:
        parts = key.split("/")
        if len(parts) == 3 and parts[0] == prefix and parts[1] in out and "#" not in parts[2]:
            out[parts[1]][parts[2]] = th.as_tensor(np.array(g[key]))
    return out
