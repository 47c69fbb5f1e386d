"""GPU tests of the learner glue kernels (csrc/cstr_mlp.hip) against torch autograd evaluating the reference's
own expressions (core/common/distributions.py:161-260, core/sac/sac.py:230-275, core/td3/td3.py:182-194,
core/common/torch_layers.py:110-183). fp32 tolerances: 1e-6 relative for forward values, 1e-5 for gradients
(the analytic squashed-Gaussian backward differs from autograd's by terms that cancel to rounding noise)."""
import math

import numpy as np
import pytest
import torch as th

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert th.cuda.is_available()
    from core.common import hip_ops

    return hip_ops


def ref_squashed(mean, log_std_raw, eps):
    """The reference's statements, in torch (float64 for the gradient reference)."""
    log_std = th.clamp(log_std_raw, -20, 2)
    std = log_std.exp()
    u = mean + std * eps
    a = th.tanh(u)
    lp = (-((u - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(dim=1)
    lp = lp - th.sum(th.log(1 - a ** 2 + 1e-6), dim=1)
    return a, lp


@pytest.mark.parametrize("B,A", [(1, 2), (256, 2), (4096, 2), (257, 3)])
def test_squashed_gaussian_fwd_bwd(ops, B, A):
    from core.common import fused

    g = th.Generator().manual_seed(B)
    mean = th.randn(B, A, generator=g)
    ls = th.randn(B, A, generator=g) * 2 - 1
    ls[0, 0], ls[-1, -1] = 3.0, -25.0  # outside the clamp range: zero gradient there
    eps = th.randn(B, A, generator=g)
    a_ref, lp_ref = ref_squashed(mean, ls, eps)
    m_d, l_d = mean.cuda().requires_grad_(True), ls.cuda().requires_grad_(True)
    a, lp = fused.squashed_gaussian(th.cat((m_d, l_d), dim=1), eps.cuda(), A)  # merged-head layout [mean | log_std_raw]
    assert rel_err(a.detach().cpu().numpy(), a_ref.numpy(), 1.0) < 1e-6  # tanh output lives in [-1, 1]
    # log(1 - a^2 + 1e-6) is ill-conditioned where tanh saturates (a 1-ulp difference between libm's and ocml's tanhf
    # is amplified by 1/(1 - a^2 + 1e-6)); this synthetic batch saturates heavily (std up to e^2)
    assert rel_err(lp.detach().cpu().numpy(), lp_ref.numpy(), 1.0) < 5e-5
    u_abs = (mean + th.clamp(ls, -20, 2).exp() * eps).abs().max(dim=1).values.numpy()
    assert rel_err(lp.detach().cpu().numpy()[u_abs < 3], lp_ref.numpy()[u_abs < 3], 1.0) < 5e-6
    ga, gl = th.randn(B, A, generator=g), th.randn(B, generator=g)
    th.autograd.backward([a, lp], [ga.cuda(), gl.cuda()])
    m64, l64 = mean.double().requires_grad_(True), ls.double().requires_grad_(True)
    a64, lp64 = ref_squashed(m64, l64, eps.double())
    th.autograd.backward([a64, lp64], [ga.double(), gl.double()])
    # where tanh saturates in fp32 (1 - a^2 rounds to 0) the fp32 gradient is exactly 0 -- also in the reference's
    # fp32 autograd -- while the fp64 reference keeps a tiny tail: compare away from saturation, check finiteness
    u64 = (mean.double() + th.clamp(ls.double(), -20, 2).exp() * eps.double()).abs().numpy()
    ok = u64 < 5.0
    assert ok.mean() > 0.8
    gm, gls = m_d.grad.cpu().numpy(), l_d.grad.cpu().numpy()
    assert np.isfinite(gm).all() and np.isfinite(gls).all()
    assert rel_err(gm[ok], m64.grad.numpy()[ok], 1e-2) < 3e-5
    assert rel_err(gls[ok], l64.grad.numpy()[ok], 1e-2) < 3e-5
    assert float(l_d.grad[0, 0]) == 0.0 and float(l_d.grad[-1, -1]) == 0.0
    # acting only (no logp, no grad)
    with th.no_grad():
        a2, none = fused.squashed_gaussian(th.cat((mean, ls), dim=1).cuda(), eps.cuda(), A, want_logp=False)
    assert none is None and th.equal(a2, a.detach())


@pytest.mark.parametrize("M,K,N,act", [(256, 6, 256, 1), (256, 256, 256, 1), (256, 256, 1, 0), (256, 256, 2, 0),
                                       (4096, 4, 256, 1), (64, 32, 2, 2), (7, 5, 3, 1)])
def test_fused_linear_matches_module(ops, M, K, N, act):
    """linear() forward/backward (with gradients written into arena views) == nn.Linear + activation + autograd."""
    from core.common import fused
    from core.common.arena import ParamArena

    th.manual_seed(M + N)
    lin = th.nn.Linear(K, N)
    ref = th.nn.Linear(K, N)
    ref.load_state_dict(lin.state_dict())
    ref = ref.cuda()
    arena = ParamArena(lin.parameters(), "cuda")
    x = th.randn(M, K, device="cuda", requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    fn = {0: lambda t: t, 1: th.relu, 2: th.tanh}[act]
    y_ref = fn(ref(x2))
    y = fused.linear(x, lin.weight, lin.bias, act, True)
    # two fp32 summation orders (rocBLAS tiles vs the k-ordered MFMA chain of the fused kernel): compare at the output scale
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), max(1e-2, float(y_ref.detach().abs().mean()))) < 1e-5
    gy = th.randn(M, N, device="cuda")
    arena.grad.fill_(123.0)  # stale values must be overwritten, not accumulated
    y.backward(gy)
    y_ref.backward(gy)
    for got, want in ((lin.weight.grad, ref.weight.grad), (lin.bias.grad, ref.bias.grad), (x.grad, x2.grad)):
        assert rel_err(got.cpu().numpy(), want.cpu().numpy(), max(1e-2, float(want.abs().mean()))) < 1e-5
    assert lin.weight.grad.data_ptr() == arena.grad.data_ptr()  # still the arena view
    # frozen parameters: only dx, arena untouched
    arena.grad.fill_(7.0)
    x3 = x.detach().clone().requires_grad_(True)
    fused.linear(x3, lin.weight, lin.bias, act, False).backward(gy)
    assert th.all(arena.grad == 7.0)
    assert rel_err(x3.grad.cpu().numpy(), x2.grad.cpu().numpy(), max(1e-2, float(x2.grad.abs().mean()))) < 1e-5


@pytest.mark.parametrize("B", [1, 256, 1000])
def test_target_loss_and_entropy_coefficient_in_one_launch(ops, B):
    """cstr_td_twin_q_loss_f32 == cstr_td_target_min_f32 -> cstr_twin_q_loss_f32 (-> cstr_sac_alpha_f32), bit for bit, in
    the SAC form (entropy term, alpha part riding along), the SAC fixed-coefficient form and the TD3 form."""
    g = th.Generator(device="cuda").manual_seed(B)
    r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
    q1t, q2t, q1, q2, rew = (r(B, 1) * 3 for _ in range(5))
    nlp, lp = r(B), r(B)
    done = (th.rand(B, 1, device="cuda", generator=g) < 0.2).float()
    la = th.tensor([-0.3], device="cuda")
    z = lambda: th.zeros(1, device="cuda")  # noqa: E731
    for form in ("sac", "sac_fixed", "td3"):
        # three launches
        t_ref, g1_ref, g2_ref = th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")
        loss_ref, acc_ref, grad_ref, ec_ref, al_ref, als_ref, ecs_ref = z(), z() + 10, z(), z(), z(), z() + 5, z() + 7
        if form == "sac":
            ops.sac_alpha(la, lp, -2.0, grad_ref, ec_ref, als_ref, ecs_ref, loss_out=al_ref)
        else:
            ec_ref.fill_(0.37)
        ops.td_target_min(q1t, q2t, None if form == "td3" else nlp, rew, done, None if form == "td3" else ec_ref, 0.99, t_ref)
        scale = 1.0 if form == "td3" else 0.5
        ops.twin_q_loss(q1, q2, t_ref, scale, g1_ref, g2_ref, loss_ref, acc_ref)
        # one launch
        t, g1, g2 = th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")
        loss, acc, grad, ec, al, als, ecs = z(), z() + 10, z(), z(), z(), z() + 5, z() + 7
        alpha = None
        if form == "sac":
            alpha = dict(log_alpha=la, logp_pi=lp, target_entropy=-2.0, grad_out=grad, ent_coef_out=ec, loss_out=al, loss_sum=als,
                         ent_coef_sum=ecs)
        else:
            ec.fill_(0.37)
        ops.td_twin_q_loss(q1t, q2t, None if form == "td3" else nlp, rew, done, None if form != "sac_fixed" else ec, 0.99, q1, q2,
                           scale, t, g1, g2, loss, acc, alpha=alpha)
        for got, want in ((t, t_ref), (g1, g1_ref), (g2, g2_ref), (loss, loss_ref), (acc, acc_ref), (grad, grad_ref), (ec, ec_ref),
                          (al, al_ref), (als, als_ref), (ecs, ecs_ref)):
            assert th.equal(got, want), form
        ops.td_twin_q_loss(q1t, q2t, None, rew, done, None, 0.99, q1, q2, scale, None, g1, g2)  # no target / loss outputs
    with pytest.raises(ValueError):
        ops.td_twin_q_loss(q1t, q2t, nlp, rew, done, None, 0.99, q1, q2, 0.5, None, g1, g2)  # entropy term without coefficient


def test_loss_heads(ops):
    g = th.Generator().manual_seed(0)
    for B in (1, 64, 256, 1000):
        q1, q2, t = (th.randn(B, 1, generator=g) * 3 for _ in range(3))
        lp = th.randn(B, generator=g)
        d = lambda x: x.cuda().contiguous()  # noqa: E731
        gq1, gq2, glp = th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda"), th.empty(B, device="cuda")
        loss, acc = th.zeros(1, device="cuda"), th.full((1,), 10.0, device="cuda")
        for scale in (0.5, 1.0):  # SAC sac.py:261 / TD3 td3.py:182
            a, b = q1.clone().requires_grad_(True), q2.clone().requires_grad_(True)
            ref = scale * (th.nn.functional.mse_loss(a, t) + th.nn.functional.mse_loss(b, t))
            ref.backward()
            acc.fill_(10.0)
            ops.twin_q_loss(d(q1), d(q2), d(t), scale, gq1, gq2, loss, acc)
            assert rel_err(float(loss), float(ref.detach()), 1e-3) < 2e-6 and abs(float(acc) - 10.0 - float(ref.detach())) < 1e-4
            assert rel_err(gq1.cpu().numpy(), a.grad.numpy(), 1e-4) < 2e-6 and rel_err(gq2.cpu().numpy(), b.grad.numpy(), 1e-4) < 2e-6
        # SAC actor loss sac.py:273-275
        ent = th.tensor([0.37])
        a, b, l = q1.clone().requires_grad_(True), q2.clone().requires_grad_(True), lp.clone().requires_grad_(True)
        mn, _ = th.min(th.cat((a, b), dim=1), dim=1, keepdim=True)
        ref = (ent * l.reshape(-1, 1) - mn).mean()
        ref.backward()
        ops.sac_actor_loss(d(lp), d(q1), d(q2), d(ent), glp, gq1, gq2, loss, None)
        assert rel_err(float(loss), float(ref.detach()), 1e-3) < 2e-6
        for got, want in ((glp, l.grad), (gq1, a.grad), (gq2, b.grad)):
            assert rel_err(got.cpu().numpy().reshape(-1), want.numpy().reshape(-1), 1e-4) < 2e-6
        # entropy coefficient sac.py:230-231
        la = th.tensor([-0.3], requires_grad=True)
        ref = -(la * (lp.reshape(-1, 1) - 2.0).detach()).mean()
        ref.backward()
        grad, ec, ls_, es = (th.zeros(1, device="cuda") for _ in range(4))
        ops.sac_alpha(d(la.detach()), d(lp), -2.0, grad, ec, ls_, es)
        assert rel_err(float(grad), float(la.grad), 1e-3) < 2e-6 and rel_err(float(ec), math.exp(-0.3), 1e-3) < 1e-6
        assert rel_err(float(ls_), float(ref.detach()), 1e-3) < 2e-6 and rel_err(float(es), math.exp(-0.3), 1e-3) < 1e-6
        # TD3 actor loss td3.py:194
        a = q1.clone().requires_grad_(True)
        ref = -a.mean()
        ref.backward()
        ops.neg_mean_loss(d(q1), gq1, loss, None)
        assert rel_err(float(loss), float(ref), 1e-3) < 2e-6 and rel_err(gq1.cpu().numpy(), a.grad.numpy(), 1e-6) < 1e-6


def test_fast_modules_equal_nn_modules():
    """FastSacActor / FastTwinCritic read the nn.Modules' tensors: same outputs as the module forwards."""
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    model = SAC("MlpPolicy", CSTRVecEnv(4), seed=3, policy_kwargs=dict(net_arch=[64, 64]))
    obs = th.rand(128, 4, device="cuda") * 2 - 1
    act = th.rand(128, 2, device="cuda") * 2 - 1
    eps = th.randn(128, 2, device="cuda")
    with th.no_grad():
        model.actor.action_dist.eps_queue = [eps.clone()]
        a_ref, lp_ref = model.actor.action_log_prob(obs)
        a, lp = model._fast_actor.action_log_prob(obs, eps=eps, train_params=False)
        q_ref = model.critic(obs, act)
        q = model._fast_critic(obs, act, train_params=False)
    assert rel_err(a.cpu().numpy(), a_ref.cpu().numpy(), 1e-3) < 1e-5 and rel_err(lp.cpu().numpy(), lp_ref.cpu().numpy(), 1.0) < 1e-5
    for x, y in zip(q, q_ref):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy(), 1e-2) < 1e-5


def test_stacked_twin_critic_and_merged_heads_equal_per_network_path():
    """The batched-GEMM critics / merged actor heads read the same arena memory as the nn.Modules: forward values and the
    gradients that land in the arena must equal the per-network fused path and autograd on the modules."""
    from core.common import fused
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    model = SAC("MlpPolicy", CSTRVecEnv(4), seed=1, policy_kwargs=dict(net_arch=[64, 48]))
    pol = model.policy
    assert pol.critic_stack is not None and model._fast_critic.stack is not None
    obs, act = th.rand(96, 4, device="cuda") * 2 - 1, th.rand(96, 2, device="cuda") * 2 - 1
    gq = th.randn(2, 96, 1, device="cuda")
    # stacked
    qs = model._fast_critic(obs, act)
    fused.backward_q(qs, gq)
    g_stacked = pol.critic_arena.grad.clone()
    # per-network FastMLP chains on the same parameters
    plain = fused.FastTwinCritic(model.critic, None)
    pol.critic_arena.grad.fill_(5.0)
    qp = plain(obs, act)
    fused.backward_q(qp, gq)
    for a, b in zip(qs, qp):
        assert rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy(), 1e-2) < 1e-5
    used = th.zeros_like(g_stacked, dtype=th.bool)
    for p, o in zip(pol.critic_arena.params, pol.critic_arena.offsets):
        used[o:o + p.numel()] = True
    gref = pol.critic_arena.grad[used].cpu().numpy()  # batched vs per-network GEMMs differ in summation order only
    assert rel_err(g_stacked[used].cpu().numpy(), gref, float(np.abs(gref).mean())) < 3e-5
    # nn.Module + autograd reference
    ref = [q.detach().clone() for q in model.critic(obs, act)]
    for a, b in zip(qs, ref):
        assert rel_err(a.detach().cpu().numpy(), b.cpu().numpy(), 1e-2) < 1e-5
    # actor: merged heads vs separate Linear heads
    eps = th.randn(96, 2, device="cuda")
    a1, lp1 = model._fast_actor.action_log_prob(obs, eps=eps)
    th.autograd.backward([a1, lp1], [th.ones_like(a1), th.ones_like(lp1)])
    g_merged = pol.actor_arena.grad.clone()
    sep = fused.FastSacActor(model.actor, None)
    pol.actor_arena.grad.zero_()
    a2, lp2 = sep.action_log_prob(obs, eps=eps)
    th.autograd.backward([a2, lp2], [th.ones_like(a2), th.ones_like(lp2)])
    assert rel_err(a1.detach().cpu().numpy(), a2.detach().cpu().numpy(), 1.0) < 1e-6
    assert rel_err(lp1.detach().cpu().numpy(), lp2.detach().cpu().numpy(), 1.0) < 1e-5
    gref = pol.actor_arena.grad.cpu().numpy()
    assert rel_err(g_merged.cpu().numpy(), gref, float(np.abs(gref).mean())) < 3e-5


@pytest.mark.parametrize("G,M,K,act", [(1, 256, 256, 1), (2, 256, 256, 1), (2, 256, 300, 1), (1, 5, 37, 2), (3, 70, 64, 0), (2, 33, 400, 2)])
def test_hidden_layer_plus_scalar_head_kernels(ops, G, M, K, act):
    """cstr_hidden_head_{fwd,bwd}_f32 = bias + activation of the last hidden layer fused with the Q network's Linear(K -> 1)
    head (torch_layers.py:110-183 with output_dim 1), against plain torch ops / autograd in fp64."""
    g = th.Generator(device="cuda").manual_seed(G * 1000 + K)
    z = th.randn(G, M, K, device="cuda", generator=g)
    b1, w2 = th.randn(G, K, device="cuda", generator=g) * 0.3, th.randn(G, K, device="cuda", generator=g) / K ** 0.5
    b2, gq = th.randn(G, device="cuda", generator=g), th.randn(G, M, device="cuda", generator=g)
    zd, b1d, w2d, b2d = (t.double().requires_grad_(True) for t in (z, b1, w2, b2))
    pre = zd + b1d[:, None, :]
    yd = th.relu(pre) if act == 1 else (th.tanh(pre) if act == 2 else pre)
    qd = (yd * w2d[:, None, :]).sum(-1) + b2d[:, None]
    qd.backward(gq.double())
    y, q = z.clone(), th.empty(G, M, 1, device="cuda")
    ops.hidden_head_fwd_(y, b1, act, w2, b2, q)
    assert rel_err(y.cpu().numpy(), yd.detach().cpu().numpy(), 1.0) < 1e-6
    assert rel_err(q.cpu().numpy().reshape(G, M), qd.detach().cpu().numpy(), 1.0) < 3e-6
    dz, gb1, gw2, gb2 = th.empty_like(y), th.full((G, K), 9.0, device="cuda"), th.full((G, K), 9.0, device="cuda"), th.full((G,), 9.0, device="cuda")
    ops.hidden_head_bwd(gq, y, act, w2, dz, gb1, gw2, gb2)
    assert rel_err(dz.cpu().numpy(), zd.grad.cpu().numpy(), 1.0) < 1e-6
    scale = float(M) ** 0.5
    assert rel_err(gb1.cpu().numpy(), b1d.grad.cpu().numpy(), scale) < 2e-6
    assert rel_err(gw2.cpu().numpy(), w2d.grad.cpu().numpy(), scale) < 2e-6
    assert rel_err(gb2.cpu().numpy(), b2d.grad.cpu().numpy(), scale) < 2e-6
    dz2 = th.empty_like(y)
    ops.hidden_head_bwd(gq, y, act, w2, dz2)  # frozen parameters: input gradient only
    assert th.equal(dz, dz2)
    with pytest.raises(ValueError):
        ops.hidden_head_bwd(gq, y, act, w2, dz2, gb1, None, None)
    if G == 1:  # 2-D operands
        q2 = th.empty(M, 1, device="cuda")
        y2 = z[0].clone()
        ops.hidden_head_fwd_(y2, b1[0], act, w2[0], b2, q2)
        assert th.equal(q2, q[0]) and th.equal(y2, y[0])


@pytest.mark.parametrize("M,K,act", [(256, 300, 1), (100, 64, 2), (16, 256, 0)])
def test_deterministic_actor_loss_inside_the_head_backward(ops, M, K, act):
    """cstr_hidden_head_bwd_root_f32 mode 3 (-mean(Q1) through the first Q network alone) against cstr_neg_mean_loss_f32 followed by
    cstr_hidden_head_bwd_f32 on one group: logged loss, running sum and dz bit-identical; with and without parameter gradients."""
    g = th.Generator(device="cuda").manual_seed(M + K)
    r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
    y = r(1, M, K) if act != 1 else th.relu(r(1, M, K))
    if act == 2:
        y = th.tanh(y)
    w2, q = r(1, K), r(M, 1)
    gq = th.empty(1, M, 1, device="cuda")
    l1, s1 = th.zeros(1, device="cuda"), th.full((1,), 2.5, device="cuda")
    l2, s2 = th.zeros(1, device="cuda"), th.full((1,), 2.5, device="cuda")
    dz1, dz2 = th.empty_like(y), th.empty_like(y)
    p1 = [th.empty(1, K, device="cuda"), th.empty(1, K, device="cuda"), th.empty(1, device="cuda")]
    p2 = [th.empty(1, K, device="cuda"), th.empty(1, K, device="cuda"), th.empty(1, device="cuda")]
    ops.neg_mean_loss(q, gq[0], l1, s1)
    ops.hidden_head_bwd(gq, y, act, w2, dz1, *p1)
    ops.hidden_head_bwd_root(dict(mode="neg_mean", q1=q, loss_out=l2, loss_sum=s2), y, act, w2, dz2, *p2)
    th.cuda.synchronize()
    assert th.equal(l1, l2) and th.equal(s1, s2) and th.equal(dz1, dz2)
    for a, b in zip(p1, p2):
        assert th.equal(a, b)
    assert abs(float(l2) + float(q.double().mean())) < 1e-5
    dz3 = th.empty_like(y)
    ops.hidden_head_bwd_root(dict(mode="neg_mean", q1=q, loss_out=None, loss_sum=None), y, act, w2, dz3)  # frozen parameters
    assert th.equal(dz3, dz1)
    with pytest.raises(Exception):  # two groups are the twin modes' business
        ops.hidden_head_bwd_root(dict(mode="neg_mean", q1=q, loss_out=None, loss_sum=None), th.cat([y, y]), act, th.cat([w2, w2]), th.cat([dz3, dz3]))


def test_gaussian_head_kernels_match_unfused_path_and_rng_statistics(ops):
    """cstr_gaussian_head_{fwd,bwd}_f32 = head bias + squashed-Gaussian sampling in one launch. With eps GIVEN it must equal
    bias_act_fwd + squashed_gaussian_fwd/bwd (+ the bias gradient's column sum); with the in-kernel Philox stream the noise
    must be standard normal, reproducible from (seed, offset) and advance across launches."""
    from core.common import hip_ops

    B, A = 256, 2
    g = th.Generator(device="cuda").manual_seed(3)
    z, bias = th.randn(B, 2 * A, device="cuda", generator=g), th.randn(2 * A, device="cuda", generator=g) * 0.2
    z[:, A:] = z[:, A:] * 8  # exercise the log-std clamp
    eps = th.randn(B, A, device="cuda", generator=g)
    ref_p = ops.bias_act_fwd_(z.clone(), bias, 0)
    ref_a, ref_lp = th.empty(B, A, device="cuda"), th.empty(B, device="cuda")
    ops.squashed_gaussian_fwd(ref_p[:, :A], ref_p[:, A:], eps, ref_a, ref_lp)
    p = z.clone()
    xbuf = th.full((B, 4 + A), 7.0, device="cuda")
    lp = th.empty(B, device="cuda")
    hip_ops.gaussian_head_fwd_(p, bias, eps, None, xbuf[:, 4:], lp)
    assert th.equal(p, ref_p) and th.equal(xbuf[:, 4:], ref_a) and th.equal(lp, ref_lp) and float(xbuf[:, :4].min()) == 7.0
    ga_full, glp = th.randn(B, 4 + A, device="cuda", generator=g), th.randn(B, device="cuda", generator=g)
    ref_gp = th.empty(B, 2 * A, device="cuda")
    ops.squashed_gaussian_bwd(ga_full[:, 4:].contiguous(), glp, ref_a, ref_p[:, A:], eps, ref_gp[:, :A], ref_gp[:, A:])
    gp, gb = th.empty(B, 2 * A, device="cuda"), th.empty(2 * A, device="cuda")
    hip_ops.gaussian_head_bwd(ga_full[:, 4:], glp, xbuf[:, 4:], p, eps, gp, gb)
    assert th.equal(gp, ref_gp)
    assert rel_err(gb.cpu().numpy(), ref_gp.double().sum(0).cpu().numpy(), float(B) ** 0.5) < 2e-6
    # in-kernel RNG
    n = 1 << 19
    ctl = hip_ops.new_rng_ctl(1234, "cuda")
    pz = th.zeros(n, 2 * A, device="cuda")
    e1, e2, act = th.empty(n, A, device="cuda"), th.empty(n, A, device="cuda"), th.empty(n, A, device="cuda")
    hip_ops.gaussian_head_fwd_(pz, None, e1, ctl, act, None)
    assert ctl.cpu().tolist() == [1234, n] + [0] * 14
    hip_ops.gaussian_head_fwd_(pz, None, e2, ctl, act, None)
    assert ctl.cpu().tolist() == [1234, 2 * n] + [0] * 14
    assert th.equal(act, th.tanh(e2))  # mean 0, log_std 0: action = tanh(eps)
    x = th.cat([e1, e2]).double().flatten()
    assert abs(float(x.mean())) < 4e-3 and abs(float(x.var()) - 1.0) < 6e-3
    assert abs(float((x ** 4).mean()) - 3.0) < 0.05 and abs(float((x ** 3).mean())) < 0.02
    assert abs(float((e1[:, 0] * e1[:, 1]).mean())) < 5e-3 and abs(float((e1 * e2).mean())) < 5e-3  # pairs / launches independent
    assert float(x.abs().max()) > 4.0
    ctl2 = hip_ops.new_rng_ctl(1234, "cuda")
    ctl2[1] = n
    e3 = th.empty(n, A, device="cuda")
    hip_ops.gaussian_head_fwd_(pz, None, e3, ctl2, act, None)
    assert th.equal(e3, e2)  # a pure function of (seed, offset)
    e4 = th.empty(n, A, device="cuda")
    hip_ops.gaussian_head_fwd_(pz, None, e4, hip_ops.new_rng_ctl(1235, "cuda"), act, None)
    assert abs(float((e4 * e1).mean())) < 5e-3 and not th.equal(e4, e1)


def test_target_smoothing_kernel(ops):
    """cstr_target_smooth_f32 = td3.py:167-173 (noise.clamp(-c, c); (a + noise).clamp(-1, 1)) with the noise given, and the
    same with sigma * N(0, 1) drawn in the kernel."""
    from core.common import hip_ops

    B, A, D = 300, 2, 4
    g = th.Generator(device="cuda").manual_seed(0)
    a = th.tanh(th.randn(B, A, device="cuda", generator=g) * 2)
    noise = th.randn(B, A, device="cuda", generator=g) * 0.4
    x = th.full((B, D + A), 5.0, device="cuda")
    hip_ops.target_smooth(a, noise, None, 0.2, 0.5, x[:, D:])
    assert th.equal(x[:, D:], (a + noise.clamp(-0.5, 0.5)).clamp(-1, 1)) and float(x[:, :D].min()) == 5.0
    ctl = hip_ops.new_rng_ctl(9, "cuda")
    n = 1 << 18
    big, out = th.zeros(n, A, device="cuda"), th.empty(n, A, device="cuda")
    hip_ops.target_smooth(big, None, ctl, 0.2, 0.5, out)
    assert ctl.cpu().tolist() == [9, n] + [0] * 14
    z = out.double().flatten()
    assert abs(float(z.mean())) < 2e-3 and abs(float(z.std()) - 0.2 * 0.98872) < 1e-3  # N(0, 0.2) clamped at 2.5 sigma
    assert float(z.abs().max()) == 0.5
    hip_ops.target_smooth(big, None, ctl, 0.0, 0.5, out)  # DDPG: no smoothing
    assert float(out.abs().max()) == 0.0
    with pytest.raises(ValueError):
        hip_ops.target_smooth(a, noise, ctl, 0.2, 0.5, x[:, D:])


@pytest.mark.parametrize("M,K,N,act", [(256, 300, 2, 2), (100, 64, 4, 2), (1000, 256, 1, 0), (16, 36, 16, 1)])
def test_last_layer_with_the_target_smoothing_inside(ops, M, K, N, act):
    """cstr_linear_smooth_fwd_f32 against cstr_linear_act_fwd_f32 followed by cstr_target_smooth_f32: bit-identical output (written
    into a column block of a wider matrix) and Philox control block, for given noise and for in-kernel noise."""
    g = th.Generator(device="cuda").manual_seed(M + K + N)
    r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
    x, w, b = r(M, K + 4)[:, :K], r(N, K) / K ** 0.5, r(N) * 0.1
    noise = (r(M, N) * 0.2).contiguous()
    for rng in (False, True):
        c1, c2 = ops.new_rng_ctl(9, "cuda"), ops.new_rng_ctl(9, "cuda")
        c1[1] = c2[1] = 12345
        o1, o2 = th.full((M, 3 + N), 7.0, device="cuda"), th.full((M, 3 + N), 7.0, device="cuda")
        a = ops.linear_act_fwd(x, w, b, act)
        ops.target_smooth(a, None if rng else noise, c1 if rng else None, 0.2, 0.5, o1[:, 3:])
        ops.linear_smooth_fwd(x, w, b, act, None if rng else noise, c2 if rng else None, 0.2, 0.5, o2[:, 3:])
        th.cuda.synchronize()
        assert th.equal(o1, o2) and th.equal(c1, c2) and float(o2[:, :3].min()) == 7.0
        assert int(c2[1]) == 12345 + (M if rng else 0) and float(o2[:, 3:].abs().max()) <= 1.0
    with pytest.raises(Exception):
        ops.linear_smooth_fwd(x, w, b, act, noise, ops.new_rng_ctl(1, "cuda"), 0.2, 0.5, th.empty(M, N, device="cuda"))  # two noise sources
    with pytest.raises(Exception):
        ops.linear_smooth_fwd(x[:, :16], w[:, :16].contiguous(), b, act, noise, None, 0.2, 0.5, th.empty(M, N, device="cuda"))  # k <= 32


@pytest.mark.parametrize("G,M,N,K,act", [(0, 256, 256, 256, 1), (0, 256, 256, 4, 1), (0, 4096, 256, 256, 1), (2, 256, 256, 6, 1),
                                        (2, 256, 256, 256, 1), (0, 100, 400, 10, 1), (0, 37, 300, 400, 2), (2, 33, 30, 50, 0),
                                        (0, 1, 16, 7, 2), (0, 256, 4, 256, 0)])
def test_fused_linear_forward_on_f32_matrix_cores(ops, G, M, N, K, act):
    """cstr_linear_act_fwd_f32 (v_mfma_f32_16x16x4_f32 tiles) against an fp64 reference: create_mlp's Linear + ReLU / Tanh
    (torch_layers.py:110-183), plain and grouped, strided / shared inputs, ragged M, N, K."""
    from core.common import hip_ops

    gen = th.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    gg = max(G, 1)
    w = th.randn(gg, N, K, device="cuda", generator=gen) / K ** 0.5
    b = th.randn(gg, N, device="cuda", generator=gen)
    wide = th.randn(gg, M, K + 3, device="cuda", generator=gen)
    for variant in ("contiguous", "row-strided", "shared"):
        if variant == "contiguous":
            x = wide[:, :, :K].contiguous()
        elif variant == "row-strided":
            x = wide[:, :, 1:K + 1]  # rows K + 3 floats apart, base not 16-byte aligned
        else:
            x = wide[0, :, :K].contiguous().unsqueeze(0).expand(gg, -1, -1)  # stride-0 group dimension
        ref = th.einsum("gmk,gnk->gmn", x.double(), w.double()) + b.double()[:, None, :]
        ref = th.relu(ref) if act == 1 else (th.tanh(ref) if act == 2 else ref)
        if G == 0:
            y = hip_ops.linear_act_fwd(x[0], w[0], b[0], act)
            ref = ref[0]
        else:
            y = hip_ops.linear_act_fwd(x, w, b, act)
        scale = max(1.0, float(ref.abs().max()))
        assert tuple(y.shape) == tuple(ref.shape)
        assert rel_err(y.cpu().numpy(), ref.cpu().numpy(), scale) < 2e-6, variant
    with pytest.raises(ValueError):
        hip_ops.linear_act_fwd(wide[0].t(), w[0], b[0], act)


@pytest.mark.parametrize("G,M,N,K,act", [(0, 256, 256, 256, 1), (2, 256, 256, 256, 1), (0, 256, 4, 256, 1), (2, 256, 256, 6, 0),
                                        (0, 100, 300, 400, 1), (0, 37, 50, 30, 2), (2, 5, 7, 9, 1), (0, 512, 256, 256, 1)])
def test_fused_linear_input_gradient_kernel(ops, G, M, N, K, act):
    """cstr_linear_bwd_input_f32: dz = (gz @ W) * act'(y) -- against fp64."""
    from core.common import hip_ops

    gen = th.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
    gg = max(G, 1)
    gz = th.randn(gg, M, N, device="cuda", generator=gen)
    w = th.randn(gg, N, K, device="cuda", generator=gen) / N ** 0.5
    pre = th.randn(gg, M, K, device="cuda", generator=gen)
    y = th.relu(pre) if act == 1 else (th.tanh(pre) if act == 2 else pre)
    dx = th.einsum("gmn,gnk->gmk", gz.double(), w.double())
    ref = dx * ((y > 0).double() if act == 1 else ((1 - y.double() ** 2) if act == 2 else 1.0))
    if G == 0:
        dz, ref = hip_ops.linear_bwd_input(gz[0], w[0], y[0], act), ref[0]
    else:
        dz = hip_ops.linear_bwd_input(gz, w, y, act)
    assert rel_err(dz.cpu().numpy(), ref.cpu().numpy(), max(1.0, float(ref.abs().max()))) < 2e-6
    if G:  # groups sharing one input: the sum over groups in the same launch
        tot = hip_ops.linear_bwd_input(gz, w, y[0], act, sum_groups=True)
        ref_sum = dx.sum(0) * ((y[0] > 0).double() if act == 1 else ((1 - y[0].double() ** 2) if act == 2 else 1.0))
        assert tuple(tot.shape) == (M, K) and rel_err(tot.cpu().numpy(), ref_sum.cpu().numpy(), max(1.0, float(ref_sum.abs().max()))) < 3e-6


@pytest.mark.parametrize("G,M,N,K", [(0, 256, 256, 256), (2, 256, 256, 6), (0, 256, 256, 4), (2, 256, 256, 256), (0, 100, 300, 400),
                                     (0, 37, 50, 30), (2, 5, 7, 9), (0, 256, 4, 256), (0, 512, 400, 10)])
def test_fused_linear_weight_gradient_kernel(ops, G, M, N, K):
    """cstr_linear_bwd_weight_f32: dW = dz^T x and db = column sums of dz in one launch -- against fp64; strided / shared x."""
    from core.common import hip_ops

    gen = th.Generator(device="cuda").manual_seed(M + 5 * N + 11 * K)
    gg = max(G, 1)
    dz = th.randn(gg, M, N, device="cuda", generator=gen)
    wide = th.randn(gg, M, K + 3, device="cuda", generator=gen)
    for variant in ("contiguous", "row-strided", "shared"):
        if variant == "contiguous":
            x = wide[:, :, :K].contiguous()
        elif variant == "row-strided":
            x = wide[:, :, 1:K + 1]
        else:
            x = wide[0, :, :K].contiguous().unsqueeze(0).expand(gg, -1, -1)
        ref_w = th.einsum("gmn,gmk->gnk", dz.double(), x.double())
        ref_b = dz.double().sum(1)
        dw, db = th.full((gg, N, K), 9.0, device="cuda"), th.full((gg, N), 9.0, device="cuda")
        if G == 0:
            hip_ops.linear_bwd_weight(dz[0], x[0], dw[0], db[0])
        else:
            hip_ops.linear_bwd_weight(dz, x, dw, db)
        scale = float(M) ** 0.5
        assert rel_err(dw.cpu().numpy(), ref_w.cpu().numpy(), scale) < 2e-6, variant
        assert rel_err(db.cpu().numpy(), ref_b.cpu().numpy(), scale) < 2e-6, variant
        dw2 = th.empty_like(dw)
        hip_ops.linear_bwd_weight(dz if G else dz[0], x if G else x[0], dw2 if G else dw2[0], None)
        assert th.equal(dw2 if G else dw2[0], dw if G else dw[0])


@pytest.mark.parametrize("B,A,K", [(256, 2, 256), (4096, 2, 256), (33, 4, 300), (1, 1, 64)])
def test_gaussian_head_with_its_linear_inside(ops, B, A, K):
    """cstr_gaussian_head_gemm_fwd_f32 == head GEMM + cstr_gaussian_head_fwd_f32 (dot-product order aside)."""
    from core.common import hip_ops

    g = th.Generator(device="cuda").manual_seed(B + K)
    wide = th.randn(B, K + 4, device="cuda", generator=g)
    h = wide[:, :K]  # row-strided, 16-byte aligned rows
    w, bias = th.randn(2 * A, K, device="cuda", generator=g) / K ** 0.5, th.randn(2 * A, device="cuda", generator=g) * 0.1
    eps = th.randn(B, A, device="cuda", generator=g)
    ref_p = th.mm(h, w.t())
    ref_a, ref_lp = th.empty(B, A, device="cuda"), th.empty(B, device="cuda")
    hip_ops.gaussian_head_fwd_(ref_p, bias, eps, None, ref_a, ref_lp)
    p, x, lp = th.empty(B, 2 * A, device="cuda"), th.full((B, 3 + A), 5.0, device="cuda"), th.empty(B, device="cuda")
    hip_ops.gaussian_head_gemm_fwd(h, w, bias, p, eps, None, x[:, 3:], lp)
    assert rel_err(p.cpu().numpy(), ref_p.cpu().numpy(), 1.0) < 2e-6
    assert rel_err(x[:, 3:].cpu().numpy(), ref_a.cpu().numpy(), 1.0) < 5e-6 and float(x[:, :3].min()) == 5.0
    assert rel_err(lp.cpu().numpy(), ref_lp.cpu().numpy(), 1.0) < 1e-4  # log(1 - a^2 + 1e-6) amplifies near saturation
    # in-kernel noise: same stream as the GEMM-less kernel for the same (seed, offset)
    c1, c2 = hip_ops.new_rng_ctl(77, "cuda"), hip_ops.new_rng_ctl(77, "cuda")
    e1, e2, a1, a2 = (th.empty(B, A, device="cuda") for _ in range(4))
    hip_ops.gaussian_head_gemm_fwd(h, w, bias, p, e1, c1, a1, None)
    hip_ops.gaussian_head_fwd_(th.mm(h, w.t()), bias, e2, c2, a2, None)
    assert th.equal(e1, e2) and th.equal(c1, c2) and int(c1[1]) == B
    assert rel_err(a1.cpu().numpy(), a2.cpu().numpy(), 1.0) < 5e-6


@pytest.mark.parametrize("B,A,H,act", [(256, 2, 256, 1), (33, 4, 300, 2), (5, 1, 64, 0), (1024, 2, 400, 1)])
def test_gaussian_head_backward_carried_through_its_linear(ops, B, A, H, act):
    """cstr_gaussian_head_bwd_input_f32 == cstr_gaussian_head_bwd_f32 + (g_params @ W) * act'(hidden) (f64 check of the
    product); with cstr_linear_bwd_weight_f32(g_params, hidden) for the head's dW / db."""
    from core.common import hip_ops

    g = th.Generator(device="cuda").manual_seed(B + H)
    wide = th.randn(B, H + 4, device="cuda", generator=g)
    hidden = th.tanh(wide[:, :H]) if act == 2 else (wide[:, :H].clamp(min=0) if act == 1 else wide[:, :H])
    if act:
        wide[:, :H] = hidden
        hidden = wide[:, :H]  # row-strided
    w = th.randn(2 * A, H, device="cuda", generator=g) / H ** 0.5
    params, eps = th.randn(B, 2 * A, device="cuda", generator=g), th.randn(B, A, device="cuda", generator=g)
    params[0, A] = 5.0  # log_std beyond LOG_STD_MAX: its gradient is cut
    x = th.zeros(B, 3 + A, device="cuda")
    lp = th.empty(B, device="cuda")
    hip_ops.gaussian_head_fwd_(params.clone(), th.zeros(2 * A, device="cuda"), eps, None, x[:, 3:], lp)
    g_x, g_lp = th.randn(B, 3 + A, device="cuda", generator=g), th.randn(B, device="cuda", generator=g)
    ref_gp, ref_gb = th.empty(B, 2 * A, device="cuda"), th.empty(2 * A, device="cuda")
    hip_ops.gaussian_head_bwd(g_x[:, 3:], g_lp, x[:, 3:], params, eps, ref_gp, ref_gb)
    ref_dz = ref_gp.double() @ w.double()
    if act == 1:
        ref_dz = ref_dz * (hidden > 0)
    elif act == 2:
        ref_dz = ref_dz * (1.0 - hidden.double() ** 2)
    gp, dz = th.full((B, 2 * A), 7.0, device="cuda"), th.full((B, H), 7.0, device="cuda")
    hip_ops.gaussian_head_bwd_input(g_x[:, 3:], g_lp, x[:, 3:], params, eps, w, hidden, act, gp, dz)
    assert th.equal(gp, ref_gp)
    assert rel_err(dz.cpu().numpy(), ref_dz.cpu().numpy(), max(1.0, float(ref_dz.abs().max()))) < 2e-6
    dw, db = th.empty(2 * A, H, device="cuda"), th.empty(2 * A, device="cuda")
    hip_ops.linear_bwd_weight(gp, hidden, dw, db)
    ref_dw = gp.double().t() @ hidden.double()
    assert rel_err(dw.cpu().numpy(), ref_dw.cpu().numpy(), max(1.0, float(ref_dw.abs().max()))) < 2e-6
    assert rel_err(db.cpu().numpy(), ref_gb.cpu().numpy(), max(1.0, float(ref_gb.abs().max()))) < 2e-6
    # no g_action / no g_logp forms
    hip_ops.gaussian_head_bwd(None, g_lp, x[:, 3:], params, eps, ref_gp, None)
    hip_ops.gaussian_head_bwd_input(None, g_lp, x[:, 3:], params, eps, w, hidden, act, gp, dz)
    assert th.equal(gp, ref_gp)


def test_weight_gradients_of_several_layers_in_one_launch(ops):
    """cstr_linear_bwd_weight_sets_f32 == one cstr_linear_bwd_weight_f32 per set, bit for bit (same tile code), for sets of
    different shapes (an actor's head 4 x 256, hidden 256 x 256 and input 256 x 4 layers), strided inputs, optional db; and
    the deferred-gradient context around a FastMLP backward."""
    from core.common import fused, hip_ops
    from core.common.arena import ParamArena
    from core.common.torch_layers import create_mlp

    g = th.Generator(device="cuda").manual_seed(3)
    for M in (256, 20):
        shapes = [(4, 256), (256, 256), (256, 4), (37, 50), (300, 400)]
        sets, refs = [], []
        for i, (N, K) in enumerate(shapes):
            dz = th.randn(M, N, device="cuda", generator=g)
            x = th.randn(M, K + 4, device="cuda", generator=g)[:, :K]
            dw, db = th.full((N, K), 9.0, device="cuda"), (None if i == 3 else th.full((N,), 9.0, device="cuda"))
            rw, rb = th.empty(N, K, device="cuda"), (None if db is None else th.empty(N, device="cuda"))
            hip_ops.linear_bwd_weight(dz, x, rw, rb)
            sets.append((dz, x, dw, db))
            refs.append((rw, rb))
        hip_ops.linear_bwd_weight_sets(sets)
        for (_, _, dw, db), (rw, rb) in zip(sets, refs):
            assert th.equal(dw, rw) and (db is None or th.equal(db, rb))
    with pytest.raises(ValueError):
        hip_ops.linear_bwd_weight_sets([])
    # FastMLP backward: queued gradients == immediate gradients
    th.manual_seed(1)
    seq = th.nn.Sequential(*create_mlp(6, 3, [64, 48], th.nn.ReLU))
    arena = ParamArena(list(seq.parameters()), "cuda")
    mlp = fused.FastMLP(seq)
    x, gy = th.randn(128, 6, device="cuda"), th.randn(128, 3, device="cuda")
    grads = lambda: th.cat([p.grad.reshape(-1) for p in seq.parameters()])  # noqa: E731
    arena.grad.zero_()
    th.autograd.backward([mlp(x)], [gy])
    ref = grads().clone()
    arena.grad.fill_(7.0)
    with fused.deferred_weight_grads():
        th.autograd.backward([mlp(x)], [gy])
        assert fused.USE_FUSED_LINEAR is False or float(grads().min()) == 7.0  # nothing written yet
    assert th.equal(grads(), ref) and float(ref.abs().max()) > 0.0


@pytest.mark.parametrize("M,K0,H1,H2,A,act", [(4096, 4, 256, 256, 2, 1), (256, 4, 256, 256, 2, 1), (100, 8, 400, 300, 4, 2), (1, 3, 64, 32, 1, 1),
                                              (37, 8, 64, 64, 2, 0)])
def test_whole_policy_network_in_one_launch(ops, M, K0, H1, H2, A, act):
    """cstr_policy_rows_fwd_f32 against the layer-by-layer evaluation (f64 reference for the hidden layers): squashed-Gaussian
    head with given noise, the same head drawing from the Philox stream (same positions as cstr_gaussian_head_fwd_f32), and
    the deterministic head."""
    from core.common import hip_ops

    g = th.Generator(device="cuda").manual_seed(M + H1)
    r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
    x = r(M, K0 + 3)[:, :K0]  # row-strided
    w1, b1, w2, b2 = r(H1, K0) / K0 ** 0.5, r(H1) * 0.1, r(H2, H1) / H1 ** 0.5, r(H2) * 0.1
    w3, b3, eps = r(2 * A, H2) / H2 ** 0.5, r(2 * A) * 0.1, r(M, A)
    f = {0: lambda t: t, 1: th.relu, 2: th.tanh}[act]
    h2 = f(f(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
    params = (h2 @ w3.double().t()).float()  # bias added by the head kernel
    ref_a, ref_lp = th.empty(M, A, device="cuda"), th.empty(M, device="cuda")
    hip_ops.gaussian_head_fwd_(params.clone(), b3, eps, None, ref_a, ref_lp)
    xbuf, lp = th.full((M, 5 + A), 9.0, device="cuda"), th.empty(M, device="cuda")
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, act, 0, 0, xbuf[:, 5:], eps=eps, logp=lp)
    assert float(xbuf[:, :5].min()) == 9.0
    assert rel_err(xbuf[:, 5:].cpu().numpy(), ref_a.cpu().numpy(), 1.0) < 1e-5
    assert rel_err(lp.cpu().numpy(), ref_lp.cpu().numpy(), 1.0) < 2e-4  # log(1 - a^2 + 1e-6) amplifies near saturation
    # in-kernel noise: the stream positions of the separate head kernel
    c1, c2 = hip_ops.new_rng_ctl(77, "cuda"), hip_ops.new_rng_ctl(77, "cuda")
    c1[1] = c2[1] = 1000
    a1, a2, e2 = th.empty(M, A, device="cuda"), th.empty(M, A, device="cuda"), th.empty(M, A, device="cuda")
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, act, 0, 0, a1, rng_ctl=c1)
    hip_ops.gaussian_head_fwd_(params.clone(), b3, e2, c2, a2, None)
    assert th.equal(c1, c2) and int(c1[1]) == 1000 + M and int(c1[2]) == 0
    assert rel_err(a1.cpu().numpy(), a2.cpu().numpy(), 1.0) < 1e-5
    # deterministic head (TD3 actor): tanh(h2 W3^T + b3) with the first A rows of w3
    out = th.empty(M, A, device="cuda")
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3[:A].contiguous(), b3[:A].contiguous(), act, 1, 2, out)
    ref = th.tanh(h2 @ w3[:A].double().t() + b3[:A].double())
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy(), 1.0) < 1e-5
    with pytest.raises(Exception):
        hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, act, 0, 0, a1)  # no noise source


@pytest.mark.parametrize("M,K0,H1,H2,A,act", [(50, 4, 512, 400, 2, 1), (300, 4, 320, 272, 2, 1), (64, 4, 16, 16, 1, 2), (129, 8, 400, 300, 4, 1),
                                              (200, 4, 256, 256, 2, 1), (77, 4, 400, 304, 2, 0), (33, 4, 64, 496, 2, 1), (4096, 4, 128, 80, 2, 1)])
def test_pipelined_policy_kernel_at_many_widths(ops, M, K0, H1, H2, A, act):
    """The pipelined policy kernel (tile-major weight copy given: layer 2 as a per-chunk register ring, round 3) at widths that exercise
    every shape of its loops -- one and two tile pairs per wave, a ring that does / does not divide K (the next pair's chunks carried
    over or requested afresh), partial last tiles and chunks, the exact-shape instantiations -- against the f64 layer-by-layer
    evaluation and against the first-version kernel (no copy given)."""
    from core.common import hip_ops

    g = th.Generator(device="cuda").manual_seed(M + H1 + H2)
    r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
    x = r(M, K0)
    w1, b1, w2, b2 = r(H1, K0) / K0 ** 0.5, r(H1) * 0.1, r(H2, H1) / H1 ** 0.5, r(H2) * 0.1
    w3, b3 = r(A, H2) / H2 ** 0.5, r(A) * 0.1
    f = {0: lambda t: t, 1: th.relu, 2: th.tanh}[act]
    h2 = f(f(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
    ref = th.tanh(h2 @ w3.double().t() + b3.double())
    tiles = hip_ops.policy_swizzle(w2)
    out, out_v1 = th.empty(M, A, device="cuda"), th.empty(M, A, device="cuda")
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, act, 1, 2, out, w2_swz=tiles)
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, act, 1, 2, out_v1)
    th.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy(), 1.0) < 1e-5
    assert rel_err(out.cpu().numpy(), out_v1.cpu().numpy(), 1.0) < 2e-6  # (the two kernels sum the head in different orders)


@pytest.mark.parametrize("N,K", [(256, 256), (300, 400), (20, 8), (16, 64)])
def test_tile_major_weight_copy_and_its_adam_shadow(ops, N, K):
    """cstr_policy_swizzle_f32 against the torch permutation; the policy kernel with the copy is bit-identical to the one
    without; cstr_adam_multi_f32 keeps the copy equal to swizzle(updated weights) step after step."""
    from core.common import hip_ops
    from core.common.arena import FlatAdam, ParamArena

    g = th.Generator(device="cuda").manual_seed(N + K)
    w = th.randn(N, K, device="cuda", generator=g)
    t, kc = -(-N // 16), -(-K // 16)
    wp = th.zeros(t * 16, kc * 16, device="cuda")
    wp[:N, :K] = w
    ref = wp.reshape(t, 16, kc, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)  # [tile][k chunk][h][r][4]
    assert th.equal(hip_ops.policy_swizzle(w), ref)
    if K % 4 == 0 and K >= 8:
        M, K0, A = 100, 4, 2
        r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
        x, w1, b1, b2, w3, b3, eps = r(M, K0), r(K, K0), r(K), r(N), r(2 * A, N) / N ** 0.5, r(2 * A), r(M, A)
        a1, a2 = th.empty(M, A, device="cuda"), th.empty(M, A, device="cuda")
        hip_ops.policy_rows_fwd(x, w1, b1, w / K ** 0.5, b2, w3, b3, 1, 0, 0, a1, eps=eps)
        hip_ops.policy_rows_fwd(x, w1, b1, w / K ** 0.5, b2, w3, b3, 1, 0, 0, a2, eps=eps, w2_swz=hip_ops.policy_swizzle(w / K ** 0.5))
        # with the copy the software-pipelined kernel runs: hidden layers bit-identical (same k order), the head is a split-K
        # MFMA tile instead of per-lane partial dot products -> a few ulp apart
        assert rel_err(a1.cpu().numpy(), a2.cpu().numpy(), 1.0) < 2e-6
    # an arena with other parameters around the matrix; three steps with and without the shadow
    def make():
        th.manual_seed(3)
        ps = [th.nn.Parameter(th.randn(7)), th.nn.Parameter(th.randn(N, K)), th.nn.Parameter(th.randn(5, 3))]
        arena = ParamArena(ps, "cuda")
        return ps, arena, FlatAdam(arena, lr=1e-2)

    (ps_a, ar_a, opt_a), (ps_b, ar_b, opt_b) = make(), make()
    shadow = opt_b.add_weight_shadow(ps_b[1])
    assert th.equal(shadow, hip_ops.policy_swizzle(ps_b[1].detach()))
    for step in range(3):
        grad = th.randn(ar_a.numel, device="cuda", generator=g)
        ar_a.grad.copy_(grad), ar_b.grad.copy_(grad)
        opt_a.step()
        opt_b.step() if step != 1 else opt_b.step_with()
        assert th.equal(ar_a.flat, ar_b.flat)  # same arithmetic through either kernel
        assert th.equal(shadow, hip_ops.policy_swizzle(ps_b[1].detach()))
    with th.no_grad():
        ps_b[1].mul_(2.0)  # torch changes the weights: the version counter moves
    opt_b.refresh_shadow(force=False)
    assert th.equal(shadow, hip_ops.policy_swizzle(ps_b[1].detach()))


def test_grouped_actor_forward_and_single_agent_backward(ops):
    """FastActorGroup (cstr_linear_act_fwd_sets_f32): four agents' actor MLPs, one launch per layer, actions written into the
    column blocks of a joint buffer; one agent differentiated -- against the per-agent nn.Modules and autograd."""
    from core.common import fused
    from core.common.arena import ParamArena
    from core.common.torch_layers import create_mlp

    th.manual_seed(0)
    n_agents, M, K0, D = 4, 96, 2, 8
    seqs = [th.nn.Sequential(*create_mlp(K0, 1, [40, 24], th.nn.ReLU, squash_output=True)) for _ in range(n_agents)]
    refs = [th.nn.Sequential(*create_mlp(K0, 1, [40, 24], th.nn.ReLU, squash_output=True)).cuda() for _ in range(n_agents)]
    arena = ParamArena([p for sq in seqs for p in sq.parameters()], "cuda")
    for sq, rf in zip(seqs, refs):
        rf.load_state_dict({k: v.detach().clone() for k, v in sq.state_dict().items()})
    mlps = [fused.FastMLP(sq) for sq in seqs]
    assert fused.FastActorGroup.supported(mlps)
    group = fused.FastActorGroup(mlps)
    obs = th.randn(M, D, device="cuda")
    ins = [obs[:, 2 * j:2 * j + 2] for j in range(n_agents)]
    cols = [(D + j, D + j + 1) for j in range(n_agents)]
    x = th.full((M, D + n_agents), 3.0, device="cuda")
    group.forward(ins, x, cols)  # no gradient
    want = th.cat([rf(i) for rf, i in zip(refs, ins)], dim=1)
    assert rel_err(x[:, D:].cpu().numpy(), want.detach().cpu().numpy(), 1.0) < 2e-6 and float(x[:, :D].min()) == 3.0
    agent = 2
    arena.grad.fill_(5.0)
    x2 = th.full((M, D + n_agents), 3.0, device="cuda")
    out = group.forward(ins, x2, cols, grad_agent=agent)
    assert out.requires_grad and rel_err(out[:, D:].detach().cpu().numpy(), want.detach().cpu().numpy(), 1.0) < 2e-6
    g = th.randn(M, D + n_agents, device="cuda")
    out.backward(g)
    want[:, agent:agent + 1].backward(g[:, D + agent:D + agent + 1])
    for (nm, p), (_, q) in zip(seqs[agent].named_parameters(), refs[agent].named_parameters()):
        scale = max(1e-2, float(q.grad.abs().mean()))
        assert rel_err(p.grad.cpu().numpy(), q.grad.cpu().numpy(), scale) < 2e-5, nm
    others = [p for j, sq in enumerate(seqs) if j != agent for p in sq.parameters()]
    assert all(bool((p.grad == 5.0).all()) for p in others)  # frozen agents' gradient views untouched


def test_policy_launch_leaves_the_philox_offset_to_the_collect_launch(ops):
    """cstr_policy_mlp_t.reserved bit 0 + cstr_collect_step_rng_f32: the rollout's policy launch skips its own 256-workgroup ticket and
    the fused collect launch that consumes the action advances the stream offset in ITS last-workgroup epilogue. Same actions as the
    self-advancing launch, offset untouched by the policy launch, advanced by exactly the row count by the collect launch, tickets
    back at zero."""
    from core import _native as nv
    from core.common import hip_ops

    M, K0, H, A = 512, 4, 64, 2
    g = th.Generator(device="cuda").manual_seed(3)
    r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
    x, w1, b1, w2, b2, w3, b3 = r(M, K0), r(H, K0), r(H), r(H, H) / 8, r(H), r(2 * A, H) / 8, r(2 * A)
    tiles = hip_ops.policy_swizzle(w2)
    c1, c2 = hip_ops.new_rng_ctl(9, "cuda"), hip_ops.new_rng_ctl(9, "cuda")
    c1[1] = c2[1] = 77
    a1, a2 = th.empty(M, A, device="cuda"), th.empty(M, A, device="cuda")
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, 0, 0, a1, rng_ctl=c1, w2_swz=tiles)
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, 0, 0, a2, rng_ctl=c2, w2_swz=tiles, defer_rng_advance=True)
    assert th.equal(a1, a2) and int(c1[1]) == 77 + M and int(c2[1]) == 77 and int(c2[2]) == 0
    ring = hip_ops.DeviceRing(4, M, 4, 2, "cuda")
    obs = (th.rand(M, 4, device="cuda", generator=g) - 0.5).contiguous()
    steps = th.zeros(M, dtype=th.int32, device="cuda")
    hip_ops.collect_step(nv.default_coef(), "euler", ring, obs, steps, a2, True, [-1, -1], [1, 1], reset_obs=obs.clone(), rng_advance=(c2, M))
    assert th.equal(c1, c2) and int(ring.ctl[0]) == 1 and int(ring.ctl[2]) == 0
    # the first version of the kernel (no tile-major copy) honours the flag too
    c3 = hip_ops.new_rng_ctl(9, "cuda")
    c3[1] = 77
    a3 = th.empty(M, A, device="cuda")
    hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, 0, 0, a3, rng_ctl=c3, defer_rng_advance=True)
    assert int(c3[1]) == 77 and rel_err(a3.cpu().numpy(), a1.cpu().numpy(), 1.0) < 2e-6
