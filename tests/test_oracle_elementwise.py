"""Pin the oracle's element-wise restatements (TD target, polyak, Adam, action scaling, PCG64
reset draw) against torch / numpy running in this process -- the libraries the reference calls.
"""
import numpy as np
import pytest
import torch as th

from oracle import cstr_oracle as orc


def _ulp(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("n", [1, 7, 8, 1023, 135682])
@pytest.mark.parametrize("tau", [0.005, 0.01, 1.0, 0.0])
def test_polyak_bit_exact_vs_torch(n, tau):
    """core/common/utils.py:478-481 -- the two in-place torch ops, bit for bit."""
    g = th.Generator().manual_seed(n)
    p, t = th.randn(n, generator=g), th.randn(n, generator=g)
    exp = t.clone()
    exp.mul_(1 - tau)
    th.add(exp, p, alpha=tau, out=exp)
    got = orc.polyak(p.numpy(), t.numpy(), tau)
    np.testing.assert_array_equal(got, exp.numpy())


def test_td_target_sac_vs_torch():
    """core/sac/sac.py:250-254"""
    g = th.Generator().manual_seed(0)
    B = 4096
    q1, q2, lp = th.randn(B, 1, generator=g) * 5, th.randn(B, 1, generator=g) * 5, th.randn(B, 1, generator=g)
    rew = -th.rand(B, 1, generator=g) * 8
    done = (th.rand(B, 1, generator=g) < 0.1).float()
    ent = th.tensor([0.37])
    nq, _ = th.min(th.cat((q1, q2), dim=1), dim=1, keepdim=True)
    nq = nq - ent * lp.reshape(-1, 1)
    exp = rew + (1 - done) * 0.99 * nq
    got = orc.td_target_min(q1.numpy(), q2.numpy(), lp.numpy(), rew.numpy(), done.numpy(), float(ent), 0.99)
    np.testing.assert_array_equal(got, exp.numpy().reshape(-1))


def test_td_target_td3_vs_torch():
    """core/td3/td3.py:174-176"""
    g = th.Generator().manual_seed(1)
    B = 1000
    q1, q2 = th.randn(B, 1, generator=g) * 5, th.randn(B, 1, generator=g) * 5
    rew, done = -th.rand(B, 1, generator=g), (th.rand(B, 1, generator=g) < 0.3).float()
    nq, _ = th.min(th.cat((q1, q2), dim=1), dim=1, keepdim=True)
    exp = rew + (1 - done) * 0.99 * nq
    got = orc.td_target_min(q1.numpy(), q2.numpy(), None, rew.numpy(), done.numpy(), 0.0, 0.99)
    np.testing.assert_array_equal(got, exp.numpy().reshape(-1))


@pytest.mark.parametrize("n", [5, 1000, 68100])
def test_adam_vs_torch(n):
    """torch.optim.Adam defaults as built by the reference policies (lr 3e-4)."""
    g = th.Generator().manual_seed(n)
    p = th.nn.Parameter(th.randn(n, generator=g))
    opt = th.optim.Adam([p], lr=3e-4)
    mine_p, m, v = p.detach().numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for step in range(1, 6):
        grad = th.randn(n, generator=g) * (10.0 ** float(th.randint(-3, 2, (1,), generator=g)))
        p.grad = grad.clone()
        opt.step()
        mine_p, m, v = orc.adam_step(mine_p, grad.numpy(), m, v, step, 3e-4)
        st = opt.state[p]
        # bit-exact except where torch's vectorised kernel hands a chunk tail to its scalar path
        # (different fusion): allow a few ulp on < 0.01 % of the elements
        for got, ref in ((m, st["exp_avg"]), (v, st["exp_avg_sq"]), (mine_p, p.detach())):
            u = _ulp(got, ref.numpy())
            assert u.max() <= 4 and (u > 0).mean() < 1e-4, (step, u.max(), (u > 0).mean())


def test_action_scale_chain_vs_numpy():
    """policies.py:388-413 + off_policy_algorithm.py:396-406 evaluated with numpy f32 arrays."""
    rng = np.random.default_rng(0)
    for low, high in (([-1, -1], [1, 1]), ([30, -2], [250, 5])):
        low, high = np.array(low, np.float32), np.array(high, np.float32)
        a = np.tanh(rng.normal(0, 2, (4096, 2))).astype(np.float32)
        u = low + (0.5 * (a + 1.0) * (high - low))          # predict(): unscale_action
        s = 2.0 * ((u - low) / (high - low)) - 1.0          # scale_action
        e = low + (0.5 * (s + 1.0) * (high - low))          # unscale_action
        assert u.dtype == s.dtype == e.dtype == np.float32
        buf, env = orc.action_scale_chain(a, True, low, high)
        np.testing.assert_array_equal(buf, s)
        np.testing.assert_array_equal(env, e)
        uw = rng.uniform(low, high, (512, 2)).astype(np.float32)  # warm-up sample already in [low, high]
        s = 2.0 * ((uw - low) / (high - low)) - 1.0
        buf, env = orc.action_scale_chain(uw, False, low, high)
        np.testing.assert_array_equal(buf, s)
        np.testing.assert_array_equal(env, low + (0.5 * (s + 1.0) * (high - low)))


def test_pcg64_reset_draw_vs_numpy_generator():
    """generate_initial_state (twoseriescstr.py:187-224) replayed with numpy's own Generator.
    UNPINNED w.r.t. the reference's gymnasium seeding; this pins the PCG64/uniform restatement."""
    seeds = [0, 1, 7, 4095, 123456]
    st = orc.pcg64_states_from_seeds(seeds)
    lo = np.array([0.0, 273.15, 0.0, 273.15], np.float32)
    hi = np.array([0.7, 400.0, 0.7, 400.0], np.float32)
    gens = [np.random.Generator(np.random.PCG64(np.random.SeedSequence(s))) for s in seeds]
    for _ in range(3):  # consecutive resets continue each env's stream
        got = orc.reset_draw(st)
        for i, gen in enumerate(gens):
            s = np.array([gen.uniform(0.05, 0.45), gen.uniform(280, 380), gen.uniform(0.05, 0.45 * 0.8), gen.uniform(280, 380)])
            s += gen.uniform(-0.05, 0.05, size=4)
            if s[1] < s[3]:
                s[1], s[3] = s[3], s[1]
            if s[0] < s[2]:
                s[0], s[2] = s[2], s[0]
            s = np.clip(s, lo, hi)
            exp = (2.0 * (s - lo) / (hi - lo) - 1.0).astype(np.float32)
            np.testing.assert_array_equal(got[i], exp)
