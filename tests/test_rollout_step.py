"""GPU parity tests of the one-launch rollout (cstr_rollout_step_f32) and the gather launch behind it
(cstr_replay_gather_packed_f32): bit-identical to the three launches they replace (cstr_policy_rows_fwd_f32 ->
cstr_collect_step_rng_f32 -> cstr_replay_sample_packed_mt19937_f32), which the other test files pin to the oracle, the
golden vectors and numpy; the one-wave MT19937 index draw is additionally checked against numpy's RandomState directly.
"""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert th.cuda.is_available(), "GPU tests need an MI355X"
    from core import _native as nv
    from core.common import hip_ops

    nv.lib()  # fail loudly if the HIP extension is missing
    return hip_ops


def _obs8(o4):
    lo, hi = th.tensor([0.0, 273.15, 0.0, 273.15], device="cuda"), th.tensor([0.7, 400.0, 0.7, 400.0], device="cuda")
    return th.cat([o4, th.minimum(th.maximum(lo + (o4 + 1.0) * (hi - lo) / 2.0, lo), hi)], dim=1).contiguous()


class _World:
    """Everything one rollout + sample touches, so that two code paths can run on identical copies."""

    def __init__(self, ops, n, d, rows, batch, h1, h2, head, seed, max_steps, static=False):
        from core import _native as nv

        g = th.Generator(device="cuda").manual_seed(seed)
        r = lambda *sh: th.randn(*sh, device="cuda", generator=g)  # noqa: E731
        a = 2
        self.ops, self.n, self.d, self.a, self.batch, self.head = ops, n, d, a, batch, head
        o4 = (th.rand(n, 4, device="cuda", generator=g) * 2 - 1).contiguous()
        self.env_obs = o4 if d == 4 else _obs8(o4)
        self.step_count = th.randint(max_steps - 6, max_steps, (n,), device="cuda", generator=g, dtype=th.int32)
        self.coef = nv.default_coef(max_steps=max_steps)
        self.ring = ops.DeviceRing(rows, n, d, a, "cuda")
        n_out = 2 * a if head == 0 else a
        self.w = [r(h1, d) / d ** 0.5, r(h1) * 0.1, r(h2, h1) / h1 ** 0.5, r(h2) * 0.1, r(n_out, h2) / h2 ** 0.5, r(n_out) * 0.1]
        self.swz = ops.policy_swizzle(self.w[2])
        self.rng_ctl = ops.new_rng_ctl(123, "cuda") if head == 0 else None
        self.pcg = th.randint(1, 2 ** 62, (n, 4), device="cuda", generator=g, dtype=th.int64)
        self.pcg[:, 3] |= 1  # odd increment, like a seeded PCG64
        # init_mode="static": the per-env drifting f64 init_state the reset draw perturbs in place (twoseriescstr.py:246-255)
        self.static_init = th.tensor([0.45, 310.0, 0.25, 290.0], dtype=th.float64, device="cuda").repeat(n, 1).contiguous() if static else None
        self.mt = th.zeros(628, dtype=th.int32, device="cuda")
        ops.mt19937_seed(self.mt, 4242 + seed)
        self.rew, self.done = th.zeros(n, device="cuda"), th.zeros(n, device="cuda")
        self.ep_return, self.ep_stats = th.zeros(n, device="cuda"), th.zeros(4, dtype=th.float64, device="cuda")
        w = d + a
        self.x_data, self.x_next, self.x_pi = (th.zeros(batch, w, device="cuda") for _ in range(3))
        self.s_done, self.s_rew = th.zeros(batch, 1, device="cuda"), th.zeros(batch, 1, device="cuda")
        self.bi, self.ei = th.zeros(batch, dtype=th.int64, device="cuda"), th.zeros(batch, dtype=th.int64, device="cuda")
        self.idx = th.zeros(2, batch, dtype=th.int32, device="cuda")
        self.low, self.high = np.array([-1, -1], np.float32), np.array([1, 1], np.float32)

    def step_separate(self, noise):
        ops, n = self.ops, self.n
        pol = th.empty(n, self.a, device="cuda")
        ops.policy_rows_fwd(self.env_obs, *self.w, 1, self.head, 2 if self.head else 0, pol, rng_ctl=self.rng_ctl, w2_swz=self.swz,
                            defer_rng_advance=True)
        ops.collect_step(self.coef, "euler" if self.d == 4 else "rk4", self.ring, self.env_obs, self.step_count, pol, 1, self.low, self.high,
                         noise=noise, pcg_state=self.pcg, static_init=self.static_init, reward_out=self.rew, done_out=self.done, ep_return=self.ep_return,
                         ep_stats=self.ep_stats, rng_advance=None if self.rng_ctl is None else (self.rng_ctl, n))
        ops.replay_sample_packed(self.ring, self.mt, self.batch, self.x_data, self.x_next, self.x_pi, self.s_done, self.s_rew, self.bi, self.ei)

    def step_fused(self, noise):
        ops, n = self.ops, self.n
        ops.rollout_step(self.env_obs, *self.w, 1, self.head, 2 if self.head else 0, self.swz, self.rng_ctl, self.coef,
                         "euler" if self.d == 4 else "rk4", self.ring, self.env_obs, self.step_count, 1, self.low, self.high, noise=noise,
                         pcg_state=self.pcg, static_init=self.static_init, reward_out=self.rew, done_out=self.done, ep_return=self.ep_return,
                         ep_stats=self.ep_stats, mt_state=self.mt, sample_idx=self.idx)
        ops.replay_gather_packed(self.ring, self.idx, self.batch, self.x_data, self.x_next, self.x_pi, self.s_done, self.s_rew, self.bi, self.ei,
                                 advance_ring=True, rng_advance=None if self.rng_ctl is None else (self.rng_ctl, n))

    def state(self):
        r = self.ring
        out = dict(env_obs=self.env_obs, step_count=self.step_count, pcg=self.pcg, mt=self.mt, rew=self.rew, done=self.done,
                   ep_return=self.ep_return, x_data=self.x_data, x_next=self.x_next, x_pi=self.x_pi, s_done=self.s_done, s_rew=self.s_rew,
                   bi=self.bi, ei=self.ei, ring_obs=r.observations, ring_next=r.next_observations, ring_act=r.actions, ring_rew=r.rewards,
                   ring_done=r.dones, ring_timeout=r.timeouts, ring_ctl=r.ctl)
        if self.rng_ctl is not None:
            out["rng_ctl"] = self.rng_ctl
        if self.static_init is not None:
            out["static_init"] = self.static_init
        return {k: v.clone() for k, v in out.items()}


@pytest.mark.parametrize("n,d,rows,batch,h1,h2,head,static", [
    (4096, 4, 5, 256, 256, 256, 0, False),   # the bench shape (SAC): register-resident B operand
    (1000, 4, 3, 100, 400, 300, 1, False),   # TD3's class-default widths (pipelined form), ragged last workgroup, deterministic head
    (2048, 8, 4, 64, 64, 64, 0, False),      # obs 8 / RK4 (north_star variant)
    (1040, 4, 7, 300, 256, 256, 1, False),
    (528, 4, 4, 32, 64, 64, 0, True),        # init_mode="static": resets walk the per-env f64 init_state
    (520, 8, 4, 32, 64, 64, 1, True),
])
def test_rollout_step_equals_the_three_launches(ops, n, d, rows, batch, h1, h2, head, static):
    """13 vec-steps (ring wraps, episodes end and reset from the PCG64 streams, the MT19937 block is twisted most steps): every
    tensor either path touches is bit-identical after each step; the f64 episode-return sum (float atomics) to 1e-12."""
    a, b = (_World(ops, n, d, rows, batch, h1, h2, head, seed=n + h1, max_steps=9, static=static) for _ in range(2))
    g = th.Generator(device="cuda").manual_seed(1)
    ends = 0
    for k in range(13):
        noise = None if k % 3 else (th.randn(n, 2, device="cuda", generator=g) * 0.1).contiguous()
        a.step_separate(noise)
        b.step_fused(noise)
        th.cuda.synchronize()
        sa, sb = a.state(), b.state()
        for key in sa:
            assert th.equal(sa[key], sb[key]), f"step {k}: {key} differs"
        ea, eb = a.ep_stats.cpu().numpy(), b.ep_stats.cpu().numpy()
        assert ea[0] == eb[0] and ea[2] == eb[2] and abs(ea[1] - eb[1]) <= 1e-12 * max(1.0, abs(ea[1]))
        ends = ea[0]
        ctl = sb["ring_ctl"].cpu().numpy()
        assert ctl[0] == (k + 1) % rows and ctl[1] == int(k + 1 >= rows) and ctl[3] == k + 1
    assert ends > n  # every env finished at least one episode: the reset draws ran inside the fused launch


@pytest.mark.parametrize("n,d,integ,h,rows,batch", [(4096, 4, "euler", 256, 5, 256), (2048, 8, "rk4", 256, 4, 256)])
def test_rollout_step_against_the_oracle_at_the_bench_shape(ops, n, d, integ, h, rows, batch):
    """the one-launch rollout's env step runs a different instruction stream (`collect_env_quad`: a quad of lanes per
    env, DPP permutes) than the kernels the other tests pin to the oracle, so it is compared with the ORACLE directly, at the bench shape
    (4096 envs, obs 4, 256 x 256 SAC head) and at the north_star-literal one (obs 8, RK4): 13 vec-steps, every step on the launch's own
    actions -- oracle action chain -> oracle vec_step (reference twoseriescstr.py:394-503 restated in C) with reset observations from the
    oracle's PCG64 reset draw (twoseriescstr.py:187-224) -> oracle ring add (off_policy_algorithm.py:445-508, buffers.py:247-283).
    Flags, step counters, stored actions and observations, PCG64 states and the ring position: bit-exact; new observations / rewards:
    1e-6 of the box scale per step (Euler; 2e-6 RK4): single-ulp expf differences."""
    from core import _native as nv
    from conftest import rel_err
    from oracle import cstr_oracle as orc

    max_steps = 9
    w = _World(ops, n, d, rows, batch, h, h, 0, seed=n + d, max_steps=max_steps)
    oring = orc.ReplayRing(rows, n, 4, 2)
    ocoef = orc.default_coef(max_steps=max_steps)
    states = np.zeros(n, orc.PCG_DTYPE)
    pcg_h = w.pcg.cpu().numpy().view(np.uint64)
    for i, f in enumerate(("state_hi", "state_lo", "inc_hi", "inc_lo")):
        states[f] = pcg_h[:, i]
    act = th.empty(n, 2, device="cuda")
    obs_floor = 1.0  # error relative to max(|x|, 1), as in tests/test_hip_kernels.py
    tol = 1e-6 if integ == "euler" else 2e-6  # <= 1e-6 of the box scale per step (numpy / libm / ocml expf differ by <= 2 ulp)
    episodes = 0
    for k in range(13):
        obs_h, steps_h = w.env_obs.cpu().numpy()[:, :4].copy(), w.step_count.cpu().numpy().copy()
        ops.rollout_step(w.env_obs, *w.w, 1, 0, 0, w.swz, w.rng_ctl, w.coef, integ, w.ring, w.env_obs, w.step_count, 1, w.low, w.high,
                         pcg_state=w.pcg, reward_out=w.rew, done_out=w.done, ep_return=w.ep_return, ep_stats=w.ep_stats, mt_state=w.mt,
                         sample_idx=w.idx, action_out=act)
        ops.replay_gather_packed(w.ring, w.idx, batch, w.x_data, w.x_next, w.x_pi, w.s_done, w.s_rew, w.bi, w.ei, advance_ring=True,
                                 rng_advance=(w.rng_ctl, n))
        th.cuda.synchronize()
        a_h = act.cpu().numpy()
        assert np.isfinite(a_h).all() and np.abs(a_h).max() <= 1.0
        buf_a, env_a = orc.action_scale_chain(a_h, True, w.low, w.high)
        _, _, _, done0, _, _ = orc.vec_step(obs_h, env_a, steps_h, None, integrator=integ, coef=ocoef)
        reset = orc.reset_draw(states, mask=done0 > 0)  # advances exactly the finished envs' streams, like the launch
        nxt, after, rew, done, tout, steps2 = orc.vec_step(obs_h, env_a, steps_h, reset, integrator=integ, coef=ocoef)
        oring.add(obs_h, nxt, buf_a, rew, done, tout)
        np.testing.assert_array_equal(done, done0)
        np.testing.assert_array_equal(w.done.cpu().numpy(), done, err_msg=f"step {k}")
        np.testing.assert_array_equal(w.step_count.cpu().numpy(), steps2, err_msg=f"step {k}")
        got = w.env_obs.cpu().numpy()
        fin = done > 0
        np.testing.assert_array_equal(got[fin, :4], after[fin], err_msg=f"step {k}: reset observations")  # PCG64 draws: bit-exact
        assert rel_err(got[~fin, :4], after[~fin], obs_floor) < tol, k
        assert rel_err(w.rew.cpu().numpy(), rew, 1.0) < 4 * tol, k
        pcg_now = w.pcg.cpu().numpy().view(np.uint64)
        for i, f in enumerate(("state_hi", "state_lo", "inc_hi", "inc_lo")):
            np.testing.assert_array_equal(pcg_now[:, i], states[f], err_msg=f"step {k}: PCG64 {f}")
        episodes += int(fin.sum())
    r = w.ring
    np.testing.assert_array_equal(r.actions.cpu().numpy(), oring.actions)
    np.testing.assert_array_equal(r.dones.cpu().numpy(), oring.dones)
    np.testing.assert_array_equal(r.timeouts.cpu().numpy(), oring.timeouts)
    np.testing.assert_array_equal(r.observations.cpu().numpy()[..., :4], oring.observations)
    assert rel_err(r.next_observations.cpu().numpy()[..., :4], oring.next_observations, obs_floor) < tol
    assert rel_err(r.rewards.cpu().numpy(), oring.rewards, 1.0) < 4 * tol
    assert r.ctl.cpu().numpy().tolist() == [13 % rows, 1, 0, 13] and episodes > n  # every env finished at least once
    if d == 8:  # the raw half of the observation: denormalised new state (twoseriescstr.py:129-150)
        lo, hi = np.array([0.0, 273.15, 0.0, 273.15], np.float32), np.array([0.7, 400.0, 0.7, 400.0], np.float32)
        got = w.env_obs.cpu().numpy()
        raw = np.clip(lo + (got[:, :4] + np.float32(1.0)) * (hi - lo) / np.float32(2.0), lo, hi)
        assert rel_err(got[:, 4:], raw, 1.0) < 1e-6


@pytest.mark.parametrize("rows,n_envs,batch", [(244, 4096, 256), (7, 17, 1024), (3, 1, 100), (100000, 3, 256), (5, 1025, 4000)])
def test_one_wave_index_draw_is_numpys(ops, rows, n_envs, batch):
    """The rollout launch's index draw against numpy's legacy stream itself: RandomState(seed).randint(0, upper, B) followed by
    randint(0, n_envs, B) (core/common/buffers.py:113, :309), stream image and position included (n_envs = 1 consumes nothing),
    over enough steps that small rings fill and the 624-word block is twisted many times."""
    w = _World(ops, n_envs, 4, rows, batch, 64, 64, 1, seed=3, max_steps=50)
    seed = 99
    ops.mt19937_seed(w.mt, seed)
    rs = np.random.RandomState(seed)
    for k in range(12):
        w.step_fused(None)
        th.cuda.synchronize()
        upper = rows if k + 1 >= rows else k + 1
        bi = rs.randint(0, upper, size=batch)
        ei = rs.randint(0, n_envs, size=batch)
        np.testing.assert_array_equal(w.idx[0].cpu().numpy(), bi)
        np.testing.assert_array_equal(w.idx[1].cpu().numpy(), ei)
        st = rs.get_state(legacy=True)
        np.testing.assert_array_equal(w.mt[:624].cpu().numpy().view(np.uint32), st[1])
        assert int(w.mt[624]) == st[2]


def test_rollout_step_rejects_what_it_does_not_cover(ops):
    w = _World(ops, 64, 4, 3, 16, 64, 64, 0, seed=1, max_steps=9)
    with pytest.raises(Exception):  # no tile-major copy
        ops.rollout_step(w.env_obs, *w.w, 1, 0, 0, None, w.rng_ctl, w.coef, "euler", w.ring, w.env_obs, w.step_count, 1, w.low, w.high,
                         pcg_state=w.pcg)
    with pytest.raises(Exception):  # sampling head without its Philox stream
        ops.rollout_step(w.env_obs, *w.w, 1, 0, 0, w.swz, None, w.coef, "euler", w.ring, w.env_obs, w.step_count, 1, w.low, w.high,
                         pcg_state=w.pcg)
    with pytest.raises(Exception):  # index draw without its output buffer
        ops.rollout_step(w.env_obs, *w.w, 1, 0, 0, w.swz, w.rng_ctl, w.coef, "euler", w.ring, w.env_obs, w.step_count, 1, w.low, w.high,
                         pcg_state=w.pcg, mt_state=w.mt)
    with pytest.raises(Exception):  # no reset source
        ops.rollout_step(w.env_obs, *w.w, 1, 0, 0, w.swz, w.rng_ctl, w.coef, "euler", w.ring, w.env_obs, w.step_count, 1, w.low, w.high)


@pytest.mark.parametrize("algo,obs_dim,integrator", [("sac", 4, "euler"), ("td3", 4, "euler"), ("sac", 8, "rk4")])
def test_learn_with_and_without_the_one_launch_rollout(algo, obs_dim, integrator, monkeypatch):
    """learn() under hipGraph replay with the one-launch rollout and with the separate launches: identical weights, ring, sampler
    stream and env state after 40 iterations (the two forms draw the same numbers in the same order)."""
    from core.common import off_policy_algorithm as opa
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC
    from core.td3 import TD3

    def run(fused_rollout):
        monkeypatch.setattr(opa, "FUSED_ROLLOUT", fused_rollout)
        env = CSTRVecEnv(2048, obs_dim=obs_dim, integrator=integrator, device="cuda")
        model = (SAC if algo == "sac" else TD3)("MlpPolicy", env, seed=7, device="cuda", learning_starts=2048 * 2, buffer_size=2048 * 6)
        model.enable_graph_capture(True)
        model.learn(total_timesteps=2048 * 40)
        th.cuda.synchronize()
        assert model.graph_status()["active"] and model.graph_status()["replays"] > 20
        if fused_rollout:
            assert model._rollout_net() is not None  # the one-launch form really ran
        rb = model.replay_buffer
        flat = th.cat([p.detach().reshape(-1) for p in model.policy.parameters()])
        return flat.clone(), rb.observations.clone(), rb.actions.clone(), rb.rewards.clone(), rb.sampler_stream.clone(), env.obs.clone(), rb.ring.ctl.clone()

    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a, b)):
        assert th.equal(x, y), f"tensor {i} differs between the one-launch rollout and the separate launches"


@pytest.mark.parametrize("d,a,batch,n_out,both,act", [(4, 2, 256, 256, True, 1), (8, 2, 100, 64, True, 2), (4, 2, 300, 400, False, 1),
                                                     (8, 4, 64, 48, True, 0), (4, 2, 16, 20, False, 2)])
def test_first_layer_with_the_gather_inside(ops, d, a, batch, n_out, both, act):
    """cstr_linear_act_fwd_gather_f32 against cstr_replay_gather_packed_f32 followed by cstr_linear_act_fwd_f32 on the gathered
    rows: layer output, packed batch, ring and Philox control words bit-identical."""
    g = th.Generator(device="cuda").manual_seed(batch + n_out)
    rows, n = 9, 333
    ring_a, ring_b = ops.DeviceRing(rows, n, d, a, "cuda"), ops.DeviceRing(rows, n, d, a, "cuda")
    for ra, rb_ in ((ring_a.observations, ring_b.observations), (ring_a.next_observations, ring_b.next_observations),
                    (ring_a.actions, ring_b.actions), (ring_a.rewards, ring_b.rewards)):
        ra.copy_(th.randn(ra.shape, device="cuda", generator=g))
        rb_.copy_(ra)
    for ra, rb_ in ((ring_a.dones, ring_b.dones), (ring_a.timeouts, ring_b.timeouts)):
        ra.copy_((th.rand(ra.shape, device="cuda", generator=g) < 0.3).float())
        rb_.copy_(ra)
    for r in (ring_a, ring_b):
        r.ctl.copy_(th.tensor([rows - 1, 0, 0, rows - 1]))  # the advance wraps the position and sets `full`
    idx = th.stack([th.randint(0, rows, (batch,), device="cuda", generator=g), th.randint(0, n, (batch,), device="cuda", generator=g)]).int().contiguous()
    w1, b1 = th.randn(n_out, d, device="cuda", generator=g), th.randn(n_out, device="cuda", generator=g)
    ca, cb = ops.new_rng_ctl(5, "cuda"), ops.new_rng_ctl(5, "cuda")
    w = d + a
    mk = lambda: (th.full((batch, w), 7.0, device="cuda"), th.full((2 * batch, w), 7.0, device="cuda"), th.zeros(batch, 1, device="cuda"),  # noqa: E731
                  th.zeros(batch, 1, device="cuda"))
    xd_a, xpn_a, dn_a, rw_a = mk()
    xd_b, xpn_b, dn_b, rw_b = mk()
    ops.replay_gather_packed(ring_a, idx, batch, xd_a, xpn_a[batch:], xpn_a[:batch], dn_a, rw_a, advance_ring=True, rng_advance=(ca, 4096))
    x_in = xpn_a[:, :d] if both else xpn_a[batch:, :d]
    y_a = ops.linear_act_fwd(x_in, w1, b1, act)
    y_b = ops.linear_act_fwd_gather(ring_b, idx, batch, both, w1, b1, act, xd_b, xpn_b[batch:], xpn_b[:batch], dn_b, rw_b, advance_ring=True,
                                    rng_advance=(cb, 4096))
    th.cuda.synchronize()
    assert th.equal(y_a, y_b)
    assert th.equal(xd_a, xd_b) and th.equal(xpn_a, xpn_b) and th.equal(dn_a, dn_b) and th.equal(rw_a, rw_b)
    assert float(xpn_b[:, d:].min()) == 7.0  # the action columns of x_pi / x_next belong to the actor head
    assert th.equal(ring_a.ctl, ring_b.ctl) and ring_b.ctl.tolist() == [0, 1, 0, rows] and th.equal(ca, cb)


@pytest.mark.parametrize("algo", ["sac", "td3"])
def test_learn_with_the_gather_in_the_first_layer(algo, monkeypatch):
    """learn() under hipGraph replay: gather inside the first layer behind the sample (SAC: the 2B-row actor pass; TD3: the target
    actor) vs the gather launch in front of it: identical weights, sampler stream and ring after 40 iterations."""
    from core.common import fused
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC
    from core.td3 import TD3

    def run(flag):
        monkeypatch.setattr(fused, "USE_GATHER_IN_FIRST_LAYER", flag)
        env = CSTRVecEnv(2048, device="cuda")
        model = (SAC if algo == "sac" else TD3)("MlpPolicy", env, seed=11, device="cuda", learning_starts=2048 * 2, buffer_size=2048 * 6)
        model.enable_graph_capture(True)
        model.learn(total_timesteps=2048 * 40)
        th.cuda.synchronize()
        assert model.graph_status()["active"] and model.graph_status()["replays"] > 20
        rb = model.replay_buffer
        flat = th.cat([p.detach().reshape(-1) for p in model.policy.parameters()])
        return flat.clone(), rb.sampler_stream.clone(), rb.ring.ctl.clone(), rb.rewards.clone(), rb.actions.clone()

    for i, (x, y) in enumerate(zip(run(True), run(False))):
        assert th.equal(x, y), f"tensor {i} differs"


def test_td3_learn_with_the_smoothing_in_the_target_actors_last_layer(monkeypatch):
    """TD3 under hipGraph replay: target policy smoothing inside the target actor's last layer vs its own launch: identical weights,
    Philox stream and ring after 40 iterations."""
    from core.common import fused
    from core.common.vec_env import CSTRVecEnv
    from core.td3 import TD3

    def run(flag):
        monkeypatch.setattr(fused, "USE_SMOOTH_IN_LAST_LAYER", flag)
        env = CSTRVecEnv(2048, device="cuda")
        model = TD3("MlpPolicy", env, seed=13, device="cuda", learning_starts=2048 * 2, buffer_size=2048 * 6)
        model.enable_graph_capture(True)
        model.learn(total_timesteps=2048 * 40)
        th.cuda.synchronize()
        assert model.graph_status()["active"] and model.graph_status()["replays"] > 20
        flat = th.cat([p.detach().reshape(-1) for p in model.policy.parameters()])
        return flat.clone(), model._device_rng().clone(), model.replay_buffer.rewards.clone()

    for i, (x, y) in enumerate(zip(run(True), run(False))):
        assert th.equal(x, y), f"tensor {i} differs"
