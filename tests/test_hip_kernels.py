"""GPU parity tests: every HIP kernel, called through the C ABI, against the oracle and the golden
vectors generated from the reference. Bit-exact for integer/index/copy work; fp32 tolerances are
written next to each check (north_star: 1e-5 rel on Q-values/returns; bit-exact replay indices).
"""
import numpy as np
import pytest
import torch as th

from conftest import rel_err
from oracle import cstr_oracle as orc

pytestmark = pytest.mark.gpu

OBS_FLOOR = 1.0  # observations/rewards: error relative to max(|x|, 1) (see tests/test_oracle_env.py)


@pytest.fixture(scope="module")
def ops():
    assert th.cuda.is_available(), "GPU tests need an MI355X"
    from core import _native as nv
    from core.common import hip_ops

    nv.lib()  # fail loudly if the HIP extension is missing
    return hip_ops


def dev(a, dtype=None):
    t = th.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def obs8(o4):
    """[normalised | raw] observation of SURVEY D2 built on the host with the reference's f32 formula."""
    lo = np.array([0.0, 273.15, 0.0, 273.15], np.float32)
    hi = np.array([0.7, 400.0, 0.7, 400.0], np.float32)
    raw = np.clip(lo + (o4 + np.float32(1.0)) * (hi - lo) / np.float32(2.0), lo, hi).astype(np.float32)
    return np.concatenate([o4, raw], axis=1)


def run_vec_step(ops, obs, act, steps, reset_obs, integrator="euler", d=4):
    from core import _native as nv

    n = len(obs)
    o = dev(obs if d == 4 else obs8(obs))
    ro = dev(reset_obs if d == 4 else obs8(reset_obs))
    a, st = dev(act), dev(steps, th.int32)
    nxt, after = th.empty_like(o), th.empty_like(o)
    rew, done, tout = (th.empty(n, dtype=th.float32, device="cuda") for _ in range(3))
    ops.vec_step(nv.default_coef(), integrator, o, a, st, ro, nxt, after, rew, done, tout)
    th.cuda.synchronize()
    return [x.cpu().numpy() for x in (nxt, after, rew, done, tout, st)]


# ------------------------------------------------------------------------------------------- env
@pytest.mark.parametrize("d", [4, 8])
def test_vec_step_vs_golden_and_oracle(ops, golden, d):
    g = golden("env_step_kat.npz")
    rng = np.random.default_rng(1)
    reset = rng.uniform(-1, 1, g["obs"].shape).astype(np.float32)
    nxt, after, rew, done, tout, st = run_vec_step(ops, g["obs"], g["act"], g["step_in"], reset, d=d)
    # vs the reference's own outputs (golden): fp32 tolerance 1e-6 of the box scale, rewards 2e-6
    assert rel_err(nxt[:, :4], g["obs_next"], OBS_FLOOR) < 1e-6
    assert rel_err(rew, g["reward"], 1.0) < 2e-6
    np.testing.assert_array_equal(done.astype(np.uint8), g["truncated"])
    np.testing.assert_array_equal(tout.astype(np.uint8), g["truncated"])
    np.testing.assert_array_equal(st, np.where(g["truncated"] > 0, 0, g["step_in"] + 1))
    # vs the oracle on the same inputs: same rounding points, only expf may differ by an ulp
    o_nxt, o_after, o_rew, o_done, o_tout, o_st = orc.vec_step(g["obs"], g["act"], g["step_in"], reset)
    assert rel_err(nxt[:, :4], o_nxt, OBS_FLOOR) < 5e-7
    assert rel_err(after[:, :4], o_after, OBS_FLOOR) < 5e-7
    fin = g["truncated"].astype(bool)
    np.testing.assert_array_equal(after[fin][:, :4], reset[fin])  # reset source copied bit-exactly
    if d == 8:
        assert rel_err(nxt[:, 4:], g["raw_next"], 1.0) < 1e-6  # info["original_state"] (twoseriescstr.py:446)
        np.testing.assert_array_equal(after[fin], obs8(reset)[fin])


def test_vec_step_nan_action(ops, golden):
    g = golden("env_nan_kat.npz")
    reset = np.full((1, 4), 0.25, np.float32)
    nxt, after, rew, done, tout, st = run_vec_step(ops, g["obs"][None], g["act"][None], np.array([7], np.int32), reset)
    np.testing.assert_array_equal(nxt[0], g["obs_next"])
    assert rew[0] == -10.0 and done[0] == 1.0 and tout[0] == 1.0 and st[0] == 0
    np.testing.assert_array_equal(after, reset)


def test_trajectories_vs_golden(ops, golden):
    """400-step roll-outs under fixed action tapes: 1e-5 (north_star bound on returns)."""
    g = golden("env_traj_kat.npz")
    obs, steps = g["obs0"].copy(), np.zeros(len(g["obs0"]), np.int32)
    T = g["actions"].shape[0]
    ret = np.zeros(len(obs), np.float64)
    for k in range(T):
        nxt, after, rew, done, tout, steps = run_vec_step(ops, obs, g["actions"][k], steps, obs)
        assert rel_err(nxt, g["obs"][k], OBS_FLOOR) < 1e-5
        assert rel_err(rew, g["reward"][k], 1.0) < 1e-5
        np.testing.assert_array_equal(tout.astype(np.uint8), g["truncated"][k])
        ret += rew
        obs = nxt
    assert rel_err(ret, g["reward"].astype(np.float64).sum(0), 1e-3) < 1e-5  # episode returns


def test_autoreset_semantics_vs_golden(ops, golden):
    g = golden("vecenv_autoreset_kat.npz")
    obs, steps = g["obs0"].copy(), g["step0"].copy()
    for k in range(g["actions"].shape[0]):
        nxt, after, rew, done, tout, steps = run_vec_step(ops, obs, g["actions"][k], steps, g["reset_obs"][k])
        assert rel_err(nxt, g["next_obs_for_buffer"][k], OBS_FLOOR) < 1e-6
        assert rel_err(after, g["obs"][k], OBS_FLOOR) < 1e-6
        np.testing.assert_array_equal(done.astype(np.uint8), g["done"][k])
        np.testing.assert_array_equal(tout.astype(np.uint8), g["timeout"][k])
        obs = after


@pytest.mark.parametrize("n", [1, 63, 4096, 70001])
def test_vec_step_rk4_and_ragged_sizes(ops, n):
    rng = np.random.default_rng(n)
    obs = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    act = rng.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)
    steps = rng.integers(0, 400, n).astype(np.int32)
    reset = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    for integ in ("euler", "rk4"):
        got = run_vec_step(ops, obs, act, steps, reset, integrator=integ)
        exp = orc.vec_step(obs, act, steps, reset, integrator=integ)
        # RK4 chains four expf-bearing stages: allow 2e-6 of the box scale
        assert rel_err(got[0], exp[0], OBS_FLOOR) < 2e-6 and rel_err(got[1], exp[1], OBS_FLOOR) < 2e-6
        assert rel_err(got[2], exp[2], 1.0) < 4e-6
        for a, b in zip(got[3:], exp[3:]):
            np.testing.assert_array_equal(a, b)


def test_reset_draw_pcg64_bit_exact(ops):
    """Device PCG64 + Generator.uniform restatement == oracle == numpy Generator (see test_oracle_elementwise)."""
    n = 1000
    st = orc.pcg64_states_from_seeds(np.arange(100, 100 + n))
    dst = dev(st.view(np.uint64).reshape(n, 4).view(np.int64))
    for rnd in range(3):
        mask = None if rnd == 0 else (np.arange(n) % (rnd + 1) == 0).astype(np.uint8)
        out = th.zeros(n, 4, device="cuda")
        ops.reset_draw(dst, None if mask is None else dev(mask), out)
        exp = orc.reset_draw(st, mask)
        got = out.cpu().numpy()
        sel = slice(None) if mask is None else mask.astype(bool)
        np.testing.assert_array_equal(got[sel], exp[sel])
        np.testing.assert_array_equal(dst.cpu().numpy().view(np.uint64).reshape(-1), st.view(np.uint64).reshape(-1))


# ---------------------------------------------------------------------------------------- replay
def _mk_ring(ops, R, N, D, A=2):
    return ops.DeviceRing(R, N, D, A, "cuda")


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_replay_add_sample_vs_golden(ops, golden, tag):
    """Reference ReplayBuffer.add/sample round trip incl. wrap-around and dones*(1-timeouts): bit-exact."""
    g = golden("replay_kat.npz")
    R, N, D, A, n_add, B = (int(x) for x in g[f"{tag}_dims"])
    ring = _mk_ring(ops, R, N, D)
    mt = th.zeros(628, dtype=th.int32, device="cuda")
    ops.mt19937_seed(mt, int(g["seed"]))
    outs = [th.empty(B, D, device="cuda"), th.empty(B, A, device="cuda"), th.empty(B, D, device="cuda"),
            th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")]
    for k in range(n_add):
        ops.replay_add(ring, dev(g[f"{tag}_obs"][k]), dev(g[f"{tag}_next_obs"][k]), dev(g[f"{tag}_act"][k]),
                       dev(g[f"{tag}_rew"][k]), dev(g[f"{tag}_done"][k], th.float32), dev(g[f"{tag}_timeout"][k], th.float32))
        ops.replay_sample(ring, mt, B, *outs)
        for name, t in zip(("observations", "actions", "next_observations", "dones", "rewards"), outs):
            np.testing.assert_array_equal(t.cpu().numpy(), g[f"{tag}_s_{name}"][k], err_msg=f"{tag} add#{k} {name}")
    for name, t in (("obs", ring.observations), ("next_obs", ring.next_observations), ("act", ring.actions),
                    ("rew", ring.rewards), ("done", ring.dones), ("timeout", ring.timeouts)):
        np.testing.assert_array_equal(t.cpu().numpy(), g[f"{tag}_ring_{name}"])
    ctl = ring.ctl.cpu().numpy()
    assert ctl[0] == int(g[f"{tag}_pos"]) and ctl[1] == int(g[f"{tag}_full"]) and ctl[2] == 0 and ctl[3] == n_add


def test_mt19937_index_stream_vs_numpy_golden(ops, golden):
    """np.random.randint index streams (buffers.py:113,309), incl. upper == 1 (consumes nothing),
    non-power-of-two bounds and back-to-back calls: int64 indices and final (key, pos) bit-exact."""
    g = golden("mt19937_randint_kat.npz")
    for ci in range(int(g["n_cases"])):
        calls = g[f"c{ci}_calls"]
        if len(calls) % 2 or calls.max() >= 2**32 - 1 or (calls[0::2, 1] != calls[1::2, 1]).any():
            continue  # the device sampler draws (rows, envs) pairs of equal batch on the 32-bit path
        mt = th.zeros(628, dtype=th.int32, device="cuda")
        ops.mt19937_seed(mt, int(g[f"c{ci}_seed"]))
        got = []
        for (upper, B), (n_envs, _) in zip(calls[0::2], calls[1::2]):
            upper, B, n_envs = int(upper), int(B), int(n_envs)
            ring = _mk_ring(ops, max(upper, 1), n_envs, 4) if upper * n_envs <= 1 << 22 else None
            if ring is None:
                got = None
                break
            ring.ctl[0] = upper % ring.rows
            ring.ctl[1] = int(upper == ring.rows)
            bi, ei = th.empty(B, dtype=th.int64, device="cuda"), th.empty(B, dtype=th.int64, device="cuda")
            outs = [th.empty(B, 4, device="cuda"), th.empty(B, 2, device="cuda"), th.empty(B, 4, device="cuda"),
                    th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")]
            ops.replay_sample(ring, mt, B, *outs, bi, ei)
            got += [bi.cpu().numpy(), ei.cpu().numpy()]
        if got is None:
            continue
        np.testing.assert_array_equal(np.concatenate(got), g[f"c{ci}_out"], err_msg=f"case {ci}")
        st = mt.cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(st[:624], g[f"c{ci}_key"])
        assert int(st[624]) == int(g[f"c{ci}_pos"])


@pytest.mark.parametrize("seed,R,N,B", [(0, 244, 4096, 256), (4095, 244, 4096, 256), (7, 976, 1024, 256),
                                        (11, 3, 5, 1), (12, 1000, 7, 4099), (13, 61, 33, 16384)])
def test_sampler_vs_oracle_random_fill(ops, seed, R, N, B):
    """Seeded random rings at BASELINE sizes: device (indices, gathered batch, MT state) == oracle, 20 calls."""
    rng = np.random.default_rng(seed)
    D = 4
    ring, oring = _mk_ring(ops, R, N, D), orc.ReplayRing(R, N, D, 2)
    n_add = min(R + 3, 40)
    for k in range(n_add):
        f = [rng.uniform(-1, 1, (N, D)).astype(np.float32), rng.uniform(-1, 1, (N, D)).astype(np.float32),
             rng.uniform(-1, 1, (N, 2)).astype(np.float32), rng.uniform(-8, 0, N).astype(np.float32),
             (rng.uniform(size=N) < 0.3).astype(np.float32), (rng.uniform(size=N) < 0.2).astype(np.float32)]
        oring.add(*f)
        ops.replay_add(ring, *[dev(x) for x in f])
    mt, omt = th.zeros(628, dtype=th.int32, device="cuda"), orc.MT19937(seed + N - 1)
    ops.mt19937_seed(mt, seed + N - 1)  # effective stream seed of an end-to-end run (SURVEY a-6)
    bi, ei = th.empty(B, dtype=th.int64, device="cuda"), th.empty(B, dtype=th.int64, device="cuda")
    outs = [th.empty(B, D, device="cuda"), th.empty(B, 2, device="cuda"), th.empty(B, D, device="cuda"),
            th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")]
    for call in range(20):
        ops.replay_sample(ring, mt, B, *outs, bi, ei)
        exp, (ebi, eei) = oring.sample(omt, B)
        np.testing.assert_array_equal(bi.cpu().numpy(), ebi, err_msg=f"row idx, call {call}")
        np.testing.assert_array_equal(ei.cpu().numpy(), eei, err_msg=f"env idx, call {call}")
        for t, e in zip(outs, exp):
            np.testing.assert_array_equal(t.cpu().numpy(), e)
    st = mt.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(st[:624], omt.key)
    assert int(st[624]) == omt.pos


# --------------------------------------------------------------------------------- fused collect
@pytest.mark.parametrize("d,integ", [(4, "euler"), (8, "euler"), (4, "rk4")])
def test_collect_step_fused_equals_unfused_oracle(ops, d, integ):
    """Fused scale-chain + step + auto-reset + ring write over 3 ring wraps, episodes ending at different
    times, vs the oracle's action chain -> vec_step -> ring add. Index/flag/copy fields bit-exact."""
    from core import _native as nv

    rng = np.random.default_rng(5)
    N, R, T = 777, 4, 13
    low, high = np.array([-1, -1], np.float32), np.array([1, 1], np.float32)
    obs = rng.uniform(-1, 1, (N, 4)).astype(np.float32)
    steps = rng.integers(380, 400, N).astype(np.int32)
    ring, oring = _mk_ring(ops, R, N, d), orc.ReplayRing(R, N, 4, 2)
    env_obs = dev(obs if d == 4 else obs8(obs))
    dsteps = dev(steps, th.int32)
    rew_o, done_o = th.empty(N, device="cuda"), th.empty(N, device="cuda")
    coef = nv.default_coef()
    for k in range(T):
        pol = np.tanh(rng.normal(0, 1.5, (N, 2))).astype(np.float32)
        noise = None if k % 2 else rng.normal(0, 0.1, (N, 2)).astype(np.float32)
        reset = rng.uniform(-1, 1, (N, 4)).astype(np.float32)
        squashed = k % 3 != 0
        ops.collect_step(coef, integ, ring, env_obs, dsteps, dev(pol), squashed, low, high,
                         noise=None if noise is None else dev(noise), reset_obs=dev(reset if d == 4 else obs8(reset)),
                         reward_out=rew_o, done_out=done_o)
        buf_a, env_a = orc.action_scale_chain(pol, squashed, low, high)
        if noise is not None:
            buf_a = np.clip(buf_a + noise, -1, 1).astype(np.float32)
            env_a = (low + (np.float32(0.5) * (buf_a + np.float32(1.0)) * (high - low))).astype(np.float32)
        nxt, after, rew, done, tout, steps2 = orc.vec_step(obs, env_a, steps, reset, integrator=integ)
        oring.add(obs, nxt, buf_a, rew, done, tout)
        tol = 5e-7 if integ == "euler" else 2e-6
        assert rel_err(env_obs.cpu().numpy()[:, :4], after, OBS_FLOOR) < tol
        np.testing.assert_array_equal(dsteps.cpu().numpy(), steps2)
        np.testing.assert_array_equal(done_o.cpu().numpy(), done)
        assert rel_err(rew_o.cpu().numpy(), rew, 1.0) < 4 * tol
        # follow the DEVICE trajectory so single-ulp expf differences cannot accumulate into flag flips
        obs, steps = env_obs.cpu().numpy()[:, :4].copy(), steps2
    tol = 5e-7 if integ == "euler" else 2e-6
    np.testing.assert_array_equal(ring.actions.cpu().numpy(), oring.actions)
    np.testing.assert_array_equal(ring.dones.cpu().numpy(), oring.dones)
    np.testing.assert_array_equal(ring.timeouts.cpu().numpy(), oring.timeouts)
    np.testing.assert_array_equal(ring.observations.cpu().numpy()[..., :4], oring.observations)
    assert rel_err(ring.next_observations.cpu().numpy()[..., :4], oring.next_observations, OBS_FLOOR) < tol
    assert rel_err(ring.rewards.cpu().numpy(), oring.rewards, 1.0) < 4 * tol
    ctl = ring.ctl.cpu().numpy()
    assert ctl[0] == T % R and ctl[1] == 1 and ctl[2] == 0 and ctl[3] == T
    assert oring.dones.sum() > 0  # auto-reset was exercised


def test_collect_step_device_reset_rng(ops):
    """pcg_state reset source: finished envs restart from the faithful PCG64 draw, others untouched."""
    from core import _native as nv

    N, R = 300, 2
    rng = np.random.default_rng(0)
    obs = rng.uniform(-1, 1, (N, 4)).astype(np.float32)
    steps = np.where(np.arange(N) % 3 == 0, 399, 5).astype(np.int32)
    st = orc.pcg64_states_from_seeds(np.arange(N))
    dst = dev(st.view(np.uint64).reshape(N, 4).view(np.int64))
    ring = _mk_ring(ops, R, N, 4)
    env_obs, dsteps = dev(obs), dev(steps, th.int32)
    pol = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
    ep_ret, ep_stats = dev(np.full(N, -3.0, np.float32)), th.zeros(4, dtype=th.float64, device="cuda")
    ops.collect_step(nv.default_coef(), "euler", ring, env_obs, dsteps, dev(pol), True,
                     [-1, -1], [1, 1], pcg_state=dst, ep_return=ep_ret, ep_stats=ep_stats)
    fin = steps == 399
    # device-side episode statistics (Monitor semantics): count, sum of returns, sum of lengths
    rew0 = ring.rewards.cpu().numpy()[0].astype(np.float64)
    es = ep_stats.cpu().numpy()
    assert es[0] == fin.sum() and es[2] == 400 * fin.sum()
    assert abs(es[1] - (rew0[fin] - 3.0).sum()) < 1e-3
    np.testing.assert_allclose(ep_ret.cpu().numpy(), np.where(fin, 0.0, rew0 - 3.0), rtol=1e-6)
    exp_reset = orc.reset_draw(st, fin.astype(np.uint8))
    got = env_obs.cpu().numpy()
    np.testing.assert_array_equal(got[fin], exp_reset[fin])
    np.testing.assert_array_equal(dst.cpu().numpy().view(np.uint64).reshape(-1), st.view(np.uint64).reshape(-1))
    np.testing.assert_array_equal(dsteps.cpu().numpy(), np.where(fin, 0, 6))
    np.testing.assert_array_equal(ring.dones.cpu().numpy()[0], fin.astype(np.float32))


def test_static_init_mode_reset_vs_reference_golden(ops, golden):
    """init_mode="static" (twoseriescstr.py:94-96, :246-255): seeded first reset then continuing resets, observations and
    the drifting f64 init_state bit-identical to the reference run; "random" mode against the same fixture."""
    g = golden("env_reset_kat.npz")
    seeds = g["seeds"]
    n = len(seeds)
    for mode in ("random", "static"):
        st = orc.pcg64_states_from_seeds(seeds)
        dst = dev(st.view(np.uint64).reshape(n, 4).view(np.int64))
        init = th.tensor(orc.STATIC_INIT_STATE, dtype=th.float64, device="cuda").repeat(n, 1).contiguous() if mode == "static" else None
        out = th.zeros(n, 4, device="cuda")
        for k in range(g[f"{mode}_obs"].shape[1]):
            ops.reset_draw(dst, None, out, static_init=init)
            np.testing.assert_array_equal(out.cpu().numpy(), g[f"{mode}_obs"][:, k])
            if init is not None:
                np.testing.assert_array_equal(init.cpu().numpy(), g["static_init_state"][:, k])


def test_static_init_mode_through_fused_collect_and_vec_env(ops, golden):
    """The reference's DummyVecEnv run with init_mode="static" (auto-reset draws continue each env's stream and init_state)
    replayed through CSTRVecEnv.step AND through the fused collect kernel."""
    from core import _native as nv
    from core.common.vec_env import CSTRVecEnv

    g = golden("env_reset_kat.npz")
    n = g["vec_obs0"].shape[0]
    for fused in (False, True):
        env = CSTRVecEnv(n, init_mode="static")
        env.seed(int(g["vec_seed"]))
        np.testing.assert_array_equal(env.reset(), g["vec_obs0"])
        env.step_count.copy_(dev(g["vec_step0"], th.int32))
        ring = _mk_ring(ops, 8, n, 4)
        for k in range(g["vec_actions"].shape[0]):
            if fused:
                ops.collect_step(env.coef, "euler", ring, env.obs, env.step_count, dev(g["vec_actions"][k]), True, [-1, -1], [1, 1],
                                 pcg_state=env.pcg_state, static_init=env.static_init, done_out=env._done)
                obs, done = env.obs.cpu().numpy(), env._done.cpu().numpy().astype(np.uint8)
            else:
                obs, _, done, _ = env.step(g["vec_actions"][k])
            np.testing.assert_array_equal(done.astype(np.uint8), g["vec_done"][k])
            d = g["vec_done"][k].astype(bool)
            np.testing.assert_array_equal(obs[d], g["vec_obs"][k][d])  # reset draws: bit-exact
            assert rel_err(obs, g["vec_obs"][k], OBS_FLOOR) < 1e-6
        np.testing.assert_array_equal(env.static_init.cpu().numpy(), g["vec_init_state"])
    with pytest.raises(ValueError):
        CSTRVecEnv(2, init_mode="bogus")


# --------------------------------------------------------------------------------- element-wise
@pytest.mark.parametrize("n", [1, 3, 256, 135682, 369704, 1 << 22])
def test_polyak_bit_exact(ops, n):
    """core/common/utils.py:478-481 on one flat arena: bit-identical to torch's two in-place ops."""
    g = th.Generator().manual_seed(n)
    p, t = th.randn(n, generator=g), th.randn(n, generator=g)
    for tau in (0.005, 1.0):
        exp = t.clone()
        exp.mul_(1 - tau)
        th.add(exp, p, alpha=tau, out=exp)
        dp, dt = p.cuda(), t.clone().cuda()
        ops.polyak(dp, dt, tau)
        np.testing.assert_array_equal(dt.cpu().numpy(), exp.numpy())
        np.testing.assert_array_equal(dt.cpu().numpy(), orc.polyak(p.numpy(), t.numpy(), tau))


@pytest.mark.parametrize("n,sac", [(1, True), (256, True), (256, False), (100003, True)])
def test_td_target_bit_exact(ops, n, sac):
    g = th.Generator().manual_seed(n)
    q1, q2, lp = th.randn(n, 1, generator=g) * 5, th.randn(n, 1, generator=g) * 5, th.randn(n, 1, generator=g)
    rew, done = -th.rand(n, 1, generator=g) * 8, (th.rand(n, 1, generator=g) < 0.2).float()
    ent = th.tensor([0.731])
    out = th.empty(n, 1, device="cuda")
    ops.td_target_min(q1.cuda(), q2.cuda(), lp.cuda() if sac else None, rew.cuda(), done.cuda(),
                      ent.cuda() if sac else None, 0.99, out)
    exp = orc.td_target_min(q1.numpy(), q2.numpy(), lp.numpy() if sac else None, rew.numpy(), done.numpy(), float(ent), 0.99)
    np.testing.assert_array_equal(out.cpu().numpy().reshape(-1), exp)
    # and the reference's torch expression (sac.py:250-254 / td3.py:174-176) evaluated on the CPU
    nq, _ = th.min(th.cat((q1, q2), dim=1), dim=1, keepdim=True)
    if sac:
        nq = nq - ent * lp.reshape(-1, 1)
    np.testing.assert_array_equal(out.cpu().numpy(), (rew + (1 - done) * 0.99 * nq).numpy())


@pytest.mark.parametrize("n", [5, 1026, 68100, 135682])
def test_adam_vs_torch_and_oracle(ops, n):
    """torch.optim.Adam (reference optimiser) for 6 steps with a device-resident step counter."""
    from core import _native as nv

    g = th.Generator().manual_seed(n)
    p0 = th.randn(n, generator=g)
    ref = th.nn.Parameter(p0.clone())
    opt = th.optim.Adam([ref], lr=3e-4)
    p, m, v = p0.clone().cuda(), th.zeros(n, device="cuda"), th.zeros(n, device="cuda")
    ctl = ops.new_adam_ctl("cuda")
    lr = th.tensor([3e-4], dtype=th.float64, device="cuda")
    op, om, ov = p0.numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for step in range(1, 7):
        grad = th.randn(n, generator=g) * (10.0 ** float(th.randint(-3, 2, (1,), generator=g)))
        ref.grad = grad.clone()
        opt.step()
        ops.adam(p, grad.cuda(), m, v, ctl, lr)
        op, om, ov = orc.adam_step(op, grad.numpy(), om, ov, step, 3e-4)
        assert int(ctl[0]) == step and int(ctl[1]) == 0
        # device pow()/sqrt() in f64 may differ from libm by an ulp before the f32 cast: <= 2 ulp, and
        # 1e-6 relative against torch
        for got, want in ((p, op), (m, om), (v, ov)):
            u = np.abs(got.cpu().numpy().view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
            assert u.max() <= 4 and (u > 1).mean() < 1e-3
        assert rel_err(p.cpu().numpy(), ref.detach().numpy(), 1e-3) < 1e-6
    # grad_scale (data-parallel mean) folds 1/W into the update
    p2, m2, v2 = p0.clone().cuda(), th.zeros(n, device="cuda"), th.zeros(n, device="cuda")
    ctl2 = ops.new_adam_ctl("cuda")
    p3, m3, v3 = p0.clone().cuda(), th.zeros(n, device="cuda"), th.zeros(n, device="cuda")
    ctl3 = ops.new_adam_ctl("cuda")
    gsum = th.randn(n, generator=g).cuda()
    ops.adam(p2, gsum, m2, v2, ctl2, lr, grad_scale=0.125)
    ops.adam(p3, gsum * 0.125, m3, v3, ctl3, lr)
    assert th.equal(p2, p3) and th.equal(m2, m3) and th.equal(v2, v3)


def test_ops_reject_bad_shapes(ops):
    from core import _native as nv

    n = 64
    o = th.zeros(n, 4, device="cuda")
    with pytest.raises(ValueError):
        ops.vec_step(nv.default_coef(), "euler", o, th.zeros(n, 3, device="cuda"), th.zeros(n, dtype=th.int32, device="cuda"),
                     o, o.clone(), o.clone(), *(th.zeros(n, device="cuda") for _ in range(3)))
    with pytest.raises(ValueError):
        ops.DeviceRing(4, 4, 5, 2, "cuda")
    ring = ops.DeviceRing(4, 4, 4, 2, "cuda")
    with pytest.raises(ValueError):
        ops.replay_sample(ring, th.zeros(628, dtype=th.int32, device="cuda"), 1 << 15,
                          *(th.zeros(1 << 15, k, device="cuda") for k in (4, 2, 4, 1, 1)))


# ------------------------------------------------------------------------- twin-train layout (8 obs / 4 act)
def _twin_expected(obs8, act4, steps, reset8, integ="euler"):
    """Two independent reference trains per env, composed from the single-train oracle."""
    A = orc.vec_step(obs8[:, :4], act4[:, :2], steps, reset8[:, :4], integrator=integ)
    B = orc.vec_step(obs8[:, 4:], act4[:, 2:], steps, reset8[:, 4:], integrator=integ)
    nxt = np.concatenate([A[0], B[0]], 1)
    done = np.maximum(A[3], B[3])
    after = np.where(done[:, None] > 0, reset8, nxt)
    return nxt, after, (A[2] + B[2]).astype(np.float32), done, done.copy(), np.where(done > 0, 0, steps + 1).astype(np.int32)


def test_twin_layout_vec_step_collect_and_sampler(ops):
    from core import _native as nv

    rng = np.random.default_rng(21)
    N, R, T = 333, 3, 7
    obs = rng.uniform(-1, 1, (N, 8)).astype(np.float32)
    steps = rng.integers(392, 400, N).astype(np.int32)
    coef = nv.default_coef()
    # unfused VecEnv.step
    act = rng.uniform(-1.3, 1.3, (N, 4)).astype(np.float32)
    reset = rng.uniform(-1, 1, (N, 8)).astype(np.float32)
    o, a, st, ro = dev(obs), dev(act), dev(steps, th.int32), dev(reset)
    nxt, after = th.empty_like(o), th.empty_like(o)
    rew, done, tout = (th.empty(N, device="cuda") for _ in range(3))
    ops.vec_step(coef, "euler", o, a, st, ro, nxt, after, rew, done, tout)
    e = _twin_expected(obs, act, steps, reset)
    assert rel_err(nxt.cpu().numpy(), e[0], OBS_FLOOR) < 5e-7 and rel_err(after.cpu().numpy(), e[1], OBS_FLOOR) < 5e-7
    assert rel_err(rew.cpu().numpy(), e[2], 1.0) < 2e-6
    np.testing.assert_array_equal(done.cpu().numpy(), e[3])
    np.testing.assert_array_equal(st.cpu().numpy(), e[5])
    # fused collect into a (8, 4) ring, then the sampler gathers float4 actions
    ring, oring = ops.DeviceRing(R, N, 8, 4, "cuda"), orc.ReplayRing(R, N, 8, 4)
    env_obs, dsteps = dev(obs), dev(steps, th.int32)
    low, high = -np.ones(4, np.float32), np.ones(4, np.float32)
    cur, cst = obs.copy(), steps.copy()
    for k in range(T):
        pol = np.tanh(rng.normal(0, 1.5, (N, 4))).astype(np.float32)
        reset = rng.uniform(-1, 1, (N, 8)).astype(np.float32)
        mode = 1 if k % 2 else 3  # single-agent chain / multi-agent passthrough
        ops.collect_step(coef, "euler", ring, env_obs, dsteps, dev(pol), mode, low, high, reset_obs=dev(reset))
        if mode == 1:
            buf_a, env_a = orc.action_scale_chain(pol, True, low, high)
        else:
            buf_a = env_a = (low + (np.float32(0.5) * (pol + np.float32(1.0)) * (high - low))).astype(np.float32)
        e = _twin_expected(cur, env_a, cst, reset)
        oring.add(cur, e[0], buf_a, e[2], e[3], e[4])
        assert rel_err(env_obs.cpu().numpy(), e[1], OBS_FLOOR) < 5e-7
        np.testing.assert_array_equal(dsteps.cpu().numpy(), e[5])
        cur, cst = env_obs.cpu().numpy().copy(), e[5]
    np.testing.assert_array_equal(ring.actions.cpu().numpy(), oring.actions)
    np.testing.assert_array_equal(ring.observations.cpu().numpy(), oring.observations)
    np.testing.assert_array_equal(ring.dones.cpu().numpy(), oring.dones)
    assert rel_err(ring.next_observations.cpu().numpy(), oring.next_observations, OBS_FLOOR) < 5e-7
    assert oring.dones.sum() > 0
    # make the oracle ring bit-identical to the device ring, then sample both
    oring.next_observations[...] = ring.next_observations.cpu().numpy()
    oring.rewards[...] = ring.rewards.cpu().numpy()
    mt, omt = th.zeros(628, dtype=th.int32, device="cuda"), orc.MT19937(77)
    ops.mt19937_seed(mt, 77)
    B = 200
    outs = [th.empty(B, 8, device="cuda"), th.empty(B, 4, device="cuda"), th.empty(B, 8, device="cuda"),
            th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")]
    for _ in range(3):
        ops.replay_sample(ring, mt, B, *outs)
        exp, _ = oring.sample(omt, B)
        for t, x in zip(outs, exp):
            np.testing.assert_array_equal(t.cpu().numpy(), x)


@pytest.mark.parametrize("seed,act_dim,n_envs", [(0, 2, 4096), (7, 2, 1), (42, 3, 5), (123, 1, 33), (2**32 - 1, 4, 1024), (5, 2, 77)])
def test_mt19937_legacy_normal_interleaved_with_sampler_vs_numpy(ops, seed, act_dim, n_envs):
    """Exploration noise from the legacy global stream (noise.py:44-45, :141-142) interleaved with ReplayBuffer.sample's
    index draws (buffers.py:113, :309), checked live against numpy's RandomState: f32 noise values, indices, the cached
    second deviate of odd counts and the final key / position are all bit-identical."""
    rs = np.random.RandomState(seed)
    mt = th.zeros(628, dtype=th.int32, device="cuda")
    ops.mt19937_seed(mt, seed)
    mu, sigma = np.linspace(-0.5, 0.5, act_dim), np.linspace(0.1, 0.3, act_dim)
    R, B = 16, 64
    ring = _mk_ring(ops, R, n_envs, 4)
    z = th.zeros(n_envs, 4, device="cuda")
    ops.replay_add(ring, z, z, th.zeros(n_envs, 2, device="cuda"), th.zeros(n_envs, device="cuda"), th.zeros(n_envs, device="cuda"),
                   th.zeros(n_envs, device="cuda"))
    ring.ctl[0], ring.ctl[1] = 0, 1  # full ring: upper = R
    outs = [th.empty(B, 4, device="cuda"), th.empty(B, 2, device="cuda"), th.empty(B, 4, device="cuda"),
            th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")]
    ri, ei = th.empty(B, dtype=th.int64, device="cuda"), th.empty(B, dtype=th.int64, device="cuda")
    out = th.empty(n_envs, act_dim, device="cuda")
    for it in range(6):
        ops.mt19937_normal(mt, mu, sigma, out)
        want = np.stack([rs.normal(mu, sigma).astype(np.float32) for _ in range(n_envs)])
        np.testing.assert_array_equal(out.cpu().numpy(), want, err_msg=f"noise draw {it}")
        ops.replay_sample(ring, mt, B, *outs, ri, ei)
        np.testing.assert_array_equal(ri.cpu().numpy(), rs.randint(0, R, size=B))
        np.testing.assert_array_equal(ei.cpu().numpy(), rs.randint(0, n_envs, size=B))
    st, w = rs.get_state(), mt.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(w[:624], st[1])
    assert (int(w[624]), int(w[625])) == (st[2], st[3])
    assert abs(float(w[626:628].view(np.float64)[0]) - st[4]) <= 4e-16 * abs(st[4])


def test_mt19937_normal_rejects_bad_arguments(ops):
    mt = th.zeros(628, dtype=th.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.mt19937_normal(mt, [0.0] * 9, [1.0] * 9, th.empty(4, 9, device="cuda"))
    with pytest.raises(ValueError):
        ops.mt19937_normal(mt, [0.0, 0.0], [1.0, -1.0], th.empty(4, 2, device="cuda"))
    with pytest.raises(ValueError):
        ops.mt19937_normal(th.zeros(625, dtype=th.int32, device="cuda"), [0.0], [1.0], th.empty(4, 1, device="cuda"))


@pytest.mark.parametrize("D,A", [(4, 2), (8, 2), (8, 4)])
def test_packed_sampler_equals_plain_sampler(ops, D, A):
    """cstr_replay_sample_packed_mt19937_f32 draws the same indices and lays the same values out as (obs | act),
    (next_obs | .), (obs | .) rows -- ReplayBuffer.sample + the critics' cat([obs, actions], 1) (policies.py:975-981)."""
    R, N, B = 7, 50, 96
    ring = _mk_ring(ops, R, N, D, A)
    g = th.Generator(device="cuda").manual_seed(D * 10 + A)
    for t in (ring.observations, ring.next_observations, ring.actions, ring.rewards):
        t.copy_(th.randn(t.shape, device="cuda", generator=g))
    ring.dones.copy_((th.rand(R, N, device="cuda", generator=g) < 0.3).float())
    ring.timeouts.copy_((th.rand(R, N, device="cuda", generator=g) < 0.5).float() * ring.dones)
    ring.ctl[0], ring.ctl[1] = 3, 1
    mt1, mt2 = th.zeros(628, dtype=th.int32, device="cuda"), th.zeros(628, dtype=th.int32, device="cuda")
    ops.mt19937_seed(mt1, 5), ops.mt19937_seed(mt2, 5)
    plain = [th.empty(B, D, device="cuda"), th.empty(B, A, device="cuda"), th.empty(B, D, device="cuda"), th.empty(B, 1, device="cuda"),
             th.empty(B, 1, device="cuda")]
    W = D + A
    xd, xn, xp = (th.full((B, W), 9.0, device="cuda") for _ in range(3))
    dn, rw = th.empty(B, 1, device="cuda"), th.empty(B, 1, device="cuda")
    ri, ei = th.empty(B, dtype=th.int64, device="cuda"), th.empty(B, dtype=th.int64, device="cuda")
    for rnd in range(3):
        ops.replay_sample(ring, mt1, B, *plain)
        ops.replay_sample_packed(ring, mt2, B, xd, xn, xp if rnd != 1 else None, dn, rw, ri, ei)
        assert th.equal(xd[:, :D], plain[0]) and th.equal(xd[:, D:], plain[1]) and th.equal(xn[:, :D], plain[2])
        assert th.equal(dn, plain[3]) and th.equal(rw, plain[4]) and th.equal(mt1, mt2)
        assert th.equal(ring.observations[ri, ei], plain[0])
        assert float(xn[:, D:].min()) == 9.0 and float(xp[:, D:].min()) == 9.0  # the action columns are not the sampler's
        if rnd != 1:
            assert th.equal(xp[:, :D], plain[0])


def test_twin_layout_reset_draw(ops):
    """Train B continues the env's PCG64 stream after train A: two consecutive generate_initial_state draws."""
    n = 64
    st = orc.pcg64_states_from_seeds(np.arange(n))
    dst = dev(st.view(np.uint64).reshape(n, 4).view(np.int64))
    out = th.zeros(n, 8, device="cuda")
    ops.reset_draw(dst, None, out, act_dim=4)
    a = orc.reset_draw(st)
    b = orc.reset_draw(st)
    np.testing.assert_array_equal(out.cpu().numpy(), np.concatenate([a, b], 1))
    np.testing.assert_array_equal(dst.cpu().numpy().view(np.uint64).reshape(-1), st.view(np.uint64).reshape(-1))


def test_multi_update_launch_equals_separate_calls(ops):
    """cstr_adam_multi_f32: two Adam segments with different hyper-parameters and a polyak segment in ONE launch produce
    exactly what the separate cstr_adam_f32 / cstr_polyak_f32 calls produce (same device functions)."""
    g = th.Generator(device="cuda").manual_seed(4)

    def arena(n):
        return [th.randn(n, device="cuda", generator=g) for _ in range(2)] + [th.zeros(n, device="cuda"), th.zeros(n, device="cuda")]

    def fresh():
        th.manual_seed(0)
        a, b = arena(135744), arena(64)
        src, tgt = th.randn(70000, device="cuda", generator=th.Generator(device="cuda").manual_seed(9)), th.randn(70000, device="cuda", generator=th.Generator(device="cuda").manual_seed(10))
        ca, cb = ops.new_adam_ctl("cuda", 0, 0.9, 0.999), ops.new_adam_ctl("cuda", 0, 0.8, 0.99)
        return a, b, src, tgt, ca, cb

    g = th.Generator(device="cuda").manual_seed(4)
    a1, b1, s1, t1, ca1, cb1 = fresh()
    g = th.Generator(device="cuda").manual_seed(4)
    a2, b2, s2, t2, ca2, cb2 = fresh()
    lra, lrb = th.tensor([3e-4], dtype=th.float64, device="cuda"), th.tensor([1e-2], dtype=th.float64, device="cuda")
    for _ in range(3):
        ops.adam(a1[0], a1[1], a1[2], a1[3], ca1, lra, 0.9, 0.999, 1e-8, 1.0)
        ops.adam(b1[0], b1[1], b1[2], b1[3], cb1, lrb, 0.8, 0.99, 1e-6, 0.5)
        ops.polyak(s1, t1, 0.005)
        ops.adam_multi([(a2[0], a2[1], a2[2], a2[3], ca2, lra, 0.9, 0.999, 1e-8, 1.0),
                        (b2[0], b2[1], b2[2], b2[3], cb2, lrb, 0.8, 0.99, 1e-6, 0.5), ("polyak", s2, t2, 0.005)])
    for x, y in zip(a1 + b1 + [t1, ca1, cb1], a2 + b2 + [t2, ca2, cb2]):
        assert th.equal(x, y)
    assert int(ca2[0]) == 3 and int(cb2[0]) == 3


def test_adam_segment_with_its_own_soft_target_update(ops):
    """cstr_adam_seg_t.own_target: the Adam step and the soft update of the SAME parameters' target in one launch equal
    cstr_adam_f32 followed by cstr_polyak_f32 bit for bit (odd tail included), next to a plain polyak segment."""
    g = th.Generator(device="cuda").manual_seed(21)
    n = 135744 + 3
    mk = lambda: [th.randn(n, device="cuda", generator=th.Generator(device="cuda").manual_seed(k)) for k in (1, 2)] + \
        [th.zeros(n, device="cuda"), th.zeros(n, device="cuda")]  # noqa: E731
    a1, a2 = mk(), mk()
    t1 = th.randn(n, device="cuda", generator=g)
    t2 = t1.clone()
    s1 = th.randn(5000, device="cuda", generator=g)
    u1 = th.randn(5000, device="cuda", generator=g)
    s2, u2 = s1.clone(), u1.clone()
    c1, c2 = ops.new_adam_ctl("cuda"), ops.new_adam_ctl("cuda")
    lr = th.tensor([1e-3], dtype=th.float64, device="cuda")
    for _ in range(3):
        ops.adam(a1[0], a1[1], a1[2], a1[3], c1, lr, 0.9, 0.999, 1e-8, 1.0)
        ops.polyak(a1[0], t1, 0.005)
        ops.polyak(s1, u1, 0.005)
        ops.adam_multi([(a2[0], a2[1], a2[2], a2[3], c2, lr, 0.9, 0.999, 1e-8, 1.0, None, (t2, 0.005)), ("polyak", s2, u2, 0.005)])
    for x, y in zip(a1 + [t1, u1, c1], a2 + [t2, u2, c2]):
        assert th.equal(x, y)
    with pytest.raises(Exception):  # a target that IS the parameter buffer
        ops.adam_multi([(a2[0], a2[1], a2[2], a2[3], c2, lr, 0.9, 0.999, 1e-8, 1.0, None, (a2[0], 0.005))])


def test_adam_streaming_regime_equals_cached_regime(ops):
    """Arenas of >= 2^24 parameters take the streaming form of the Adam kernel (non-temporal loads and stores, two quads per
    stream in flight): the same arithmetic, so the result equals the cached form's bit for bit -- checked by running the same data
    as ONE 2^24 + 8-element arena and as two halves below the threshold (separate control words, same step)."""
    n = (1 << 24) + 8
    g = th.Generator(device="cuda").manual_seed(7)
    p0, gr = th.randn(n, device="cuda", generator=g), th.randn(n, device="cuda", generator=g) * 0.01
    lr = th.tensor([1e-3], dtype=th.float64, device="cuda")
    pa, ma, va, ca = p0.clone(), th.zeros(n, device="cuda"), th.zeros(n, device="cuda"), ops.new_adam_ctl("cuda")
    h = n // 2  # a multiple of 4: both halves stay 16-byte aligned
    halves = []
    for lo, hi in ((0, h), (h, n)):
        halves.append((p0[lo:hi].clone(), th.zeros(hi - lo, device="cuda"), th.zeros(hi - lo, device="cuda"), ops.new_adam_ctl("cuda"), gr[lo:hi].clone()))
    for _ in range(3):
        ops.adam(pa, gr, ma, va, ca, lr)
        for p, m, v, c, gg in halves:
            ops.adam(p, gg, m, v, c, lr)
    for name, whole, idx in (("param", pa, 0), ("exp_avg", ma, 1), ("exp_avg_sq", va, 2)):
        assert th.equal(whole, th.cat([halves[0][idx], halves[1][idx]])), name
    assert int(ca[0]) == 3 and int(ca[1]) == 0
