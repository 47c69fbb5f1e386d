"""GPU parity tests of the device VecNormalize (cstr_vecnorm_*; reference: core/common/vec_env/vec_normalize.py,
core/common/running_mean_std.py, core/common/buffers.py:143-155) against the reference's own run
(tests/golden/vecnormalize_kat.npz) and the NumPy oracle.

Tolerances: the reference reduces a float32 batch in float32 (np.mean / np.var) before merging into f64 running moments;
the kernel accumulates in f64. Statistics therefore agree to ~1e-6 relative, normalised values to 2e-6 of the clip scale.
"""
import numpy as np
import pytest
import torch as th

from conftest import rel_err
from oracle.vecnorm_np import VecNormOracle

pytestmark = pytest.mark.gpu

CASES = {"default": {}, "tight": dict(clip_obs=1.5, clip_reward=0.8, gamma=0.9, epsilon=1e-4),
         "obs_only": dict(norm_reward=False), "rew_only": dict(norm_obs=False)}


def dev(a, dtype=None):
    t = th.as_tensor(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda().contiguous()


def _cfg(nv, d, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8):
    return nv.VecNormCfg(int(training), int(norm_obs), int(norm_reward), d, clip_obs, clip_reward, gamma, epsilon)


@pytest.mark.parametrize("tag", list(CASES))
def test_vecnorm_kernels_vs_reference_golden(golden, tag):
    from core import _native as nv
    from core.common import hip_ops as ops

    g = golden("vecnormalize_kat.npz")
    raw_obs, raw_rew, done = g["raw_obs"], g["raw_rew"], g["done"]
    T, N, D = raw_rew.shape[0], raw_obs.shape[1], raw_obs.shape[2]
    kw = dict(CASES[tag])
    st = th.zeros(nv.VECNORM_STATE_WORDS, dtype=th.float64, device="cuda")
    ops.vecnorm_init(st)
    ret = th.ones(N, dtype=th.float64, device="cuda")  # reset must zero it
    n_obs, n_rew = th.empty(N, D, device="cuda"), th.empty(N, device="cuda")
    ops.vecnorm_step(_cfg(nv, D, **kw), st, ret, dev(raw_obs[0]), None, None, n_obs, None)
    assert float(ret.abs().max()) == 0.0
    assert rel_err(n_obs.cpu().numpy(), g[f"{tag}_norm_obs"][0], 1.0) < 2e-6
    for k in range(T):
        if tag == "default" and k == 18:
            kw["training"] = False
        ops.vecnorm_step(_cfg(nv, D, **kw), st, ret, dev(raw_obs[k + 1]), dev(raw_rew[k]), dev(done[k], th.float32), n_obs, n_rew)
        assert rel_err(n_obs.cpu().numpy(), g[f"{tag}_norm_obs"][k + 1], 1.0) < 2e-6, k
        assert rel_err(n_rew.cpu().numpy(), g[f"{tag}_norm_rew"][k], 1.0) < 2e-6, k
        s = st.cpu().numpy()
        want = g[f"{tag}_stats"][k]
        got = np.concatenate([s[0:D], s[8:8 + D], [s[16] if kw.get("norm_obs", True) else 0.0], s[17:20]])
        assert rel_err(got, want, 1e-3) < 2e-6, (k, got, want)
    assert rel_err(ret.cpu().numpy(), g[f"{tag}_returns"], 1.0) < 1e-6
    # ReplayBuffer._get_samples(env=vec_normalize): held-out batch, in place
    o, o2, r = dev(g["held_obs"]), dev(g["held_obs"][::-1].copy()), dev(g["held_rew"])
    ops.vecnorm_apply(_cfg(nv, D, **kw), st, o, o2, r)
    assert rel_err(o.cpu().numpy(), g[f"{tag}_held_obs"], 1.0) < 2e-6
    assert rel_err(o2.cpu().numpy(), g[f"{tag}_held_obs"][::-1], 1.0) < 2e-6
    assert rel_err(r.cpu().numpy(), g[f"{tag}_held_rew"], 1.0) < 2e-6
    if tag == "tight":  # the clips really bind in this case
        assert float(o.abs().max()) == 1.5 and abs(float(r.abs().max()) - 0.8) < 1e-7


def test_vecnorm_rejects_bad_arguments():
    from core import _native as nv
    from core.common import hip_ops as ops

    st = th.zeros(nv.VECNORM_STATE_WORDS, dtype=th.float64, device="cuda")
    ret, o = th.zeros(4, dtype=th.float64, device="cuda"), th.zeros(4, 4, device="cuda")
    with pytest.raises(ValueError):
        ops.vecnorm_step(_cfg(nv, 4), st, ret, o, None, th.zeros(4, device="cuda"))  # reset form with done
    with pytest.raises(ValueError):
        ops.vecnorm_step(_cfg(nv, 4), st, ret, o, None, None, norm_obs_out=o)  # aliasing
    with pytest.raises(ValueError):
        ops.vecnorm_step(_cfg(nv, 8), st, ret, o, None, None)
    with pytest.raises(ValueError):
        ops.vecnorm_apply(_cfg(nv, 4), st, None, None, None)
    with pytest.raises(RuntimeError):
        ops.vecnorm_apply(_cfg(nv, 4, clip_obs=-1.0), st, o, None, None)


def test_vecnormalize_wrapper_api_follows_the_oracle(tmp_path):
    """The wrapper around a real CSTRVecEnv: reset / step / normalize_* / get_original_* / save / load against the NumPy
    restatement fed with the env's own raw outputs."""
    from core.common.vec_env import CSTRVecEnv, VecNormalize, unwrap_vec_normalize

    N = 32
    env = CSTRVecEnv(N)
    vn = VecNormalize(env, clip_obs=5.0, gamma=0.95)
    assert unwrap_vec_normalize(vn) is vn and unwrap_vec_normalize(env) is None and vn.unwrapped is env
    assert vn.num_envs == N and vn.obs_dim == 4 and vn.env_is_wrapped(VecNormalize) == [True] * N
    orc = VecNormOracle(N, 4, clip_obs=5.0, gamma=0.95)
    vn.seed(3)
    o = vn.reset()
    raw = vn.get_original_obs()
    assert rel_err(o, orc.reset(raw), 1.0) < 2e-6 and o.dtype == np.float32
    env.step_count.fill_(396)
    rng = np.random.default_rng(0)
    for k in range(8):
        o, r, d, infos = vn.step(rng.uniform(-1, 1, (N, 2)).astype(np.float32))
        raw, raw_r = vn.get_original_obs(), vn.get_original_reward()
        eo, er = orc.step(raw, raw_r, d)
        assert rel_err(o, eo, 1.0) < 2e-6 and rel_err(r, er, 1.0) < 2e-6
        if k == 3:
            assert d.all() and all("terminal_observation" in i and i["TimeLimit.truncated"] for i in infos)
        else:
            assert not d.any()
    assert rel_err(vn.returns, orc.returns, 1.0) < 1e-6
    assert rel_err(vn.obs_rms.mean, orc.obs_m.mean, 1e-3) < 2e-6 and rel_err(vn.obs_rms.var, orc.obs_m.var, 1e-3) < 2e-6
    assert abs(vn.obs_rms.count - orc.obs_m.count) < 1e-9 and abs(vn.ret_rms.count - orc.ret_m.count) < 1e-9
    x = rng.normal(size=(5, 4)).astype(np.float32)
    assert rel_err(vn.normalize_obs(x), orc.normalize_obs(x), 1.0) < 2e-6
    assert rel_err(vn.normalize_obs(th.as_tensor(x)).cpu().numpy(), orc.normalize_obs(x), 1.0) < 2e-6
    z = vn.normalize_obs(x)
    keep = np.abs(z) < 5.0  # round trip only where the clip did not bind
    assert keep.sum() >= 8 and rel_err(vn.unnormalize_obs(z)[keep], x[keep], 1.0) < 1e-5
    rr = rng.normal(size=7).astype(np.float32) * 30
    assert rel_err(vn.normalize_reward(rr), orc.normalize_reward(rr), 1.0) < 2e-6
    assert rel_err(vn.unnormalize_reward(vn.normalize_reward(rr[np.abs(rr) < 5])), rr[np.abs(rr) < 5], 1.0) < 1e-5
    # save / load keeps settings and statistics; a frozen copy no longer updates
    path = str(tmp_path / "vn.pkl")
    vn.save(path)
    vn2 = VecNormalize.load(path, CSTRVecEnv(N))
    assert (vn2.clip_obs, vn2.gamma, vn2.training) == (5.0, 0.95, True)
    np.testing.assert_array_equal(vn2.obs_rms.mean, vn.obs_rms.mean)
    np.testing.assert_array_equal(vn2.ret_rms.var, vn.ret_rms.var)
    vn2.training = False
    before = vn2.obs_rms.mean
    vn2.reset()
    vn2.step(np.zeros((N, 2), np.float32))
    np.testing.assert_array_equal(vn2.obs_rms.mean, before)
    with pytest.raises(ValueError):
        VecNormalize(env, norm_obs_keys=["a"])


@pytest.mark.parametrize("algo", ["sac", "td3"])
def test_learn_with_vecnormalize_eager_equals_graph_and_ring_stays_raw(algo):
    """off_policy_algorithm.py:469-496 + buffers.py:312-323: the ring stores ORIGINAL observations / rewards, the policy
    sees normalised observations, sampled batches are normalised with the current statistics; the captured-graph
    iteration reproduces the eager one."""
    from core.common.vec_env import CSTRVecEnv, VecNormalize
    from core.sac import SAC
    from core.td3 import TD3

    N, B, iters = 64, 32, 14
    res = []
    for graph in (False, True):
        env = VecNormalize(CSTRVecEnv(N))
        cls = SAC if algo == "sac" else TD3
        model = cls("MlpPolicy", env, seed=4, batch_size=B, buffer_size=N * 32, learning_starts=N * 2, policy_kwargs=dict(net_arch=[32, 32]))
        model.enable_graph_capture(graph)
        model.learn(N * iters)
        assert model.get_vec_normalize_env() is env and model.replay_buffer.normalizer is env
        assert bool(model._graph) == graph and model._n_updates == iters - 2
        th.cuda.synchronize()
        rb = model.replay_buffer
        res.append(dict(stats=env._state.cpu().numpy(), returns=env.returns, ring_obs=rb.observations.cpu().numpy(),
                        ring_rew=rb.rewards.cpu().numpy(), actor=model.policy.actor_arena.flat.cpu().numpy(),
                        last=model._last_obs.cpu().numpy(), raw=env.unwrapped.obs.cpu().numpy()))
        assert abs(env.obs_rms.count - (N * (iters + 1) + 1e-4)) < 1e-6 and abs(env.ret_rms.count - (N * iters + 1e-4)) < 1e-6
        # the ring holds raw observations: row k+1's obs is row k's next_obs (no reset happened yet)
        np.testing.assert_array_equal(rb.observations[1:iters].cpu().numpy(), rb.next_observations[:iters - 1].cpu().numpy())
        assert float(rb.rewards[:iters].min()) < -1.0  # raw CSTR rewards, not the normalised O(1) ones
        # _last_obs is the normalised view of the env's raw observation
        ref = VecNormOracle(N, 4)
        ref.obs_m.mean, ref.obs_m.var = env.obs_rms.mean, env.obs_rms.var
        assert rel_err(res[-1]["last"], ref.normalize_obs(res[-1]["raw"]), 1.0) < 2e-6
        # sampled batches are normalised with the current statistics
        batch, bi, ei = rb.sample_with_indices(16)
        raw_o = rb.observations[bi, ei].cpu().numpy()
        raw_r = rb.rewards[bi, ei].cpu().numpy()
        ref.ret_m.var = env.ret_rms.var
        assert rel_err(batch.observations.cpu().numpy(), ref.normalize_obs(raw_o), 1.0) < 2e-6
        assert rel_err(batch.rewards.cpu().numpy().ravel(), ref.normalize_reward(raw_r), 1.0) < 2e-6
    e, g = res
    for k in ("stats", "returns", "ring_obs", "ring_rew", "actor", "last"):
        np.testing.assert_allclose(e[k], g[k], rtol=2e-3, atol=2e-4, err_msg=k)


def test_evaluate_policy_through_vecnormalize():
    from core.common.evaluation import evaluate_policy
    from core.common.vec_env import CSTRVecEnv, VecNormalize
    from core.sac import SAC

    env = VecNormalize(CSTRVecEnv(8), training=False)
    model = SAC("MlpPolicy", env, seed=0, policy_kwargs=dict(net_arch=[16, 16]))
    rets, lens = evaluate_policy(model, env, n_eval_episodes=8, return_episode_rewards=True)
    assert len(rets) == 8 and all(l == 400 for l in lens) and all(r < -50 for r in rets)  # ORIGINAL rewards
