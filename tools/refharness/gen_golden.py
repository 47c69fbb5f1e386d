#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference in the dev container.

Usage (dev container only; /root/reference must exist):
    python tools/refharness/gen_golden.py [--only env,sampler,replay,sac,td3,init,vecenv]

Every array written is data (inputs + the reference's outputs); no reference source is copied.
Injected quantities (never produced by the stand-in gymnasium): initial states, actions, batches.
Recorded-by-hook quantities: the Normal eps draws and the (current_q, target_q) arguments of
mse_loss; the hooks wrap torch-level functions, the reference code itself is untouched.

Reference entry points exercised (file:line in /root/reference):
  twoseriescstr.py:394-503              TwoSeriesCSTREnv.step / _dynamics / compute_reward
  core/common/vec_env/dummy_vec_env.py:56-73   auto-reset, terminal_observation, TimeLimit.truncated
  core/common/buffers.py:106-115,247-325 ReplayBuffer.add / sample / _get_samples
  core/sac/sac.py:199-296                SAC.train
  core/td3/td3.py:154-211                TD3.train
  core/common/utils.py:457-481           polyak_update
"""
import argparse
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refload  # noqa: E402

refload.load()
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
th.set_num_threads(1)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {len(arrays)} arrays, {os.path.getsize(path)} bytes")


def slim_weights(out, keep=4096, head=64, with_shape=True):
    """Digest form for class-default nets: tensors above `keep` elements are stored as sum / abs-sum / first `head` values
    (the initial weights are reproducible from the seed -- checked bit-exactly by the init KAT and by #head/#sum here)."""
    slim = {}
    for k, v in out.items():
        if (k.startswith("before/") or k.startswith("after/")) and v.size > keep:
            slim[k + "#sum"] = np.float64(v.astype(np.float64).sum())
            slim[k + "#abs"] = np.float64(np.abs(v.astype(np.float64)).sum())
            slim[k + "#head"] = v.reshape(-1)[:head].copy()
            if with_shape:
                slim[k + "#shape"] = np.array(v.shape, np.int64)
        else:
            slim[k] = v
    return slim


# --------------------------------------------------------------------------------------- env
def gen_env():
    from twoseriescstr import TwoSeriesCSTREnv

    rng = np.random.default_rng(20250418)
    env = TwoSeriesCSTREnv()
    env.reset(seed=0)

    # ---- single-step known-answer tests -------------------------------------------------
    edge_obs = [
        [0, 0, 0, 0], [1, 1, 1, 1], [-1, -1, -1, -1], [1, -1, 1, -1], [-1, 1, -1, 1],
        [0.5, 0.9, 0.5, 0.9], [-0.42857143, -0.2, -0.42857143, -0.5], [1.5, 1.5, -1.5, -1.5],
        [0.3, 0.99, 0.2, 0.97], [-0.9, -0.95, -0.99, -0.9],
    ]
    edge_act = [
        [0, 0], [-1, -1], [1, 1], [2, -3], [-1, 1], [0.25, -0.75], [1, -1], [-5, 5], [0.999, -0.999], [0.1, 0.1],
    ]
    K = 4096
    obs = rng.uniform(-1.0, 1.0, size=(K, 4)).astype(np.float32)
    # a slice slightly outside the box exercises the raw-state safety clip (twoseriescstr.py:406-410)
    obs[:256] = rng.uniform(-1.2, 1.2, size=(256, 4)).astype(np.float32)
    # hot reactors (runaway region: big exp term, upper clip)
    obs[256:768, 1] = rng.uniform(0.5, 1.0, size=512).astype(np.float32)
    obs[512:768, 3] = rng.uniform(0.5, 1.0, size=256).astype(np.float32)
    act = rng.uniform(-1.0, 1.0, size=(K, 2)).astype(np.float32)
    act[:256] = rng.uniform(-1.5, 1.5, size=(256, 2)).astype(np.float32)
    obs = np.concatenate([np.array(edge_obs, np.float32), obs])
    act = np.concatenate([np.array(edge_act, np.float32), act])
    step_in = rng.integers(0, 399, size=len(obs)).astype(np.int32)
    step_in[:16] = [398, 399, 0, 397, 398, 399, 1, 2, 398, 399, 398, 399, 100, 200, 398, 399]

    n = len(obs)
    o2 = np.zeros((n, 4), np.float32)
    rew = np.zeros(n, np.float32)
    trunc = np.zeros(n, np.uint8)
    term = np.zeros(n, np.uint8)
    raw_next = np.zeros((n, 4), np.float32)
    raw_act = np.zeros((n, 2), np.float32)
    conc_r = np.zeros(n, np.float32)
    temp_p = np.zeros(n, np.float32)
    for i in range(n):
        env.reset()  # clears the stateful reward memory
        env.state = obs[i].copy()
        env.current_step = int(step_in[i])
        s, r, te, tr, info = env.step(act[i].copy())
        assert s.dtype == np.float32 and isinstance(r, np.floating) and r.dtype == np.float32
        o2[i], rew[i], term[i], trunc[i] = s, r, te, tr
        raw_next[i], raw_act[i] = info["original_state"], info["raw_action"]
        conc_r[i], temp_p[i] = info["concentration_reward"], info["temp_penalty"]
    save("env_step_kat.npz", obs=obs, act=act, step_in=step_in, obs_next=o2, reward=rew, terminated=term,
         truncated=trunc, raw_next=raw_next, raw_action=raw_act, concentration_reward=conc_r, temp_penalty=temp_p)

    # ---- NaN action: dynamics raises -> (old state, -10, False, True) twoseriescstr.py:413-421
    env.reset()
    env.state = np.array([0.1, 0.2, -0.3, 0.4], np.float32)
    env.current_step = 7
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        s, r, te, tr, info = env.step(np.array([np.nan, 0.0], np.float32))
    save("env_nan_kat.npz", obs=np.array([0.1, 0.2, -0.3, 0.4], np.float32), act=np.array([np.nan, 0.0], np.float32),
         obs_next=np.asarray(s, np.float32), reward=np.float32(r), terminated=np.uint8(te), truncated=np.uint8(tr),
         step_after=np.int32(env.current_step))

    # ---- 400-step trajectories under fixed action tapes -----------------------------------
    M, T = 6, 400
    obs0 = np.array([
        [0.28571430, -0.41931412, -0.28571430, -0.73472524],  # raw ~[0.45,310,0.25,290]
        [0.0, 0.0, 0.0, 0.0],
        [-0.5, 0.2, -0.6, 0.1],
        [0.9, 0.6, 0.8, 0.5],
        [-0.8, -0.8, -0.9, -0.9],
        [0.2, 0.75, 0.1, 0.7],
    ], np.float32)
    t = np.arange(T, dtype=np.float32)
    actions = np.zeros((T, M, 2), np.float32)
    actions[:, 0] = [0.0, 0.0]
    actions[:, 1] = np.stack([np.sin(0.05 * t), np.cos(0.03 * t)], 1)
    actions[:, 2] = rng.uniform(-1, 1, size=(T, 2))
    actions[:, 3] = [1.0, 1.0]
    actions[:, 4] = [-1.0, -1.0]
    actions[:, 5] = rng.uniform(-1, 1, size=(T, 2)) * 0.3
    actions = actions.astype(np.float32)
    obs_t = np.zeros((T, M, 4), np.float32)
    rew_t = np.zeros((T, M), np.float32)
    trunc_t = np.zeros((T, M), np.uint8)
    for m in range(M):
        env.reset()
        env.state = obs0[m].copy()
        env.current_step = 0
        for k in range(T):
            s, r, te, tr, info = env.step(actions[k, m].copy())
            obs_t[k, m], rew_t[k, m], trunc_t[k, m] = s, r, tr
    save("env_traj_kat.npz", obs0=obs0, actions=actions, obs=obs_t, reward=rew_t, truncated=trunc_t)


# ----------------------------------------------------------------------------------- vec env
def gen_vecenv():
    from core.common.vec_env.dummy_vec_env import DummyVecEnv
    from twoseriescstr import TwoSeriesCSTREnv

    rng = np.random.default_rng(7)
    N, T = 5, 6
    venv = DummyVecEnv([lambda: TwoSeriesCSTREnv() for _ in range(N)])
    venv.seed(11)
    venv.reset()
    obs0 = rng.uniform(-0.8, 0.8, size=(N, 4)).astype(np.float32)
    step0 = np.array([397, 398, 399, 10, 396], np.int32)
    for i, e in enumerate(venv.envs):
        e.state = obs0[i].copy()
        e.current_step = int(step0[i])
    actions = rng.uniform(-1, 1, size=(T, N, 2)).astype(np.float32)
    obs = np.zeros((T, N, 4), np.float32)       # what VecEnv.step returns (post-reset obs for done envs)
    term_obs = np.zeros((T, N, 4), np.float32)  # next_obs as stored in the buffer (terminal obs on done)
    rew = np.zeros((T, N), np.float32)
    done = np.zeros((T, N), np.uint8)
    timeout = np.zeros((T, N), np.uint8)
    for k in range(T):
        o, r, d, infos = venv.step(actions[k])
        obs[k], rew[k], done[k] = o, r, d
        for i in range(N):
            timeout[k, i] = infos[i].get("TimeLimit.truncated", False)
            term_obs[k, i] = infos[i]["terminal_observation"] if d[i] else o[i]
    # reset_obs[k, i] is the INJECTED reset source: the post-reset observation the reference drew
    # (gymnasium np_random -> stand-in -> unpinned), so it is an input of this fixture, not an output.
    save("vecenv_autoreset_kat.npz", obs0=obs0, step0=step0, actions=actions, obs=obs, next_obs_for_buffer=term_obs,
         reward=rew, done=done, timeout=timeout, reset_obs=obs.copy())


# ----------------------------------------------------------------------------------- sampler
def gen_sampler():
    """np.random.randint exactly as called at core/common/buffers.py:113-114 and :309."""
    cases = [
        # (seed, [(upper, B), (n_envs, B)] * calls)
        (0, [(244, 256), (4096, 256)]),
        (3, [(1, 256), (4, 256), (2, 256), (4, 256), (3, 256), (4, 256)]),
        (4095, [(1, 256), (4096, 256), (2, 256), (4096, 256), (244, 256), (4096, 256)]),
        (42, [(100000, 256), (1, 256), (99999, 100), (1, 100)]),
        (123, [(244, 256), (4096, 256)] * 8),
        (7, [(976, 256), (1024, 256)] * 4),
        (2**32 - 1, [(5, 1), (7, 1), (1000, 3), (6, 2048), (65536, 700), (65537, 700)]),
        (99, [(2**31 - 1, 64), (2**32, 64), (3, 1500)]),
    ]
    out = {}
    for ci, (seed, calls) in enumerate(cases):
        np.random.seed(seed)
        res = []
        for (upper, b) in calls:
            r = np.random.randint(0, upper, size=b)
            assert r.dtype == np.int64
            res.append(r)
        st = np.random.get_state()
        out[f"c{ci}_seed"] = np.uint64(seed)
        out[f"c{ci}_calls"] = np.array(calls, np.int64)
        out[f"c{ci}_out"] = np.concatenate(res)
        out[f"c{ci}_key"] = st[1].astype(np.uint32)
        out[f"c{ci}_pos"] = np.int32(st[2])
    out["n_cases"] = np.int32(len(cases))
    save("mt19937_randint_kat.npz", **out)


# ------------------------------------------------------------------------------------ replay
def gen_replay():
    from core.common.buffers import ReplayBuffer
    from gymnasium import spaces

    rng = np.random.default_rng(5)
    out = {}
    for tag, (R, N, D, A, n_add, B) in {"small": (5, 3, 4, 2, 8, 16), "wide": (7, 8, 8, 2, 5, 64)}.items():
        ospace = spaces.Box(-1, 1, (D,), np.float32)
        aspace = spaces.Box(-1, 1, (A,), np.float32)
        buf = ReplayBuffer(R * N, ospace, aspace, device="cpu", n_envs=N)
        assert buf.buffer_size == R
        obs = rng.uniform(-1, 1, (n_add, N, D)).astype(np.float32)
        nobs = rng.uniform(-1, 1, (n_add, N, D)).astype(np.float32)
        act = rng.uniform(-1, 1, (n_add, N, A)).astype(np.float32)
        rew = rng.uniform(-8, 0, (n_add, N)).astype(np.float32)
        done = (rng.uniform(size=(n_add, N)) < 0.4)
        tout = done & (rng.uniform(size=(n_add, N)) < 0.6)
        samples = []
        np.random.seed(1234)
        for k in range(n_add):
            infos = [{"TimeLimit.truncated": bool(tout[k, i])} for i in range(N)]
            buf.add(obs[k], nobs[k], act[k], rew[k], done[k], infos)
            s = buf.sample(B)
            samples.append([x.numpy() for x in s])
        out.update({
            f"{tag}_dims": np.array([R, N, D, A, n_add, B], np.int64), f"{tag}_obs": obs, f"{tag}_next_obs": nobs,
            f"{tag}_act": act, f"{tag}_rew": rew, f"{tag}_done": done.astype(np.uint8), f"{tag}_timeout": tout.astype(np.uint8),
            f"{tag}_ring_obs": buf.observations.copy(), f"{tag}_ring_next_obs": buf.next_observations.copy(),
            f"{tag}_ring_act": buf.actions.copy(), f"{tag}_ring_rew": buf.rewards.copy(),
            f"{tag}_ring_done": buf.dones.copy(), f"{tag}_ring_timeout": buf.timeouts.copy(),
            f"{tag}_pos": np.int64(buf.pos), f"{tag}_full": np.uint8(buf.full),
        })
        for fi, fname in enumerate(["observations", "actions", "next_observations", "dones", "rewards"]):
            out[f"{tag}_s_{fname}"] = np.stack([s[fi] for s in samples])
    out["seed"] = np.int64(1234)
    save("replay_kat.npz", **out)


# ------------------------------------------------------------------------------ learner KATs
class _Recorder:
    def __init__(self):
        self.eps, self.mse = [], []


def _flat_sd(prefix, sd):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in sd.items()}


def _fill_buffer(model, rng, n_rows, N, D, A):
    for _ in range(n_rows):
        obs = rng.uniform(-1, 1, (N, D)).astype(np.float32)
        nobs = np.clip(obs + rng.normal(0, 0.05, (N, D)), -1, 1).astype(np.float32)
        act = rng.uniform(-1, 1, (N, A)).astype(np.float32)
        rew = rng.uniform(-8, 0, (N,)).astype(np.float32)
        done = rng.uniform(size=N) < 0.2
        tout = done & (rng.uniform(size=N) < 0.5)
        model.replay_buffer.add(obs, nobs, act, rew, done, [{"TimeLimit.truncated": bool(t)} for t in tout])


def _make_venv(N):
    from core.common.vec_env.dummy_vec_env import DummyVecEnv
    from twoseriescstr import TwoSeriesCSTREnv

    return DummyVecEnv([lambda: TwoSeriesCSTREnv() for _ in range(N)])


def gen_sac():
    import torch.distributions.normal as tdn
    import torch.nn.functional as F_real

    import core.sac.sac as sacmod
    from core.common.logger import Logger
    from core.sac.sac import SAC

    for tag, net_arch, B, n_steps in (("small", [64, 64], 64, 3), ("default", None, 256, 2)):
        rec = _Recorder()
        orig_sn = tdn._standard_normal

        def rec_sn(shape, dtype, device):
            e = orig_sn(shape, dtype, device)
            rec.eps.append(e.clone())
            return e

        class FProxy:
            def __getattr__(self, name):
                return getattr(F_real, name)

            @staticmethod
            def mse_loss(a, b, *args, **kw):
                rec.mse.append((a.detach().clone(), b.detach().clone()))
                return F_real.mse_loss(a, b, *args, **kw)

        N, D, A = 4, 4, 2
        venv = _make_venv(N)
        pk = {} if net_arch is None else {"policy_kwargs": dict(net_arch=net_arch)}
        model = SAC("MlpPolicy", venv, seed=0, device="cpu", batch_size=B, buffer_size=64 * N, **pk)
        model.set_logger(Logger(folder=None, output_formats=[]))
        rng = np.random.default_rng(99)
        _fill_buffer(model, rng, 40, N, D, A)
        out = {}
        out.update(_flat_sd("before/actor", model.actor.state_dict()))
        out.update(_flat_sd("before/critic", model.critic.state_dict()))
        out.update(_flat_sd("before/critic_target", model.critic_target.state_dict()))
        out["before/log_ent_coef"] = model.log_ent_coef.detach().numpy().copy()
        rb = model.replay_buffer
        out.update(ring_obs=rb.observations.copy(), ring_next_obs=rb.next_observations.copy(), ring_act=rb.actions.copy(),
                   ring_rew=rb.rewards.copy(), ring_done=rb.dones.copy(), ring_timeout=rb.timeouts.copy(),
                   ring_pos=np.int64(rb.pos), ring_full=np.uint8(rb.full))
        np.random.seed(2024)
        th.manual_seed(77)
        orig_sample = rb.sample
        batches = []

        def rec_sample(batch_size, env=None):
            s = orig_sample(batch_size, env=env)
            batches.append([x.numpy().copy() for x in s])
            return s

        rb.sample = rec_sample
        tdn._standard_normal = rec_sn
        sacmod.F = FProxy()
        try:
            for k in range(n_steps):
                model.train(gradient_steps=1, batch_size=B)
                lv = model.logger.name_to_value
                out[f"step{k}/critic_loss"] = np.float32(lv["train/critic_loss"])
                out[f"step{k}/actor_loss"] = np.float32(lv["train/actor_loss"])
                out[f"step{k}/ent_coef_loss"] = np.float32(lv["train/ent_coef_loss"])
                out[f"step{k}/ent_coef"] = np.float32(lv["train/ent_coef"])
        finally:
            tdn._standard_normal = orig_sn
            sacmod.F = F_real
            rb.sample = orig_sample
        assert len(rec.eps) == 2 * n_steps and len(rec.mse) == 2 * n_steps
        for k in range(n_steps):
            for fi, fname in enumerate(["observations", "actions", "next_observations", "dones", "rewards"]):
                out[f"step{k}/batch_{fname}"] = batches[k][fi]
            out[f"step{k}/eps_pi"] = rec.eps[2 * k].numpy()
            out[f"step{k}/eps_next"] = rec.eps[2 * k + 1].numpy()
            out[f"step{k}/current_q1"] = rec.mse[2 * k][0].numpy()
            out[f"step{k}/current_q2"] = rec.mse[2 * k + 1][0].numpy()
            out[f"step{k}/target_q"] = rec.mse[2 * k][1].numpy()
        out.update(_flat_sd("after/actor", model.actor.state_dict()))
        out.update(_flat_sd("after/critic", model.critic.state_dict()))
        out.update(_flat_sd("after/critic_target", model.critic_target.state_dict()))
        out["after/log_ent_coef"] = model.log_ent_coef.detach().numpy().copy()
        out["hyper"] = np.array([model.gamma, model.tau, model.target_entropy, model.lr_schedule(1), B, n_steps], np.float64)
        out["np_seed"], out["th_seed"] = np.int64(2024), np.int64(77)
        if tag == "default":
            # keep the committed fixture small: weights are reproducible from seed 0 (checked by the
            # init KAT), so store only digests of the big tensors for the default-size nets
            out = slim_weights(out, with_shape=False)
        save(f"sac_train_kat_{tag}.npz", **out)


def gen_td3():
    _gen_td3("small")
    _gen_td3("default")


def gen_ddpg():
    _gen_td3("small", algo="ddpg")
    _gen_td3("default", algo="ddpg")


def _gen_td3(tag, algo="td3"):
    """tag "small": net_arch [48, 32], batch 64; "default": the class default [400, 300] (td3/policies.py:141-145), batch 256.
    algo "ddpg": core/ddpg/ddpg.py:14-130 -- TD3.train with policy_delay 1, ONE critic (one mse_loss per step) and a
    target-smoothing draw clamped to [-0, 0] (target_policy_noise 0.1, target_noise_clip 0.0)."""
    import torch.nn.functional as F_real

    import core.td3.td3 as td3mod
    from core.common.logger import Logger
    from core.td3.td3 import TD3

    if algo == "ddpg":
        from core.ddpg.ddpg import DDPG as TD3
    n_q = 1 if algo == "ddpg" else 2

    rec = _Recorder()

    class FProxy:
        def __getattr__(self, name):
            return getattr(F_real, name)

        @staticmethod
        def mse_loss(a, b, *args, **kw):
            rec.mse.append((a.detach().clone(), b.detach().clone()))
            return F_real.mse_loss(a, b, *args, **kw)

    N, D, A, n_steps = 4, 4, 2, 4
    B = 64 if tag == "small" else 256
    venv = _make_venv(N)
    pk = dict(policy_kwargs=dict(net_arch=[48, 32])) if tag == "small" else {}
    model = TD3("MlpPolicy", venv, seed=0, device="cpu", batch_size=B, buffer_size=64 * N, **pk)
    model.set_logger(Logger(folder=None, output_formats=[]))
    rng = np.random.default_rng(314)
    _fill_buffer(model, rng, 40, N, D, A)
    out = {}
    for nm in ("actor", "actor_target", "critic", "critic_target"):
        out.update(_flat_sd(f"before/{nm}", getattr(model, nm).state_dict()))
    rb = model.replay_buffer
    out.update(ring_obs=rb.observations.copy(), ring_next_obs=rb.next_observations.copy(), ring_act=rb.actions.copy(),
               ring_rew=rb.rewards.copy(), ring_done=rb.dones.copy(), ring_timeout=rb.timeouts.copy(),
               ring_pos=np.int64(rb.pos), ring_full=np.uint8(rb.full))
    np.random.seed(555)
    orig_sample = rb.sample
    batches = []

    def rec_sample(batch_size, env=None):
        s = orig_sample(batch_size, env=env)
        batches.append([x.numpy().copy() for x in s])
        return s

    rb.sample = rec_sample
    td3mod.F = FProxy()
    try:
        for k in range(n_steps):
            th.manual_seed(1000 + k)
            # the target-smoothing noise td3.py:169 is the first torch-RNG consumer of the step
            g = th.Generator().manual_seed(1000 + k)
            out[f"step{k}/noise_raw"] = th.empty(B, A).normal_(0, model.target_policy_noise, generator=g).numpy()
            model.train(gradient_steps=1, batch_size=B)
            lv = model.logger.name_to_value
            out[f"step{k}/critic_loss"] = np.float32(lv["train/critic_loss"])
            if model._n_updates % model.policy_delay == 0:
                out[f"step{k}/actor_loss"] = np.float32(lv["train/actor_loss"])
    finally:
        td3mod.F = F_real
        rb.sample = orig_sample
    assert len(rec.mse) == n_q * n_steps and len(model.critic.q_networks) == n_q
    for k in range(n_steps):
        for fi, fname in enumerate(["observations", "actions", "next_observations", "dones", "rewards"]):
            out[f"step{k}/batch_{fname}"] = batches[k][fi]
        out[f"step{k}/current_q1"] = rec.mse[n_q * k][0].numpy()
        if n_q == 2:
            out[f"step{k}/current_q2"] = rec.mse[2 * k + 1][0].numpy()
        out[f"step{k}/target_q"] = rec.mse[n_q * k][1].numpy()
    for nm in ("actor", "actor_target", "critic", "critic_target"):
        out.update(_flat_sd(f"after/{nm}", getattr(model, nm).state_dict()))
    out["hyper"] = np.array([model.gamma, model.tau, model.target_policy_noise, model.target_noise_clip,
                             model.policy_delay, model.lr_schedule(1), B, n_steps], np.float64)
    out["np_seed"] = np.int64(555)
    if tag == "default":
        assert model.actor.mu[0].out_features == 400 and model.actor.mu[2].out_features == 300
        save(f"{algo}_train_kat_default.npz", **slim_weights(out))
    else:
        save(f"{algo}_train_kat.npz", **out)


def gen_iddpg():
    gen_maddpg(algo="iddpg")


def gen_maddpg_default():
    gen_maddpg(tag="default")


class _BoxOnlyEnv:
    """Harness-side env for learner fixtures whose shape no reference env has (BASELINE config 5: 4 agents, 8 obs / 4 act):
    MADDPG.train() (maddpg.py:117-191) uses the env's SPACES only -- the batch comes from the injected ring rows."""

    def __new__(cls, D, A):
        import gymnasium
        from gymnasium import spaces

        class BoxOnly(gymnasium.Env):
            observation_space = spaces.Box(-1, 1, (D,), np.float32)
            action_space = spaces.Box(-1, 1, (A,), np.float32)

            def reset(self, *, seed=None, options=None):
                return np.zeros(D, np.float32), {}

            def step(self, action):
                return np.zeros(D, np.float32), 0.0, False, False, {}

        return BoxOnly()


def gen_maddpg4_default():
    gen_maddpg(tag="default", n_agents=4)


def gen_maddpg4():
    gen_maddpg(tag="small", n_agents=4)


def gen_maddpg(algo="maddpg", tag="small", n_agents=2):
    """MADDPG / IDDPG on the natural 2-agent split of the CSTR env: agent 0 = reactor 1 ([C1,T1] -> F1), agent 1 = reactor 2.
    tag "default": the class-default per-agent nets [400, 300] (maddpg/policies.py:344-353), batch 256.
    n_agents=4: BASELINE config 5's learner shape -- 8 obs / 4 act, one agent per (C, T) pair and coolant flow
    (obs splits [[0,1],[2,3],[4,5],[6,7]], act splits [[0],[1],[2],[3]]); the env only supplies the spaces."""
    import torch.nn.functional as F_real

    from core.common.logger import Logger
    if algo == "maddpg":
        import core.maddpg.maddpg as mmod
        from core.maddpg.maddpg import MADDPG
    else:
        import core.iddpg.iddpg as mmod
        from core.iddpg.iddpg import IDDPG as MADDPG

    rec = _Recorder()

    class FProxy:
        def __getattr__(self, name):
            return getattr(F_real, name)

        @staticmethod
        def mse_loss(a, b, *args, **kw):
            rec.mse.append((a.detach().clone(), b.detach().clone()))
            return F_real.mse_loss(a, b, *args, **kw)

    N, D, A, n_steps = 4, 2 * n_agents, n_agents, 4
    B = 64 if tag == "small" else 256
    if n_agents == 2:
        venv = _make_venv(N)
    else:
        from core.common.vec_env.dummy_vec_env import DummyVecEnv

        venv = DummyVecEnv([lambda: _BoxOnlyEnv(D, A) for _ in range(N)])
    pk = dict(policy_kwargs=dict(net_arch=[[32, 24]] * n_agents)) if tag == "small" else {}
    model = MADDPG(n_agents, "MlpPolicy", venv, [[2 * a, 2 * a + 1] for a in range(n_agents)], [[a] for a in range(n_agents)],
                   learning_rate_list=[1e-3] * n_agents, seed=0, device="cpu", batch_size=B, buffer_size=64 * N, **pk)
    model.set_logger(Logger(folder=None, output_formats=[]))
    rng = np.random.default_rng(2718)
    _fill_buffer(model, rng, 40, N, D, A)
    out = {}
    for nm in ("actor", "actor_target", "critic", "critic_target"):
        out.update(_flat_sd(f"before/{nm}", getattr(model, nm).state_dict()))
    rb = model.replay_buffer
    out.update(ring_obs=rb.observations.copy(), ring_next_obs=rb.next_observations.copy(), ring_act=rb.actions.copy(),
               ring_rew=rb.rewards.copy(), ring_done=rb.dones.copy(), ring_timeout=rb.timeouts.copy(),
               ring_pos=np.int64(rb.pos), ring_full=np.uint8(rb.full))
    np.random.seed(777)
    orig_sample = rb.sample
    batches = []

    def rec_sample(batch_size, env=None):
        s = orig_sample(batch_size, env=env)
        batches.append([x.numpy().copy() for x in s])
        return s

    rb.sample = rec_sample
    mmod.F = FProxy()
    try:
        for k in range(n_steps):
            th.manual_seed(2000 + k)
            g = th.Generator().manual_seed(2000 + k)
            for a in range(n_agents):  # maddpg.py:137: one normal_ draw per agent, in agent order
                out[f"step{k}/noise_raw_agent{a}"] = th.empty(B, 1).normal_(0, model.target_policy_noise, generator=g).numpy()
            model.train(gradient_steps=1, batch_size=B)
            lv = model.logger.name_to_value
            for a in range(n_agents):
                out[f"step{k}/agent{a}_critic_loss"] = np.float32(lv[f"train/agent_{a}_critic_loss"])
                if model._n_updates % model.policy_delay == 0:
                    out[f"step{k}/agent{a}_actor_loss"] = np.float32(lv[f"train/agent_{a}_actor_loss"])
    finally:
        mmod.F = F_real
        rb.sample = orig_sample
    assert len(rec.mse) == 2 * n_agents * n_steps, len(rec.mse)
    for k in range(n_steps):
        for fi, fname in enumerate(["observations", "actions", "next_observations", "dones", "rewards"]):
            out[f"step{k}/batch_{fname}"] = batches[k][fi]
        for a in range(n_agents):
            base = 2 * n_agents * k + 2 * a
            out[f"step{k}/agent{a}_current_q1"] = rec.mse[base][0].numpy()
            out[f"step{k}/agent{a}_current_q2"] = rec.mse[base + 1][0].numpy()
            out[f"step{k}/agent{a}_target_q"] = rec.mse[base][1].numpy()
    for nm in ("actor", "actor_target", "critic", "critic_target"):
        out.update(_flat_sd(f"after/{nm}", getattr(model, nm).state_dict()))
    out["hyper"] = np.array([model.gamma, model.tau, model.target_policy_noise, model.target_noise_clip, model.policy_delay,
                             1e-3, B, n_steps, n_agents], np.float64)
    out["np_seed"] = np.int64(777)
    # the reference's _sample_action for multi-agent algos: no scaling, no noise (multiagent_policy_algorithm.py:369, 391-392)
    obs = rng.uniform(-1, 1, (N, D)).astype(np.float32)
    model._last_obs = obs
    model.num_timesteps = 10**6
    from core.common.noise import NormalActionNoise

    act, buf_act = model._sample_action(0, NormalActionNoise(np.zeros(1), np.ones(1)), N)
    pred, _ = model.predict(obs, deterministic=False)
    out.update(sa_obs=obs, sa_action=act, sa_buffer_action=buf_act, sa_predict=pred)
    name = algo if n_agents == 2 else f"{algo}{n_agents}"
    if tag == "default":
        shapes = sorted({tuple(v.shape) for k, v in out.items() if k.startswith("before/actor/") and v.ndim == 2})
        assert (400, 2) in shapes and (300, 400) in shapes, shapes
        save(f"{name}_train_kat_default.npz", **slim_weights(out))
    else:
        save(f"{name}_train_kat.npz", **out)


def gen_config1():
    """BASELINE config 1 (plumbing): the unmodified reference SAC("MlpPolicy") at the class defaults on ONE CSTR env, learn(10_000)
    on the CPU. Recorded: the counters and the global legacy numpy stream afterwards. With n_envs = 1 the env-index draw
    `randint(0, high=1)` consumes nothing (buffers.py:309), so the stream position is a function of the row-index draws only."""
    import time

    from core.common.logger import Logger
    from core.sac.sac import SAC

    seed, total = 0, 10_000
    model = SAC("MlpPolicy", _make_venv(1), seed=seed, device="cpu")
    model.set_logger(Logger(folder=None, output_formats=[]))
    t0 = time.time()
    model.learn(total)
    dt = time.time() - t0
    st = np.random.get_state()
    rb = model.replay_buffer
    save("config1_sac_single_env_kat.npz", seed=np.int64(seed), total_timesteps=np.int64(total), num_timesteps=np.int64(model.num_timesteps),
         n_updates=np.int64(model._n_updates), episode_num=np.int64(model._episode_num), ring_pos=np.int64(rb.pos), ring_full=np.uint8(rb.full),
         ring_rows=np.int64(rb.buffer_size), batch_size=np.int64(model.batch_size), learning_starts=np.int64(model.learning_starts),
         mt_key=st[1].astype(np.uint32), mt_pos=np.int32(st[2]), mt_has_gauss=np.int32(st[3]),
         reference_env_steps_per_s=np.float64(total / dt), actor_adam_step=np.float64(float(model.actor.optimizer.state_dict()["state"][0]["step"])),
         dones_sum=np.float64(rb.dones[:rb.pos].sum()), timeouts_sum=np.float64(rb.timeouts[:rb.pos].sum()))
    print(f"reference SAC, 1 env, {total} steps: {total / dt:.1f} env-steps/s on this container (1 torch thread)")


def gen_checkpoint():
    """A checkpoint WRITTEN BY THE REFERENCE (SAC.save, base_class.py:842-888) after two gradient steps, plus the
    deterministic predictions of the saved model: the product's SAC.load must read it (weights_only tensors + JSON)."""
    from core.common.logger import Logger
    from core.sac.sac import SAC

    N, D, A, B = 4, 4, 2, 64
    model = SAC("MlpPolicy", _make_venv(N), seed=0, device="cpu", batch_size=B, buffer_size=64 * N, learning_starts=77, gamma=0.98,
                policy_kwargs=dict(net_arch=[64, 64]))
    model.set_logger(Logger(folder=None, output_formats=[]))
    rng = np.random.default_rng(4242)
    _fill_buffer(model, rng, 40, N, D, A)
    np.random.seed(1)
    th.manual_seed(1)
    model.train(gradient_steps=2, batch_size=B)
    model.num_timesteps = 1234
    path = os.path.join(OUT, "sac_reference_checkpoint.zip")
    model.save(path)
    obs = rng.uniform(-1, 1, (16, D)).astype(np.float32)
    pred, _ = model.predict(obs, deterministic=True)
    with th.no_grad():
        q1, q2 = model.critic(th.as_tensor(obs), th.as_tensor(pred))
    opt = model.critic.optimizer.state_dict()
    save("sac_reference_checkpoint_kat.npz", obs=obs, pred=pred, q1=q1.numpy(), q2=q2.numpy(), log_ent_coef=model.log_ent_coef.detach().numpy(),
         critic_adam_step=np.float32(float(opt["state"][0]["step"])), critic_exp_avg0=opt["state"][0]["exp_avg"].numpy(),
         n_updates=np.int64(model._n_updates))
    print("wrote", path, os.path.getsize(path))


def gen_eval():
    """The reference's evaluate_policy (core/common/evaluation.py:11-140) over DummyVecEnv([TwoSeriesCSTREnv] * N):
    (a) a replaying predictor (any object with predict(), evaluation.py:88-93) over a fixed action tape with out-of-range
        actions and two NaN actions (-> the env's exception path ends that episode early, twoseriescstr.py:413-421),
        n_eval_episodes = 7 over 3 envs (targets [2, 2, 3], evaluation.py:79-82): per-episode returns (f64 sums of f32 rewards),
        lengths, their order, and the (mean, std) form;
    (b) the seeded, untrained SAC / TD3 policies at the class defaults, deterministic=True, 6 episodes over 4 envs."""
    import contextlib
    import io

    from core.common.evaluation import evaluate_policy
    from core.common.vec_env.dummy_vec_env import DummyVecEnv
    from core.sac.sac import SAC
    from core.td3.td3 import TD3
    from twoseriescstr import TwoSeriesCSTREnv

    N, n_eval, T = 3, 7, 1300
    rng = np.random.default_rng(808)
    tape = rng.uniform(-1.2, 1.2, size=(T, N, 2)).astype(np.float32)
    tape[137, 0, 1] = np.nan    # env 0's first episode ends at length 138
    tape[655, 2, 0] = np.nan    # env 2's second episode ends at length 256

    class Tape:
        def __init__(self):
            self.t = 0

        def predict(self, observations, state=None, episode_start=None, deterministic=False):
            a = tape[self.t].copy()
            self.t += 1
            return a, state

    def run(model, n_envs, seed, n_episodes, **kw):
        venv = DummyVecEnv([lambda: TwoSeriesCSTREnv() for _ in range(n_envs)])
        venv.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            return evaluate_policy(model, venv, n_eval_episodes=n_episodes, warn=False, **kw)

    out = dict(tape=tape, tape_dims=np.array([N, n_eval, 11], np.int64))
    tm = Tape()
    rets, lens = run(tm, N, 11, n_eval, return_episode_rewards=True)
    out.update(tape_returns=np.asarray(rets, np.float64), tape_lengths=np.asarray(lens, np.int64), tape_steps_used=np.int64(tm.t))
    mean, std = run(Tape(), N, 11, n_eval)
    out.update(tape_mean=np.float64(mean), tape_std=np.float64(std))
    for name, cls in (("sac", SAC), ("td3", TD3)):
        model = cls("MlpPolicy", DummyVecEnv([lambda: TwoSeriesCSTREnv() for _ in range(2)]), seed=0, device="cpu")
        rets, lens = run(model, 4, 5, 6, deterministic=True, return_episode_rewards=True)
        out[f"{name}_returns"], out[f"{name}_lengths"] = np.asarray(rets, np.float64), np.asarray(lens, np.int64)
    out["model_dims"] = np.array([4, 6, 5, 0], np.int64)  # n_envs, n_eval_episodes, env seed, model seed
    save("evaluate_policy_kat.npz", **out)


def gen_info():
    """The 16-key `info` dict of TwoSeriesCSTREnv.step (twoseriescstr.py:441-452) with the nine compute_reward diagnostics (:379-389),
    five of which carry weight 0.0 and per-env memory: a seeded 80-step trajectory that approaches the target (stability counter runs),
    leaves it, and is reset in between (memory cleared)."""
    from twoseriescstr import TwoSeriesCSTREnv

    env = TwoSeriesCSTREnv()
    rng = np.random.default_rng(77)
    keys = ["concentration_reward", "concentration_proximity_reward", "concentration_trend_reward", "stability_reward", "temp_penalty",
            "action_smoothness_penalty", "extreme_penalty", "concentration_error", "stable_steps"]
    T = 80
    acts = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    acts[0:36] = np.array([0.34522182, 0.3167114], np.float32) + rng.normal(0, 0.01, (36, 2)).astype(np.float32)  # a calm stretch near the target
    acts[60] = np.array([1.7, -2.0], np.float32)
    starts = {0: np.array([-0.59437853, 0.23396769, -0.42535093, 0.3727205], np.float32), 45: np.array([0.9, 0.9, 0.95, 0.8], np.float32)}
    out = {k: np.zeros(T, np.float64) for k in keys}
    obs_next, rew, raw_next, raw_act, step_no = np.zeros((T, 4), np.float32), np.zeros(T, np.float32), np.zeros((T, 4), np.float32), np.zeros((T, 2), np.float32), np.zeros(T, np.int64)
    for t in range(T):
        if t in starts:
            env.reset(seed=1)
            env.state = starts[t].copy()
        s, r, te, tr, info = env.step(acts[t].copy())
        assert set(info) == set(keys) | {"reward", "raw_action", "truncated", "state", "original_state", "target_C2", "step"}
        for k in keys:
            out[k][t] = info[k]
        obs_next[t], rew[t], raw_next[t], raw_act[t], step_no[t] = s, r, info["original_state"], info["raw_action"], info["step"]
    save("env_info_kat.npz", actions=acts, reset_at=np.array(sorted(starts), np.int64), reset_state=np.stack([starts[k] for k in sorted(starts)]),
         obs_next=obs_next, reward=rew, original_state=raw_next, raw_action=raw_act, step=step_no, **out)


def gen_init():
    """Initial weights of the reference policies for seed 0 (construction order = RNG order)."""
    from core.sac.sac import SAC
    from core.td3.td3 import TD3

    out = {}
    for name, cls in (("sac", SAC), ("td3", TD3)):
        for seed in (0, 5):
            model = cls("MlpPolicy", _make_venv(2), seed=seed, device="cpu")
            mods = ["actor", "critic", "critic_target"] + (["actor_target"] if name == "td3" else [])
            for nm in mods:
                for k, v in getattr(model, nm).state_dict().items():
                    a = v.numpy()
                    out[f"{name}{seed}/{nm}/{k}#shape"] = np.array(a.shape, np.int64)
                    out[f"{name}{seed}/{nm}/{k}#sum"] = np.float64(a.astype(np.float64).sum())
                    out[f"{name}{seed}/{nm}/{k}#head"] = a.reshape(-1)[:16].copy()
            # global legacy numpy stream after construction (seed clobbering is in reset, see SURVEY a-6)
            st = np.random.get_state()
            out[f"{name}{seed}/np_key_head"] = st[1][:8].astype(np.uint32)
            out[f"{name}{seed}/np_pos"] = np.int32(st[2])
            model._setup_learn(100, None)
            st = np.random.get_state()
            out[f"{name}{seed}/np_key_head_after_setup_learn"] = st[1][:8].astype(np.uint32)
            out[f"{name}{seed}/np_pos_after_setup_learn"] = np.int32(st[2])
    save("policy_init_kat.npz", **out)


def gen_resets():
    """TwoSeriesCSTREnv.reset (twoseriescstr.py:226-269) in both init modes: seeded first reset, then unseeded resets that
    continue the env's generator; "static" also records the drifting f64 init_state. The generator behind
    `self.np_random` is the harness stand-in's Generator(PCG64(SeedSequence(seed))) = gymnasium's documented construction."""
    from twoseriescstr import TwoSeriesCSTREnv

    seeds, n_resets = [0, 1, 7, 42, 4095, 123456], 6
    out = {"seeds": np.array(seeds, np.int64)}
    for mode in ("random", "static"):
        obs = np.zeros((len(seeds), n_resets, 4), np.float32)
        init = np.zeros((len(seeds), n_resets, 4), np.float64)
        for i, sd in enumerate(seeds):
            env = TwoSeriesCSTREnv(init_mode=mode)
            for k in range(n_resets):
                o, info = env.reset(seed=sd) if k == 0 else env.reset()
                assert o.dtype == np.float32
                obs[i, k] = o
                if mode == "static":
                    init[i, k] = env.init_state
            # a few steps between resets must not touch the reset stream
        out[f"{mode}_obs"] = obs
        if mode == "static":
            out["static_init_state"] = init
    # static mode through the vectorised env: auto-reset draws continue each env's own stream
    from core.common.vec_env.dummy_vec_env import DummyVecEnv

    N, T = 3, 5
    venv = DummyVecEnv([lambda: TwoSeriesCSTREnv(init_mode="static") for _ in range(N)])
    venv.seed(21)
    o0 = venv.reset()
    step0 = np.array([398, 399, 397], np.int32)
    for i, e in enumerate(venv.envs):
        e.current_step = int(step0[i])
    rng = np.random.default_rng(3)
    actions = rng.uniform(-1, 1, size=(T, N, 2)).astype(np.float32)
    obs = np.zeros((T, N, 4), np.float32)
    done = np.zeros((T, N), np.uint8)
    for k in range(T):
        o, r, d, infos = venv.step(actions[k])
        obs[k], done[k] = o, d
    out.update(vec_seed=np.int64(21), vec_obs0=o0, vec_step0=step0, vec_actions=actions, vec_obs=obs, vec_done=done,
               vec_init_state=np.stack([e.init_state for e in venv.envs]))
    save("env_reset_kat.npz", **out)


def gen_vecnorm():
    """The reference's VecNormalize / RunningMeanStd (core/common/vec_env/vec_normalize.py:174-290,
    core/common/running_mean_std.py) driven by a replaying inner VecEnv: inputs are the raw obs / reward / done sequences,
    outputs the normalised obs / rewards and the running statistics after every step, plus normalize_obs /
    normalize_reward of a held-out batch as ReplayBuffer._get_samples applies them (buffers.py:143-155, :312-323)."""
    from core.common.vec_env.base_vec_env import VecEnv
    from core.common.vec_env.vec_normalize import VecNormalize
    from gymnasium import spaces

    N, D, T = 48, 4, 25
    rng = np.random.default_rng(5)
    scale, shift = np.array([0.3, 2.0, 0.05, 7.0]), np.array([0.1, -1.0, 0.0, 3.0])
    raw_obs = (rng.normal(size=(T + 1, N, D)) * scale + shift).astype(np.float32)
    raw_rew = (rng.normal(size=(T, N)) * 5.0 - 20.0).astype(np.float32)
    done = (rng.uniform(size=(T, N)) < 0.08)

    class Replay(VecEnv):
        def __init__(self):
            super().__init__(N, spaces.Box(-np.inf, np.inf, (D,), np.float32), spaces.Box(-1, 1, (2,), np.float32))
            self.k = 0

        def reset(self):
            self.k = 0
            return raw_obs[0].copy()

        def step_async(self, actions):
            pass

        def step_wait(self):
            k = self.k
            self.k += 1
            return raw_obs[k + 1].copy(), raw_rew[k].copy(), done[k].copy(), [{} for _ in range(N)]

        def close(self):
            pass

        def get_attr(self, attr_name, indices=None):
            return [None] * N

        def set_attr(self, attr_name, value, indices=None):
            pass

        def env_method(self, method_name, *a, indices=None, **kw):
            return [None] * N

        def env_is_wrapped(self, wrapper_class, indices=None):
            return [False] * N

    out = dict(raw_obs=raw_obs, raw_rew=raw_rew, done=done.astype(np.uint8))
    held_obs = (rng.normal(size=(64, D)) * scale * 3 + shift).astype(np.float32)
    held_rew = (rng.normal(size=(64, 1)) * 30.0 - 20.0).astype(np.float32)
    out.update(held_obs=held_obs, held_rew=held_rew)
    for tag, kw in (("default", {}), ("tight", dict(clip_obs=1.5, clip_reward=0.8, gamma=0.9, epsilon=1e-4)),
                    ("obs_only", dict(norm_reward=False)), ("rew_only", dict(norm_obs=False))):
        vn = VecNormalize(Replay(), **kw)
        o = vn.reset()
        n_obs, n_rew = [o], []
        stats = []
        for k in range(T):
            if tag == "default" and k == 18:
                vn.training = False  # frozen statistics for the tail
            o, r, d, _ = vn.step(np.zeros((N, 2), np.float32))
            n_obs.append(o)
            n_rew.append(r)
            om = vn.obs_rms if vn.norm_obs else None
            stats.append(np.concatenate([om.mean if om else np.zeros(D), om.var if om else np.ones(D), [om.count if om else 0.0],
                                         [vn.ret_rms.mean, vn.ret_rms.var, vn.ret_rms.count]]))
            assert np.array_equal(vn.get_original_obs(), raw_obs[k + 1]) and np.array_equal(vn.get_original_reward(), raw_rew[k])
        out[f"{tag}_norm_obs"] = np.stack(n_obs).astype(np.float32)
        out[f"{tag}_norm_rew"] = np.stack(n_rew).astype(np.float32)
        out[f"{tag}_stats"] = np.stack(stats)
        out[f"{tag}_returns"] = vn.returns.copy()
        out[f"{tag}_held_obs"] = vn.normalize_obs(held_obs)
        out[f"{tag}_held_rew"] = vn.normalize_reward(held_rew).astype(np.float32)
        out[f"{tag}_unnorm_obs"] = np.asarray(vn.unnormalize_obs(out[f"{tag}_held_obs"]), np.float64)
        out[f"{tag}_unnorm_rew"] = np.asarray(vn.unnormalize_reward(out[f"{tag}_held_rew"]), np.float64)
    save("vecnormalize_kat.npz", **out)


GENS = {"maddpg_default": gen_maddpg_default, "maddpg4_default": gen_maddpg4_default, "maddpg4": gen_maddpg4, "ddpg": gen_ddpg, "eval": gen_eval, "info": gen_info, "config1": gen_config1, "env": gen_env, "resets": gen_resets, "vecnorm": gen_vecnorm, "vecenv": gen_vecenv, "sampler": gen_sampler, "replay": gen_replay, "sac": gen_sac,
        "td3": gen_td3, "init": gen_init, "maddpg": gen_maddpg, "iddpg": gen_iddpg, "checkpoint": gen_checkpoint}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(GENS))
    args = ap.parse_args()
    for k in args.only.split(","):
        GENS[k]()
