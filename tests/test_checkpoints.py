"""Checkpoint compatibility: archives in the reference's zip layout (core/common/save_util.py:294-336).
GPU tests: (a) an archive WRITTEN BY THE REFERENCE is loaded (weights_only tensors + plain JSON, nothing unpickled) and
reproduces the reference's predictions, Q-values and optimiser state; (b) save -> load round trip continues training
bit-identically; (c) the archive members and optimiser state-dict layout are the reference's."""
import os
import zipfile

import numpy as np
import pytest
import torch as th

from conftest import GOLDEN, rel_err

REF_ZIP = os.path.join(GOLDEN, "sac_reference_checkpoint.zip")


def test_safe_loader_reads_reference_archive_without_unpickling():
    """CPU: members, JSON data (cloudpickled entries skipped), tensors via weights_only."""
    from core.common.save_util import load_from_zip_file

    data, params, variables = load_from_zip_file(REF_ZIP)
    assert set(params) == {"policy", "actor.optimizer", "critic.optimizer", "ent_coef_optimizer"}
    assert data["gamma"] == 0.98 and data["learning_starts"] == 77 and data["num_timesteps"] == 1234 and data["_n_updates"] == 2
    assert "observation_space" not in data and "policy_class" not in data  # ":serialized:" entries are dropped, never unpickled
    assert variables["log_ent_coef"].shape == (1,)
    assert "actor.latent_pi.0.weight" in params["policy"] and "critic_target.qf1.4.bias" in params["policy"]


@pytest.mark.gpu
def test_load_reference_written_checkpoint(golden):
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    g = golden("sac_reference_checkpoint_kat.npz")
    model = SAC.load(REF_ZIP, env=CSTRVecEnv(4))
    assert (model.gamma, model.learning_starts, model.batch_size, model.buffer_size) == (0.98, 77, 64, 256)
    assert model.num_timesteps == 1234 and model._n_updates == int(g["n_updates"])
    pred, _ = model.predict(g["obs"], deterministic=True)
    assert rel_err(pred, g["pred"], 1.0) < 2e-6
    with th.no_grad():
        q1, q2 = model.critic(th.as_tensor(g["obs"]).cuda(), th.as_tensor(g["pred"]).cuda())
    assert rel_err(q1.cpu().numpy(), g["q1"], float(np.abs(g["q1"]).mean())) < 1e-5
    assert rel_err(q2.cpu().numpy(), g["q2"], float(np.abs(g["q2"]).mean())) < 1e-5
    assert float(model.log_ent_coef.detach()) == float(g["log_ent_coef"][0])
    opt = model.critic.optimizer
    assert opt.step_count == int(g["critic_adam_step"])
    sd = opt.state_dict()
    np.testing.assert_array_equal(sd["state"][0]["exp_avg"].cpu().numpy(), g["critic_exp_avg0"])


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["sac", "td3"])
def test_save_load_round_trip_continues_identically(tmp_path, algo):
    from core.common import legacy_rng
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC
    from core.td3 import TD3

    cls = SAC if algo == "sac" else TD3
    env = CSTRVecEnv(32)
    model = cls("MlpPolicy", env, seed=5, batch_size=32, buffer_size=32 * 16, gamma=0.97, policy_kwargs=dict(net_arch=[32, 32]))
    model.learn(32 * 12)
    path = str(tmp_path / "ckpt")
    model.save(path)
    with zipfile.ZipFile(path + ".zip") as z:
        names = set(z.namelist())
    want = {"data", "policy.pth", "actor.optimizer.pth", "critic.optimizer.pth", "_stable_baselines3_version", "system_info.txt"}
    assert want <= names and (("ent_coef_optimizer.pth" in names and "pytorch_variables.pth" in names) == (algo == "sac"))
    sd = model.actor.optimizer.state_dict()  # torch.optim.Adam layout
    assert set(sd) == {"state", "param_groups"} and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert sd["param_groups"][0]["params"] == list(range(len(list(model.actor.parameters()))))

    clone = cls.load(path, env=CSTRVecEnv(32))
    assert clone.gamma == 0.97 and clone.num_timesteps == model.num_timesteps and clone._n_updates == model._n_updates
    for a, b in zip(model.policy.state_dict().values(), clone.policy.state_dict().values()):
        assert th.equal(a, b)
    assert clone.critic.optimizer.step_count == model.critic.optimizer.step_count
    # continue both from the same ring / sampler state with the same noise: identical updates
    for name in ("observations", "next_observations", "actions", "rewards", "dones", "timeouts"):
        getattr(clone.replay_buffer, name).copy_(getattr(model.replay_buffer, name))
    clone.replay_buffer._adds = model.replay_buffer._adds
    clone.replay_buffer.ring.ctl.copy_(model.replay_buffer.ring.ctl)
    outs = []
    for m in (model, clone):
        m.set_random_seed(7)  # torch's generator and the fused kernels' own counter streams
        legacy_rng.seed(99, m.device)
        m.train(gradient_steps=2, batch_size=32)
        outs.append([p.detach().clone() for p in m.policy.parameters()])
    for a, b in zip(*outs):
        assert th.equal(a, b)


@pytest.mark.gpu
def test_replay_buffer_save_load_round_trip(tmp_path):
    """save_replay_buffer / load_replay_buffer (reference: off_policy_algorithm.py:214-254): the pickle carries the reference's
    attribute set as host NumPy arrays; after loading, the ring, its position and the sampled batches are identical and
    training continues from the loaded ring (also from a captured hipGraph)."""
    import pickle

    from core.common import legacy_rng
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    N, R = 32, 16
    model = SAC("MlpPolicy", CSTRVecEnv(N), seed=5, batch_size=32, buffer_size=N * R, policy_kwargs=dict(net_arch=[32, 32]))
    model.learn(N * 21)  # wraps the ring: pos 5, full
    path = tmp_path / "sub" / "rb"
    model.save_replay_buffer(path)
    with open(str(path) + ".pkl", "rb") as f:
        raw = pickle.load(f).__getstate__()
    assert raw["observations"].shape == (R, N, 4) and raw["actions"].shape == (R, N, 2) and raw["rewards"].shape == (R, N)
    assert isinstance(raw["observations"], np.ndarray) and (raw["pos"], raw["full"]) == (5, True)

    other = SAC("MlpPolicy", CSTRVecEnv(N), seed=5, batch_size=32, buffer_size=N * R, policy_kwargs=dict(net_arch=[32, 32]))
    other.enable_graph_capture()
    other.learn(N * 8)  # captures graphs against the ring that is about to be replaced
    other.load_replay_buffer(str(path) + ".pkl")
    a, b = model.replay_buffer, other.replay_buffer
    assert (b.pos, b.full, b.size(), b.buffer_size, b.n_envs) == (a.pos, a.full, a.size(), a.buffer_size, a.n_envs)
    assert th.equal(a.ring.ctl, b.ring.ctl)
    for name in ("observations", "next_observations", "actions", "rewards", "dones", "timeouts"):
        assert th.equal(getattr(a, name), getattr(b, name)), name
    legacy_rng.seed(11, a.device)
    sa = a.sample(64)
    legacy_rng.seed(11, a.device)
    sb = b.sample(64)
    for x, y in zip(sa, sb):
        assert th.equal(x, y)
    other.learn(N * 4, reset_num_timesteps=False)
    assert other.replay_buffer.pos == (5 + 4) % R and int(other.replay_buffer.ring.ctl[0]) == (5 + 4) % R
    for p in other.policy.parameters():
        assert th.isfinite(p).all()
