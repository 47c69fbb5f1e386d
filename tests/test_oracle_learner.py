"""Pin oracle/sac_cpu.py (torch-fp32 CPU restatement of SAC.train) to the reference's golden vectors, and
check the host-side policy construction of the product reproduces the reference's seeded initial weights.
CPU only."""
import numpy as np
import torch as th

from conftest import rel_err
from oracle import cstr_oracle as orc
from oracle import sac_cpu


def test_sac_cpu_step_vs_reference_golden(golden):
    g = golden("sac_train_kat_small.npz")
    gamma, tau, target_entropy, lr, B, n_steps = g["hyper"]
    th.set_num_threads(1)
    m = sac_cpu.SacCpu(sac_cpu.params_from_golden(g), lr=lr, gamma=gamma, tau=tau, target_entropy=target_entropy,
                       log_ent_coef=float(g["before/log_ent_coef"][0]))
    # the batches themselves come from the C oracle's ring + MT19937 (bit-exact vs the reference's sample())
    ring = orc.ReplayRing(64, 4, 4, 2)
    for name, key in (("observations", "ring_obs"), ("next_observations", "ring_next_obs"), ("actions", "ring_act"),
                      ("rewards", "ring_rew"), ("dones", "ring_done"), ("timeouts", "ring_timeout")):
        getattr(ring, name)[...] = g[key]
    ring.c.pos, ring.c.full = int(g["ring_pos"]), int(g["ring_full"])
    mt = orc.MT19937(int(g["np_seed"]))
    for k in range(int(n_steps)):
        (o, a, no, d, r), _ = ring.sample(mt, int(B))
        for got, name in ((o, "observations"), (a, "actions"), (no, "next_observations"), (d, "dones"), (r, "rewards")):
            np.testing.assert_array_equal(got, g[f"step{k}/batch_{name}"])
        out = m.train_step(*(th.as_tensor(x) for x in (o, a, no, d, r)), th.as_tensor(g[f"step{k}/eps_pi"]),
                           th.as_tensor(g[f"step{k}/eps_next"]))
        assert rel_err(out["target_q"].numpy(), g[f"step{k}/target_q"], 1e-2) < 2e-6
        assert rel_err(out["current_q1"].numpy(), g[f"step{k}/current_q1"], 1e-2) < 2e-6
        assert rel_err(out["current_q2"].numpy(), g[f"step{k}/current_q2"], 1e-2) < 2e-6
        for key in ("critic_loss", "actor_loss", "ent_coef_loss", "ent_coef"):
            assert rel_err(out[key], float(g[f"step{k}/{key}"]), 1e-3) < 2e-6, key
    after = sac_cpu.params_from_golden(g, "after")
    for nm, cur in (("actor", m.actor), ("critic", m.critic), ("critic_target", m.critic_target)):
        for k, v in cur.items():
            np.testing.assert_allclose(v.detach().numpy(), after[nm][k].numpy(), rtol=1e-4, atol=2e-6, err_msg=f"{nm}/{k}")


def test_sac_cpu_init_matches_reference(golden):
    g = golden("sac_train_kat_small.npz")
    p = sac_cpu.init_params(4, 2, [64, 64], seed=0)
    ref = sac_cpu.params_from_golden(g)
    for nm in ("actor", "critic", "critic_target"):
        assert set(p[nm]) == set(ref[nm])
        for k in p[nm]:
            np.testing.assert_array_equal(p[nm][k].numpy(), ref[nm][k].numpy(), err_msg=f"{nm}/{k}")


def test_product_policy_construction_order_matches_reference(golden):
    """SACPolicy / TD3Policy built on the CPU generator (before any arena exists) == reference initial weights."""
    from core.common.spaces import Box
    from core.sac.policies import SACPolicy
    from core.td3.policies import TD3Policy

    g = golden("policy_init_kat.npz")
    ospace, aspace = Box(-1, 1, (4,)), Box(-1, 1, (2,))
    for name, cls, mods in (("sac", SACPolicy, ["actor", "critic", "critic_target"]),
                            ("td3", TD3Policy, ["actor", "critic", "critic_target", "actor_target"])):
        for seed in (0, 5):
            th.manual_seed(seed)
            pol = cls(ospace, aspace, lambda _: 3e-4)
            for nm in mods:
                for k, v in getattr(pol, nm).state_dict().items():
                    a = v.numpy()
                    assert tuple(a.shape) == tuple(g[f"{name}{seed}/{nm}/{k}#shape"]), (nm, k)
                    np.testing.assert_array_equal(a.reshape(-1)[:16], g[f"{name}{seed}/{nm}/{k}#head"], err_msg=f"{name}{seed}/{nm}/{k}")
                    assert a.astype(np.float64).sum() == float(g[f"{name}{seed}/{nm}/{k}#sum"])


def test_effective_sampler_seed_is_seed_plus_n_minus_1(golden):
    """SURVEY a-6: after _setup_learn the reference's global numpy stream == RandomState(seed + n_envs - 1)
    (fixture built with 2 envs: seeds 0 -> 1, 5 -> 6)."""
    g = golden("policy_init_kat.npz")
    for name in ("sac", "td3"):
        for seed in (0, 5):
            st = np.random.RandomState(seed + 2 - 1).get_state()
            np.testing.assert_array_equal(g[f"{name}{seed}/np_key_head_after_setup_learn"], st[1][:8])
            assert int(g[f"{name}{seed}/np_pos_after_setup_learn"]) == st[2]
            st0 = np.random.RandomState(seed).get_state()  # right after construction: set_random_seed(seed)
            np.testing.assert_array_equal(g[f"{name}{seed}/np_key_head"], st0[1][:8])
