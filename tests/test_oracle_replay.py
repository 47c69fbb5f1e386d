"""Pin the oracle's MT19937 randint + replay ring to numpy / the reference's ReplayBuffer.

Bit-exact (int64 indices, f32 payload copies). Golden vectors: mt19937_randint_kat.npz
(np.random.seed / np.random.randint exactly as buffers.py:113,309 calls them) and
replay_kat.npz (reference ReplayBuffer.add/sample incl. wrap-around and dones*(1-timeouts)).
"""
import numpy as np
import pytest

from oracle import cstr_oracle as orc


def test_mt19937_golden(golden):
    g = golden("mt19937_randint_kat.npz")
    for ci in range(int(g["n_cases"])):
        mt = orc.MT19937(int(g[f"c{ci}_seed"]))
        outs = [mt.randint(int(u), int(b)) for u, b in g[f"c{ci}_calls"]]
        np.testing.assert_array_equal(np.concatenate(outs), g[f"c{ci}_out"])
        np.testing.assert_array_equal(mt.key, g[f"c{ci}_key"])
        assert mt.pos == int(g[f"c{ci}_pos"])


@pytest.mark.parametrize("seed", [0, 1, 4095, 2**31, 2**32 - 1])
def test_mt19937_live_vs_numpy(seed):
    """Same contract checked live against this container's numpy (legacy global RandomState)."""
    rs = np.random.RandomState(seed)
    mt = orc.MT19937(seed)
    rng = np.random.default_rng(seed)
    for _ in range(40):
        high = int(rng.choice([1, 2, 3, 5, 244, 245, 256, 257, 976, 4096, 100000, 2**20 + 1, 2**32, 2**33 + 5]))
        n = int(rng.integers(1, 700))
        np.testing.assert_array_equal(mt.randint(high, n), rs.randint(0, high, size=n))
    st = rs.get_state()
    np.testing.assert_array_equal(mt.key, st[1])
    assert mt.pos == st[2]


@pytest.mark.parametrize("seed,act_dim,n_envs", [(0, 2, 64), (7, 2, 1), (42, 3, 5), (123, 1, 33), (2**32 - 1, 4, 16)])
def test_mt19937_legacy_normal_interleaved_with_randint_vs_numpy(seed, act_dim, n_envs):
    """VectorizedActionNoise(NormalActionNoise) (noise.py:44-45, :141-142) draws n_envs x np.random.normal(mu, sigma) from
    the SAME global stream as ReplayBuffer.sample's randint (buffers.py:113, :309): values (f32, bit-exact on this
    libm), the cached second deviate (odd counts) and the stream position must all follow numpy."""
    rs, mt = np.random.RandomState(seed), orc.MT19937(seed)
    mu, sigma = np.linspace(-0.5, 0.5, act_dim), np.linspace(0.1, 0.3, act_dim)
    for it in range(12):
        want = np.stack([rs.normal(mu, sigma).astype(np.float32) for _ in range(n_envs)])
        np.testing.assert_array_equal(mt.normal(mu, sigma, n_envs), want)
        np.testing.assert_array_equal(mt.randint(244, 256), rs.randint(0, 244, size=256))
        np.testing.assert_array_equal(mt.randint(n_envs, 256), rs.randint(0, n_envs, size=256))
    st = rs.get_state()
    np.testing.assert_array_equal(mt.key, st[1])
    assert (mt.pos, int(mt.st.has_gauss), float(mt.st.gauss)) == (st[2], st[3], st[4])


def test_survey_index_example():
    """seed 0, upper 244, B 256 -> 266 words; then n_envs 4096 -> 256 words."""
    mt = orc.MT19937(0)
    r = mt.randint(244, 256)
    assert list(r[:8]) == [172, 47, 117, 192, 67, 195, 103, 9] and mt.last_used == 266
    e = mt.randint(4096, 256)
    assert list(e[:8]) == [2827, 166, 2159, 3421, 4089, 1153, 2783, 1910] and mt.last_used == 256


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_replay_ring_golden(golden, tag):
    g = golden("replay_kat.npz")
    R, N, D, A, n_add, B = (int(x) for x in g[f"{tag}_dims"])
    ring = orc.ReplayRing(R, N, D, A)
    mt = orc.MT19937(int(g["seed"]))
    for k in range(n_add):
        ring.add(g[f"{tag}_obs"][k], g[f"{tag}_next_obs"][k], g[f"{tag}_act"][k], g[f"{tag}_rew"][k],
                 g[f"{tag}_done"][k], g[f"{tag}_timeout"][k])
        (o, a, no, d, r), _ = ring.sample(mt, B)
        for name, got in (("observations", o), ("actions", a), ("next_observations", no), ("dones", d), ("rewards", r)):
            np.testing.assert_array_equal(got, g[f"{tag}_s_{name}"][k], err_msg=f"{tag} add#{k} {name}")
    for name, arr in (("obs", ring.observations), ("next_obs", ring.next_observations), ("act", ring.actions),
                      ("rew", ring.rewards), ("done", ring.dones), ("timeout", ring.timeouts)):
        np.testing.assert_array_equal(arr, g[f"{tag}_ring_{name}"])
    assert ring.pos == int(g[f"{tag}_pos"]) and ring.full == bool(g[f"{tag}_full"])


def test_sample_empty_ring_raises():
    ring = orc.ReplayRing(4, 2, 4, 2)
    with pytest.raises(ValueError):  # numpy: "high <= 0" (buffers.py:113 with pos == 0)
        ring.sample(orc.MT19937(0), 8)


def test_config1_single_env_stream_matches_the_reference_run(golden):
    """BASELINE config 1: the UNMODIFIED reference SAC("MlpPolicy", DummyVecEnv([CSTR]), seed=0).learn(10_000) on the CPU left
    its global legacy stream in `mt_key / mt_pos` (tools/refharness/gen_golden.py:gen_config1). The oracle replays the sampler's
    calls: seed + n_envs - 1 = 0; per gradient step randint(0, rows_written) for 256 rows, then randint(0, 1), which consumes
    nothing (core/common/buffers.py:113-114, :309)."""
    g = golden("config1_sac_single_env_kat.npz")
    total, ls, B = int(g["total_timesteps"]), int(g["learning_starts"]), int(g["batch_size"])
    assert int(g["n_updates"]) == total - ls and int(g["ring_pos"]) == total and not bool(g["ring_full"]) and int(g["episode_num"]) == total // 400
    mt, rs = orc.MT19937(int(g["seed"]) + 1 - 1), np.random.RandomState(int(g["seed"]))
    for k in range(ls + 1, total + 1):  # trains once num_timesteps > learning_starts; `upper` = pos = k rows written
        a = mt.randint(k, B)
        e = mt.randint(1, B)
        assert not e.any() and a.max() < k
        if k % 1237 == 0:
            np.testing.assert_array_equal(a, rs.randint(0, k, size=B))
        else:
            rs.randint(0, k, size=B)
    np.testing.assert_array_equal(mt.key, g["mt_key"])
    assert mt.pos == int(g["mt_pos"]) and int(g["mt_has_gauss"]) == 0
    st = rs.get_state()
    np.testing.assert_array_equal(st[1], g["mt_key"])
    assert st[2] == int(g["mt_pos"])
