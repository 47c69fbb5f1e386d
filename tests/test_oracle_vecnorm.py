"""Pin oracle/vecnorm_np.py to the reference's own VecNormalize run (tests/golden/vecnormalize_kat.npz, produced by
tools/refharness/gen_golden.py --only vecnorm from core/common/vec_env/vec_normalize.py + running_mean_std.py)."""
import numpy as np
import pytest

from oracle.vecnorm_np import VecNormOracle

CASES = {"default": {}, "tight": dict(clip_obs=1.5, clip_reward=0.8, gamma=0.9, epsilon=1e-4),
         "obs_only": dict(norm_reward=False), "rew_only": dict(norm_obs=False)}


@pytest.mark.parametrize("tag", list(CASES))
def test_vecnormalize_restatement_matches_reference_bit_for_bit(golden, tag):
    g = golden("vecnormalize_kat.npz")
    raw_obs, raw_rew, done = g["raw_obs"], g["raw_rew"], g["done"]
    vn = VecNormOracle(raw_obs.shape[1], raw_obs.shape[2], **CASES[tag])
    np.testing.assert_array_equal(vn.reset(raw_obs[0]), g[f"{tag}_norm_obs"][0])
    for k in range(raw_rew.shape[0]):
        if tag == "default" and k == 18:
            vn.training = False
        o, r = vn.step(raw_obs[k + 1], raw_rew[k], done[k])
        np.testing.assert_array_equal(o, g[f"{tag}_norm_obs"][k + 1])
        np.testing.assert_array_equal(r, g[f"{tag}_norm_rew"][k])
        np.testing.assert_array_equal(vn.stats(), g[f"{tag}_stats"][k])
    np.testing.assert_array_equal(vn.returns, g[f"{tag}_returns"])
    np.testing.assert_array_equal(vn.normalize_obs(g["held_obs"]), g[f"{tag}_held_obs"])
    np.testing.assert_array_equal(vn.normalize_reward(g["held_rew"]), g[f"{tag}_held_rew"])
