"""Pin oracle/cstr_oracle.c (env step, reward, auto-reset) to the reference's golden vectors.

Golden vectors: tests/golden/env_*.npz, vecenv_autoreset_kat.npz, produced by running the
unmodified reference (twoseriescstr.py:394-503, dummy_vec_env.py:56-73) -- see
tools/refharness/gen_golden.py. Tolerance: fp32, 1e-6 relative per single step (libm expf vs
numpy's SIMD expf differ by <= ~2 ulp), 1e-5 over 400-step trajectories (north_star's bound).
"""
import numpy as np

from conftest import rel_err
from oracle import cstr_oracle as orc

# Observations are normalised to the unit box [-1, 1]: the error is measured relative to
# max(|x|, 1), i.e. to the box scale. (A 1-ulp difference in expf moves the raw state by 1 ulp,
# which after 2(x-lo)/span-1 is ~1e-7 ABSOLUTE even where the normalised value is ~0.)
# Rewards likewise: d(reward)/d(C2) is 5..30, so the same 1-ulp wobble is ~1e-6 absolute; the
# reward error is measured relative to max(|r|, 1).
OBS_FLOOR = 1.0


def test_single_step_kat(golden):
    g = golden("env_step_kat.npz")
    nxt, after, rew, done, tout, steps = orc.vec_step(g["obs"], g["act"], g["step_in"])
    assert rel_err(nxt, g["obs_next"], OBS_FLOOR) < 1e-6
    assert rel_err(rew, g["reward"], 1.0) < 2e-6
    np.testing.assert_array_equal(done.astype(np.uint8), g["truncated"] | g["terminated"])
    np.testing.assert_array_equal(tout.astype(np.uint8), g["truncated"] & (1 - g["terminated"]))
    assert g["terminated"].sum() == 0
    # bit-exactness is not claimed for floats, but most results should coincide
    same = (nxt.view(np.uint32) == g["obs_next"].view(np.uint32)).mean()
    assert same > 0.95, same
    # step counter: +1, reset to 0 on truncation (twoseriescstr.py:396, :264)
    exp_steps = np.where(g["truncated"] > 0, 0, g["step_in"] + 1)
    np.testing.assert_array_equal(steps, exp_steps)


def test_survey_kats(golden):
    """The three hand-checked vectors quoted in SURVEY.md 8c."""
    obs = np.array([[0, 0, 0, 0], [1, 1, 1, 1], [-1, -1, -1, -1]], np.float32)
    act = np.array([[0, 0], [-1, -1], [1, 1]], np.float32)
    nxt, _, rew, *_ = orc.vec_step(obs, act, np.zeros(3, np.int32))
    np.testing.assert_allclose(nxt[0], [0.020529151, 0.062121868, -0.0008994341, 0.07518828], rtol=2e-6)
    np.testing.assert_allclose(nxt[1], [0.7714423, 1, 0.80001366, 1], rtol=2e-6)
    np.testing.assert_allclose(nxt[2], [-0.9285714, -0.58131665, -1, -0.61825], rtol=2e-6)
    np.testing.assert_allclose(rew, [-1.4486028, -7.9997063, -2.25], rtol=2e-6)


def test_nan_action_path(golden):
    g = golden("env_nan_kat.npz")
    nxt, after, rew, done, tout, steps = orc.vec_step(g["obs"][None], g["act"][None], np.array([7], np.int32),
                                                      reset_obs=np.full((1, 4), 0.5, np.float32))
    np.testing.assert_array_equal(nxt[0], g["obs_next"])  # old state returned (twoseriescstr.py:418)
    assert rew[0] == g["reward"] == -10.0
    assert done[0] == 1.0 and tout[0] == 1.0 and g["truncated"] == 1 and g["terminated"] == 0
    assert g["step_after"] == 8  # current_step is incremented before the failure (:396)
    np.testing.assert_array_equal(after[0], np.full(4, 0.5, np.float32))  # VecEnv resets it
    assert steps[0] == 0


def test_trajectories(golden):
    g = golden("env_traj_kat.npz")
    obs = g["obs0"].copy()
    steps = np.zeros(len(obs), np.int32)
    worst_o = worst_r = 0.0
    T = g["actions"].shape[0]
    for k in range(T):
        nxt, after, rew, done, tout, steps = orc.vec_step(obs, g["actions"][k], steps, reset_obs=obs)
        worst_o = max(worst_o, rel_err(nxt, g["obs"][k], OBS_FLOOR))
        worst_r = max(worst_r, rel_err(rew, g["reward"][k], 1.0))
        np.testing.assert_array_equal(tout.astype(np.uint8), g["truncated"][k])
        obs = nxt  # follow the trajectory, not the reset
    assert g["truncated"][:-1].sum() == 0 and g["truncated"][-1].all()
    assert worst_o < 1e-5 and worst_r < 1e-5, (worst_o, worst_r)


def test_vecenv_autoreset(golden):
    g = golden("vecenv_autoreset_kat.npz")
    obs, steps = g["obs0"].copy(), g["step0"].copy()
    for k in range(g["actions"].shape[0]):
        nxt, after, rew, done, tout, steps = orc.vec_step(obs, g["actions"][k], steps, reset_obs=g["reset_obs"][k])
        assert rel_err(nxt, g["next_obs_for_buffer"][k], OBS_FLOOR) < 1e-6
        assert rel_err(after, g["obs"][k], OBS_FLOOR) < 1e-6
        assert rel_err(rew, g["reward"][k], 1.0) < 2e-6
        np.testing.assert_array_equal(done.astype(np.uint8), g["done"][k])
        np.testing.assert_array_equal(tout.astype(np.uint8), g["timeout"][k])
        # envs that finished return the injected reset obs bit-exactly
        d = g["done"][k].astype(bool)
        np.testing.assert_array_equal(after[d], g["reset_obs"][k][d])
        obs = after
    assert g["done"].sum() >= 4  # the fixture really exercises auto-reset


def test_reset_draws_both_init_modes_vs_reference(golden):
    """TwoSeriesCSTREnv.reset (twoseriescstr.py:226-269) run by the reference itself, seeded then continuing: the
    "random" draw (generate_initial_state) and the "static" mode's drifting f64 init_state, both bit-exact."""
    g = golden("env_reset_kat.npz")
    seeds = g["seeds"]
    st = orc.pcg64_states_from_seeds(seeds)
    for k in range(g["random_obs"].shape[1]):
        np.testing.assert_array_equal(orc.reset_draw(st), g["random_obs"][:, k])
    st = orc.pcg64_states_from_seeds(seeds)
    init = np.tile(np.array(orc.STATIC_INIT_STATE, np.float64), (len(seeds), 1))
    for k in range(g["static_obs"].shape[1]):
        np.testing.assert_array_equal(orc.reset_draw(st, static_init=init), g["static_obs"][:, k])
        np.testing.assert_array_equal(init, g["static_init_state"][:, k])
    assert np.abs(g["static_init_state"][:, -1] - np.array(orc.STATIC_INIT_STATE)).max() > 1.0  # it really drifts


def test_static_mode_autoreset_through_vecenv(golden):
    """DummyVecEnv auto-reset with init_mode="static" (dummy_vec_env.py:68-72): only finished envs draw."""
    g = golden("env_reset_kat.npz")
    n = g["vec_obs0"].shape[0]
    st = orc.pcg64_states_from_seeds(int(g["vec_seed"]) + np.arange(n))
    init = np.tile(np.array(orc.STATIC_INIT_STATE, np.float64), (n, 1))
    obs = orc.reset_draw(st, static_init=init)
    np.testing.assert_array_equal(obs, g["vec_obs0"])
    steps = g["vec_step0"].copy()
    for k in range(g["vec_actions"].shape[0]):
        nxt, _, _, done, _, steps = orc.vec_step(obs, g["vec_actions"][k], steps, reset_obs=obs)
        d = done.astype(np.uint8)
        np.testing.assert_array_equal(d, g["vec_done"][k])
        fresh = orc.reset_draw(st, d, static_init=init)
        obs = np.where(d[:, None].astype(bool), fresh, nxt)
        assert rel_err(obs, g["vec_obs"][k], OBS_FLOOR) < 1e-6
        np.testing.assert_array_equal(obs[d.astype(bool)], g["vec_obs"][k][d.astype(bool)])
    np.testing.assert_array_equal(init, g["vec_init_state"])
    assert g["vec_done"].sum() == 3


def test_rk4_consistency():
    """RK4 has no reference counterpart: check it is a 4th-order refinement of the
    same RHS -- one RK4 step must be far closer than one Euler step to 1024 Euler sub-steps."""
    rng = np.random.default_rng(0)
    obs = rng.uniform(-0.6, 0.3, (512, 4)).astype(np.float32)
    act = rng.uniform(-1, 1, (512, 2)).astype(np.float32)
    z = np.zeros(512, np.int32)
    e1 = orc.vec_step(obs, act, z, integrator="euler")[0]
    r1 = orc.vec_step(obs, act, z, integrator="rk4")[0]
    c = orc.default_coef()
    c.dt = 0.1 / 1024
    fine = obs
    for _ in range(1024):
        fine = orc.vec_step(fine, act, z, integrator="euler", coef=c)[0]
    err_e, err_r = np.abs(e1 - fine).max(), np.abs(r1 - fine).max()
    assert err_r < err_e / 20, (err_e, err_r)
