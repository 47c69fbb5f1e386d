"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's VecNormalize arithmetic, used by tests/ as the checker
for the device kernels (cstr_vecnorm_*). Never imported by the product.

Follows core/common/running_mean_std.py:34-55 (parallel-variance merge of batch moments into f64 running moments) and
core/common/vec_env/vec_normalize.py:174-246 (update order inside step_wait, clip((x - mean) / sqrt(var + eps)), discounted
return tracking). Pinned against the reference itself by tests/golden/vecnormalize_kat.npz (test_oracle_vecnorm.py).
"""
import numpy as np


class Moments:
    """running_mean_std.py:5-15: mean 0, var 1, count 1e-4 at start; all f64"""

    def __init__(self, shape=()):
        self.mean, self.var, self.count = np.zeros(shape, np.float64), np.ones(shape, np.float64), 1e-4

    def merge_batch(self, x: np.ndarray) -> None:
        """running_mean_std.py:34-55. NumPy reduces a float32 batch in float32 (:35-36)."""
        b_mean, b_var, n = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
        tot = self.count + n
        d = b_mean - self.mean
        m2 = self.var * self.count + b_var * n + np.square(d) * self.count * n / tot
        self.mean, self.var, self.count = self.mean + d * n / tot, m2 / tot, tot


class VecNormOracle:
    def __init__(self, n_envs, obs_dim, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0,
                 gamma=0.99, epsilon=1e-8):
        self.obs_m, self.ret_m = Moments((obs_dim,)), Moments(())
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.returns = np.zeros(n_envs)

    def normalize_obs(self, obs):
        """vec_normalize.py:206-214, :225-241"""
        if not self.norm_obs:
            return obs
        z = (obs - self.obs_m.mean) / np.sqrt(self.obs_m.var + self.epsilon)
        return np.clip(z, -self.clip_obs, self.clip_obs).astype(np.float32)

    def normalize_reward(self, rew):
        """vec_normalize.py:243-252"""
        if self.norm_reward:
            rew = np.clip(rew / np.sqrt(self.ret_m.var + self.epsilon), -self.clip_reward, self.clip_reward)
        return rew.astype(np.float32)

    def reset(self, obs):
        """vec_normalize.py:291-307"""
        self.returns = np.zeros_like(self.returns)
        if self.training and self.norm_obs:
            self.obs_m.merge_batch(obs)
        return self.normalize_obs(obs)

    def step(self, obs, rew, done):
        """vec_normalize.py:174-204: obs stats, normalise obs, returns/ret stats, normalise reward, zero finished returns"""
        if self.training and self.norm_obs:
            self.obs_m.merge_batch(obs)
        n_obs = self.normalize_obs(obs)
        if self.training:
            self.returns = self.returns * self.gamma + rew
            self.ret_m.merge_batch(self.returns)
        n_rew = self.normalize_reward(rew)
        self.returns[done.astype(bool)] = 0
        return n_obs, n_rew

    def stats(self):
        return np.concatenate([self.obs_m.mean, self.obs_m.var, [self.obs_m.count if self.norm_obs else 0.0],
                               [self.ret_m.mean, self.ret_m.var, self.ret_m.count]])
