"""`evaluate_policy` (reference: core/common/evaluation.py:11-140).

Two evaluation loops with the reference's episode accounting (static split of the episodes over the sub-environments,
returns = float64 sums of the float32 step rewards, episodes appended in sub-environment order within a vec-step):

  * device loop -- `model` is one of this stack's algorithms / policies and `env` a `CSTRVecEnv` (optionally wrapped in
    `VecNormalize`), no per-step callback: actor, `predict()`'s action post-processing (policies.py:379-386), env step and the
    accounting all stay in HBM; the host reads one flag per vec-step and the finished episodes' rows;
  * host loop -- any object with `predict(observations, state=, episode_start=, deterministic=)` (evaluation.py:88-93) and any
    `VecEnv` of this package, NumPy in / NumPy out like the reference; also taken when a `callback` wants `locals()` per
    (step, env) (evaluation.py:104-105).
"""
import warnings
from typing import Callable, Optional, Tuple, Union

import numpy as np
import torch as th

from core.common.vec_env import CSTRVecEnv
from core.common.vec_env.base_vec_env import VecEnv


def _device_post(policy, dev):
    """`BasePolicy.predict`'s post-processing of the actor output (policies.py:379-386) as device ops with numpy's float32
    rounding points: squashed policies un-scale (`low + 0.5 * (a + 1) * (high - low)`), others clip into the action box.
    MADDPG's per-agent un-scaling (multi_agent_policies.py:605-610) is the same expression on the agent's columns."""
    low = th.as_tensor(np.asarray(policy.action_space.low, np.float32), device=dev)
    high = th.as_tensor(np.asarray(policy.action_space.high, np.float32), device=dev)
    if policy.squash_output:
        span = high - low
        return lambda a: low + (0.5 * (a + 1.0)) * span
    return lambda a: th.minimum(th.maximum(a, low), high)


def _evaluate_on_device(policy, env, targets_h: np.ndarray, deterministic: bool):
    n, dev = env.num_envs, env.unwrapped.device
    targets = th.as_tensor(targets_h, device=dev)
    counts = th.zeros(n, dtype=th.long, device=dev)
    cur_ret = th.zeros(n, dtype=th.float64, device=dev)   # current_rewards = np.zeros(n_envs): float64 (evaluation.py:84)
    cur_len = th.zeros(n, dtype=th.long, device=dev)
    rets, lens = [], []
    obs = env.reset_device()
    policy.set_training_mode(False)
    post = _device_post(policy, dev)
    while bool((counts < targets).any()):
        with th.no_grad():
            act = post(policy._predict(obs, deterministic=deterministic))
        obs, rew, done, _, _ = env.step_device(act.contiguous())
        cur_ret += rew.to(th.float64)
        cur_len += 1
        fin = (done > 0) & (counts < targets)
        if bool(fin.any()):
            rets += cur_ret[fin].cpu().tolist()
            lens += cur_len[fin].cpu().tolist()
            counts += fin.long()
            cur_ret[fin] = 0.0
            cur_len[fin] = 0
    return rets, lens


def _evaluate_on_host(model, env, targets: np.ndarray, deterministic: bool, render: bool, callback):
    n_envs = env.num_envs
    episode_rewards, episode_lengths = [], []
    episode_counts = np.zeros(n_envs, dtype="int")
    episode_count_targets = targets
    current_rewards = np.zeros(n_envs)
    current_lengths = np.zeros(n_envs, dtype="int")
    observations = env.reset()
    states = None
    episode_starts = np.ones((n_envs,), dtype=bool)
    while (episode_counts < episode_count_targets).any():
        actions, states = model.predict(observations, state=states, episode_start=episode_starts, deterministic=deterministic)
        new_observations, rewards, dones, infos = env.step(actions)
        current_rewards += rewards
        current_lengths += 1
        for i in np.nonzero(episode_counts < episode_count_targets)[0]:
            reward, done, info = rewards[i], dones[i], infos[i]
            episode_starts[i] = done
            if callback is not None:  # locals() once per (vec-step, active env), before the episode is closed (:104-105)
                callback(locals(), globals())
            if dones[i]:
                episode_rewards.append(current_rewards[i])
                episode_lengths.append(current_lengths[i])
                episode_counts[i] += 1
                current_rewards[i] = 0
                current_lengths[i] = 0
        observations = new_observations
        if render:
            env.render()
    return episode_rewards, episode_lengths


def evaluate_policy(model, env, n_eval_episodes: int = 10, deterministic: bool = True, render: bool = False,
                    callback: Optional[Callable] = None, reward_threshold: Optional[float] = None,
                    return_episode_rewards: bool = False, warn: bool = True) -> Union[Tuple[float, float], Tuple[list, list]]:
    if not isinstance(env, VecEnv) and not isinstance(getattr(env, "unwrapped", None), CSTRVecEnv):
        raise ValueError("evaluate_policy: pass a VecEnv of this package (CSTRVecEnv, optionally wrapped in VecNormalize)")
    if warn:  # no Monitor / VecMonitor wrapper exists in this stack: the reference's warning always applies (evaluation.py:64-70)
        warnings.warn(
            "Evaluation environment is not wrapped with a ``Monitor`` wrapper. "
            "This may result in reporting modified episode lengths and rewards, if other wrappers happen to modify these. "
            "Consider wrapping environment first with ``Monitor`` wrapper.",
            UserWarning,
        )
    n = env.num_envs
    targets = np.array([(n_eval_episodes + i) // n for i in range(n)], dtype="int")  # :79-82
    policy = getattr(model, "policy", model)
    on_device = (callback is None and not render and isinstance(getattr(env, "unwrapped", None), CSTRVecEnv)
                 and hasattr(env, "step_device") and hasattr(policy, "_predict") and hasattr(policy, "squash_output"))
    if on_device:
        rets, lens = _evaluate_on_device(policy, env, targets, deterministic)
    else:
        rets, lens = _evaluate_on_host(model, env, targets, deterministic, render, callback)
    mean_reward, std_reward = np.mean(rets), np.std(rets)
    if reward_threshold is not None:
        assert mean_reward > reward_threshold, "Mean reward below threshold: " f"{mean_reward:.2f} < {reward_threshold:.2f}"
    if return_episode_rewards:
        return rets, lens
    return mean_reward, std_reward
