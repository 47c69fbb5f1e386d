"""`evaluate_policy` over a device-resident CSTRVecEnv (reference: core/common/evaluation.py:11-140): runs
`n_eval_episodes` complete episodes spread over the env's sub-environments, all stepping on the GPU; only the final
per-episode returns / lengths come back to the host."""
from typing import Callable, Optional, Tuple, Union

import numpy as np
import torch as th

from core.common.vec_env import CSTRVecEnv


def evaluate_policy(model, env, n_eval_episodes: int = 10, deterministic: bool = True, render: bool = False,
                    callback: Optional[Callable] = None, reward_threshold: Optional[float] = None,
                    return_episode_rewards: bool = False, warn: bool = True) -> Union[Tuple[float, float], Tuple[list, list]]:
    if not isinstance(getattr(env, "unwrapped", None), CSTRVecEnv):
        raise ValueError("evaluate_policy: this stack evaluates on a CSTRVecEnv (optionally wrapped in VecNormalize)")
    n = env.num_envs
    # reference :79-82: episodes are split over the envs as evenly as possible
    targets = th.tensor([(n_eval_episodes + i) // n for i in range(n)], device=env.device)
    counts = th.zeros(n, dtype=th.long, device=env.device)
    cur_ret = th.zeros(n, device=env.device)
    cur_len = th.zeros(n, dtype=th.long, device=env.device)
    rets, lens = [], []
    obs = env.reset_device()
    policy = model.policy
    policy.set_training_mode(False)
    while bool((counts < targets).any()):
        with th.no_grad():
            if deterministic and hasattr(policy, "actor") and hasattr(policy.actor, "get_action_dist_params"):
                act = policy._predict(obs, deterministic=True)
            else:
                act = policy._predict(obs, deterministic=deterministic)
        obs, rew, done, _, _ = env.step_device(act.contiguous())
        active = counts < targets
        cur_ret += rew * active
        cur_len += active.long()
        fin = (done > 0) & active
        if bool(fin.any()):
            rets += cur_ret[fin].cpu().tolist()
            lens += cur_len[fin].cpu().tolist()
            counts += fin.long()
            cur_ret[done > 0] = 0.0
            cur_len[done > 0] = 0
    mean_reward, std_reward = float(np.mean(rets)), float(np.std(rets))
    if reward_threshold is not None:
        assert mean_reward > reward_threshold, f"Mean reward below threshold: {mean_reward:.2f} < {reward_threshold:.2f}"
    if return_episode_rewards:
        return rets, lens
    return mean_reward, std_reward
