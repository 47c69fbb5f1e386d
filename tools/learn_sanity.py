#!/usr/bin/env python3
"""End-to-end sanity on the GPU: does SAC / TD3 on the device CSTR env actually improve the episode return?
usage: learn_sanity.py [algo=sac] [n_envs=256] [iters=20000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)
import torch as th  # noqa: E402

from core.common.evaluation import evaluate_policy  # noqa: E402
from core.common.vec_env import CSTRVecEnv  # noqa: E402
from core.sac import SAC  # noqa: E402
from core.td3 import TD3  # noqa: E402

if __name__ == "__main__":
    algo = sys.argv[1] if len(sys.argv) > 1 else "sac"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    env = CSTRVecEnv(n)
    eval_env = CSTRVecEnv(64)
    eval_env.seed(1234)
    cls = SAC if algo == "sac" else TD3
    model = cls("MlpPolicy", env, seed=0, learning_starts=n * 10)
    model.enable_graph_capture()
    before = evaluate_policy(model, eval_env, n_eval_episodes=64)
    t0 = time.time()
    done = 0
    for chunk in range(5):
        model.learn(n * iters // 5, reset_num_timesteps=(chunk == 0))
        th.cuda.synchronize()
        eval_env.seed(1234)
        r = evaluate_policy(model, eval_env, n_eval_episodes=64)
        print(f"{algo} after {model.num_timesteps} env-steps / {model._n_updates} updates ({time.time() - t0:.1f} s): "
              f"eval return {r[0]:.2f} +- {r[1]:.2f}  (before training {before[0]:.2f})", flush=True)
