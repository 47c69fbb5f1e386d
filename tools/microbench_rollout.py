#!/usr/bin/env python3
"""Launch the one-launch rollout (cstr_rollout_step_f32) + the gather launch at the bench shape (4096 envs, 4 -> 256 -> 256 -> 2x2,
batch 256) for rocprofv3 --pmc / --kernel-trace passes (BENCH=tools/microbench_rollout.py KERNEL=rollout_step tools/pmc_policy.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch as th  # noqa: E402

from core.common import hip_ops  # noqa: E402
from test_rollout_step import _World  # noqa: E402

if __name__ == "__main__":
    w = _World(hip_ops, 4096, 4, 244, 256, 256, 256, 0, seed=1, max_steps=400)
    w.step_count.zero_()
    for _ in range(200):
        w.step_fused(None)
    th.cuda.synchronize()
