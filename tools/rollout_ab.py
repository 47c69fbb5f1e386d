#!/usr/bin/env python3
"""A/B of the rollout at the bench shape (4096 envs, 4 -> 256 -> 256 -> 2x2, batch 256): the separate launches (policy, collect,
sampler) against the one-launch rollout (+ gather), each as a graph-replayed chain, HIP events on the launch stream."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch as th  # noqa: E402

from core.common import hip_ops as ops  # noqa: E402
from test_rollout_step import _World  # noqa: E402


def timed(fn, reps=20, replays=50):
    side = th.cuda.Stream()
    with th.cuda.stream(side):
        for _ in range(3):
            fn()
        th.cuda.synchronize()
        g = th.cuda.CUDAGraph()
        with th.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        for _ in range(5):
            g.replay()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(replays):
            g.replay()
        e1.record(side)
        th.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


if __name__ == "__main__":
    n = int(os.environ.get("N", "4096"))
    w = _World(ops, n, 4, 244, 256, 256, 256, int(os.environ.get("HEAD", "0")), seed=1, max_steps=400)
    w.step_count.zero_()
    pol = th.empty(n, 2, device="cuda")
    hd, oa = w.head, (2 if w.head else 0)

    def policy():
        ops.policy_rows_fwd(w.env_obs, *w.w, 1, hd, oa, pol, rng_ctl=w.rng_ctl, w2_swz=w.swz, defer_rng_advance=True)

    def collect():
        ops.collect_step(w.coef, "euler", w.ring, w.env_obs, w.step_count, pol, 1, w.low, w.high, pcg_state=w.pcg, reward_out=w.rew,
                         done_out=w.done, ep_return=w.ep_return, ep_stats=w.ep_stats, rng_advance=None if w.rng_ctl is None else (w.rng_ctl, n))

    def sampler():
        ops.replay_sample_packed(w.ring, w.mt, w.batch, w.x_data, w.x_next, w.x_pi, w.s_done, w.s_rew)

    def rollout(mt=True):
        ops.rollout_step(w.env_obs, *w.w, 1, hd, oa, w.swz, w.rng_ctl, w.coef, "euler", w.ring, w.env_obs, w.step_count, 1, w.low, w.high,
                         pcg_state=w.pcg, reward_out=w.rew, done_out=w.done, ep_return=w.ep_return, ep_stats=w.ep_stats,
                         mt_state=w.mt if mt else None, sample_idx=w.idx if mt else None)

    def gather():
        ops.replay_gather_packed(w.ring, w.idx, w.batch, w.x_data, w.x_next, w.x_pi, w.s_done, w.s_rew, advance_ring=True,
                                 rng_advance=None if w.rng_ctl is None else (w.rng_ctl, n))

    w.step_separate(None)  # a row in the ring
    rows = [("policy", policy), ("policy + collect", lambda: (policy(), collect())),
            ("policy + collect + sampler", lambda: (policy(), collect(), sampler())),
            ("rollout without the index draw", lambda: rollout(False)), ("rollout", rollout),
            ("rollout + gather", lambda: (rollout(), gather())), ("sampler", sampler), ("gather", gather), ("collect", collect)]
    for name, fn in rows:
        print(f"{name:40s} {timed(fn):7.2f} us per chain")
    del np
