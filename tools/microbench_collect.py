#!/usr/bin/env python3
"""Launch the fused collect kernel alone (for rocprofv3 --pmc / --kernel-trace passes).
usage: microbench_collect.py [n_envs=4194304] [launches=20] [obs_dim=4] [integrator=euler]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    integ = sys.argv[4] if len(sys.argv) > 4 else "euler"
    import json

    print(json.dumps(bench.roofline_collect(n, d, integ, launches)))
