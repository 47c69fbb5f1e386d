#!/usr/bin/env python3
"""A/B of the rollout's whole-policy kernel at the bench shape (4096 x 4 -> 256 -> 256 -> 2x2): the first version
(CSTR_POLICY_V1=1 in a child process) against the software-pipelined one, graph-replayed launches timed with HIP events,
interleaved rounds. Also the TD3 shape (4 -> 400 -> 300 -> 2, deterministic head)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)


def child():
    import torch as th

    sys.path.insert(0, ROOT)
    from bench import event_time_us
    from core.common import hip_ops

    out = {}
    r = lambda *s: th.randn(*s, device="cuda")  # noqa: E731
    for name, (m, k0, h1, h2, a, head) in dict(sac=(4096, 4, 256, 256, 2, 0), td3=(4096, 4, 400, 300, 2, 1), sac8=(4096, 8, 256, 256, 2, 0)).items():
        x, w1, b1, w2, b2 = r(m, k0), r(h1, k0), r(h1), r(h2, h1) / h1 ** 0.5, r(h2)
        n_out = 2 * a if head == 0 else a
        w3, b3 = r(n_out, h2) / 16, r(n_out)
        ctl, act, tiles = hip_ops.new_rng_ctl(1, "cuda"), th.empty(m, a, device="cuda"), hip_ops.policy_swizzle(w2)
        for defer in (False, True):
            if head == 1 and defer:
                continue
            fn = lambda: hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, head, 2 if head else 0, act, rng_ctl=ctl if head == 0 else None,  # noqa: E731
                                                 w2_swz=tiles, defer_rng_advance=defer)
            ts = [event_time_us(fn, 200, th.cuda.current_stream(), in_graph=True) for _ in range(5)]
            flops = 2.0 * m * (k0 * h1 + h1 * h2 + h2 * n_out)
            out[f"{name}{'_defer' if defer else ''}"] = dict(us=round(min(ts), 3), us_all=[round(t, 2) for t in ts], frac=round(flops / min(ts) / 1e6 / 157.3, 4))
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        res = {}
        for tag, env in (("v1", dict(CSTR_POLICY_V1="1")), ("v2", {})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            res[tag] = json.loads(line[-1]) if line else dict(error=p.stderr[-2000:])
        print(json.dumps(res, indent=1))
