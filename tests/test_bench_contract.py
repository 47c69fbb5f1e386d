"""bench.py's driver contract: `python bench.py --gpus N` starts N ranks by itself, proves the rank
count on the JSON line, and refuses to report a run that is not what it says it is."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_without_enough_devices_is_refused_not_relabelled():
    """No launcher, no GPUs (this container) or fewer than N: non-zero exit, no JSON line (round 1 silently ran 1 rank)."""
    import torch as th

    if th.cuda.device_count() >= 2:
        pytest.skip("needs a node with < 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "2", "--warmup", "1"], env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "refusing to label" in r.stderr


def test_a_rank_dying_during_startup_fails_the_job_quickly(monkeypatch):
    """VERDICT r2 next-3 / ADVICE: rank 1 exits 7 before the rendezvous (CSTR_BENCH_FAIL_RANK). The parent polls every child,
    terminates the siblings (rank 0 would otherwise sit in the rendezvous until the backend's timeout) and returns non-zero
    in seconds, with no JSON line. Runs on the CPU: the ranks never get as far as the GPU."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"],
             env=dict(CSTR_DIST_BACKEND="gloo", CSTR_BENCH_SINGLE_DEVICE="1", CSTR_BENCH_FAIL_RANK="1"), timeout=120)
    took = time.monotonic() - t0
    assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout, r.stderr[-2000:])
    assert "rank 1 exited with code 7" in r.stderr and "siblings terminated" in r.stderr
    assert took < 30, took


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
def test_bench_gpus_n_launches_its_ranks_by_itself(n):
    """The plain command the driver's SCALE run issues, rehearsed on ONE MI355X: all ranks on cuda:0, gloo transport (RCCL
    refuses two ranks on one device). Asserts the line's own proof of the rank count and the honesty keys."""
    r = _run(["--gpus", str(n), "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--no-roofline"],
             env=dict(CSTR_DIST_BACKEND="gloo", CSTR_BENCH_SINGLE_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["rccl_world"] == n and rec["config"]["parallelism"] == f"dp{n}"
    assert rec["allreduce_checksum"] == dict(expected=n * (n + 1) // 2, got=n * (n + 1) // 2, backend="gloo")
    assert rec["weights_identical_across_ranks"] is True
    assert rec["graph_collectives"] == "segmented"  # gloo collectives are never recorded into a hipGraph
    assert rec["hip_graph_active"] is True and rec["eager_iterations_in_timed_region"] == 0
    assert rec["hip_graph_replays_in_timed_region"] == rec["timed_steps_total"] >= rec["steps"] == 10
    assert rec["timed_seconds_total"] >= 0.25  # a 10-step region is ~2 ms: it must have been repeated
    assert rec["config"]["global_batch"] == 256 * n and rec["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_roofline_baseline_and_guards():
    r = _run(["--steps", "20", "--warmup", "5", "--cpu-seconds", "2", "--no-variant"])
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = [ln for ln in r.stdout.splitlines() if ln.strip()]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["rccl_world"] == 1 and rec["steps"] == 20
    assert rec["timed_steps_total"] % 20 == 0 and rec["timed_seconds_total"] >= 0.25
    assert rec["hip_graph_active"] is True and rec["hip_graph_error"] is None
    # roofline = the dominant kernel of the REPLAYED graph, the collect kernel alone under another key
    assert rec["roofline"]["kernel"] == "rollout_step_kernel" and rec["roofline"]["bound"] == "mfma" and 0 < rec["roofline"]["frac"] < 1
    assert rec["roofline"]["hbm"]["algorithmic_bytes_per_launch"] >= 4096 * 104 and "traffic_source" in rec["roofline"]
    assert rec["roofline_collect"]["bound"] == "hbm" and rec["roofline_stream"]["bound"] == "hbm" and rec["roofline_mfma"]["bound"] == "mfma"
    it = rec["roofline_iteration"]
    assert 2.0e6 < it["algorithmic_bytes"] < 2.2e6 and 1.0e9 < it["algorithmic_flops"] < 1.6e9  # 2.09 MB; ~1.25 GFLOP
    assert it["launches_per_iteration"] == rec["config"]["launches_per_iteration"] and 5 <= it["launches_per_iteration"] <= 30
    assert abs(it["mean_launch_interval_us"] * it["launches_per_iteration"] - 1e3 * rec["ms_per_step"]) < 0.5
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["cpu_model"] and cb["single_thread"]["cores"] == 1 and cb["single_thread"]["value"] > 0
    assert rec["reference_python_env_steps_per_s"]["value"] == 4200.0 and "cross_machine_ratio" in rec["reference_python_env_steps_per_s"]
    assert rec["config"]["graph_unroll"] == 8 and "8 iterations per graph" in rec["config"]["workload"]


@pytest.mark.gpu
def test_bench_fails_when_graph_capture_falls_back():
    """A capture regression must not pass as `hip_graph: true`: CSTR_BENCH_BREAK_CAPTURE makes the captured body raise."""
    r = _run(["--steps", "5", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-variant"], env=dict(CSTR_BENCH_BREAK_CAPTURE="1"))
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "INVALID RUN" in r.stderr
    (line,) = [ln for ln in r.stdout.splitlines() if ln.strip()]
    rec = json.loads(line)
    assert rec["hip_graph_active"] is False and "injected" in rec["hip_graph_error"]
