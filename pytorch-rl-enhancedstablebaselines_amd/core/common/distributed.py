"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in CPU tests). The reference has no distributed code at all; the design is
SURVEY 8e: every rank owns n_envs environments, its own replay ring and its own sampler stream; the only
exchange is ONE summing all-reduce per optimiser step on that optimiser's flat gradient arena. The 1/world
scale is folded into the Adam kernel (`grad_scale`), so gradient averaging costs no extra pass.

Message sizes are tiny (SAC: 543 KB critic + entropy coefficient, 272 KB actor per step): the collective is
latency-bound, not xGMI link-bound, so the two dependent all-reduces are issued as-is on the training stream and
RCCL picks its low-latency (tree / one-shot) protocol; no bucketing beyond the per-optimiser arena is useful.
Under hipGraph replay the collectives are recorded into the iteration's graph when `graph_collectives_ok` (a start-up
capture-and-replay trial on every rank) passes, and stay between graph segments otherwise
(OffPolicyAlgorithm._capture_segments; CSTR_GRAPH_COLLECTIVES=auto|0|1).
"""
import datetime
import os
import sys
from typing import Optional, Tuple

import torch as th
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns
    (rank, local_rank, world). No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if th.cuda.is_available() else "gloo"
        # a rank that never shows up (died during start-up) must fail the rendezvous / the first collective in minutes, not
        # after the backend's default (10 min RCCL, 30 min gloo)
        timeout = datetime.timedelta(seconds=float(os.environ.get("CSTR_DIST_TIMEOUT_S", "120")))
        if backend == "nccl":
            th.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=th.device("cuda", local_rank), timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local_rank, world


def shard_seed(seed: Optional[int], rank: int, n_envs: int) -> Optional[int]:
    """seed_r = seed + rank * n_envs: env i of rank r gets seed_r + i -- the globally unique seeds a single
    process with world * n_envs envs would hand out (reference base_vec_env.py:308); the rank's sampler stream
    ends up seeded seed_r + n_envs - 1 like an independent reference run with seed = seed_r."""
    return None if seed is None else seed + rank * n_envs


def allreduce_sum_(flat: th.Tensor) -> th.Tensor:
    """In-place summing all-reduce of one flat gradient arena (no-op for world 1)."""
    if is_distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


_GRAPH_COLLECTIVES_OK = None


def _agree(ok: bool, device) -> bool:
    """MIN over ranks of a local verdict. `th.full` is a fill kernel (no host-to-device copy), safe right after a capture."""
    verdict = th.full((1,), 1.0 if ok else 0.0, dtype=th.float32, device=device)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    return bool(verdict.item() == 1.0)


def graph_collectives_ok(device) -> bool:
    """Start-up trial for recording collectives into hipGraphs: capture ONE all-reduce, replay it twice and check the sums on
    every rank. Only the RCCL backend on a GPU qualifies. Must be called by every rank, outside any capture.

    Every rank issues the SAME sequence of collectives whatever happens locally: [warm all-reduce] [captured all-reduce, not
    executed] [MIN verdict "captured"] and, only if every rank captured, [replay, replay] [MIN verdict "sums right"]. A rank
    whose capture raises leaves capture mode (capture_end) before anything else touches the stream or the graph object is
    destroyed -- a graph destroyed mid-capture aborts the process (round 1, rc 134 in CUDAGraph::~CUDAGraph)."""
    global _GRAPH_COLLECTIVES_OK
    if _GRAPH_COLLECTIVES_OK is not None:
        return _GRAPH_COLLECTIVES_OK
    device = th.device(device)
    if not is_distributed() or dist.get_backend() != "nccl" or device.type != "cuda":
        _GRAPH_COLLECTIVES_OK = False
        return False
    import gc

    rank, world = dist.get_rank(), dist.get_world_size()
    with th.cuda.device(device):
        t = th.zeros(4096, dtype=th.float32, device=device)
        dist.all_reduce(t)  # the communicator and its buffers exist before anything is recorded
        th.cuda.synchronize(device)
        side, g = th.cuda.Stream(device=device), th.cuda.CUDAGraph()
        side.wait_stream(th.cuda.current_stream(device))
        captured = False
        gc.collect()  # like torch.cuda.graph(): a collection that frees device memory mid-capture invalidates the capture
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            with th.cuda.stream(side):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    dist.all_reduce(t)
                    g.capture_end()
                    captured = True
                except Exception as exc:  # noqa: BLE001 -- any failure means "keep the collectives between graph segments"
                    try:
                        g.capture_end()  # leave capture mode FIRST
                    except Exception:  # noqa: BLE001
                        pass
                    print(f"[distributed] collectives stay outside hipGraphs (capture failed): {exc!r}", file=sys.stderr)
        finally:
            if gc_was_enabled:
                gc.enable()
        th.cuda.current_stream(device).wait_stream(side)
        th.cuda.synchronize(device)
        if not _agree(captured, device):
            del g
            _GRAPH_COLLECTIVES_OK = False
            return False
        ok = True
        try:
            for k in (1.0, 3.0):
                t.fill_(k * (rank + 1))
                g.replay()
                th.cuda.synchronize(device)
                ok = ok and bool((t == k * world * (world + 1) / 2).all())
        except Exception as exc:  # noqa: BLE001
            print(f"[distributed] collectives stay outside hipGraphs (replay failed): {exc!r}", file=sys.stderr)
            ok = False
        _GRAPH_COLLECTIVES_OK = _agree(ok, device)
    return _GRAPH_COLLECTIVES_OK


def broadcast_(t: th.Tensor, src: int = 0) -> th.Tensor:
    if is_distributed():
        dist.broadcast(t, src=src)
    return t


def allreduce_mean_scalar(x: float) -> float:
    if not is_distributed():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = th.tensor([x], dtype=th.float64, device=dev)
    dist.all_reduce(t)
    return float(t) / dist.get_world_size()
