"""Data-parallel path on CPU: two processes, gloo backend (the GPU run uses the same code over RCCL).

Covers: rank/world discovery from the torchrun environment, the flat-arena gradient all-reduce
(sum over ranks, 1/world folded into the optimiser) being equal to a single-process step on the concatenated
batch, rank-0 parameter broadcast, and the shard seeding rule seed_r = seed + rank * n_envs.
"""
import os
import socket

import numpy as np
import torch as th
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _mlp(seed):
    th.manual_seed(seed)
    return th.nn.Sequential(th.nn.Linear(6, 32), th.nn.ReLU(), th.nn.Linear(32, 32), th.nn.ReLU(), th.nn.Linear(32, 1))


def _worker(rank, world, port, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "pytorch-rl-enhancedstablebaselines_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    th.set_num_threads(1)
    from core.common import distributed as du
    from core.common.arena import ParamArena

    r, lr_, w = du.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and du.rank_world() == (rank, world) and du.is_distributed()
    assert du.graph_collectives_ok("cpu") is False  # only RCCL on a GPU qualifies for collectives inside hipGraphs

    # different init per rank, then rank-0 broadcast of the flat arena
    net = _mlp(100 + rank)
    arena = ParamArena(net.parameters(), "cpu")
    du.broadcast_(arena.flat, 0)
    ref0 = _mlp(100)
    for p, q in zip(net.parameters(), ref0.parameters()):
        assert th.equal(p.detach(), q.detach())

    # each rank: local batch -> backward into the flat gradient arena -> summing all-reduce
    g = th.Generator().manual_seed(7)
    x, y = th.randn(2 * 16, 6, generator=g), th.randn(2 * 16, 1, generator=g)
    xs, ys = x[rank * 16:(rank + 1) * 16], y[rank * 16:(rank + 1) * 16]
    arena.zero_grad()
    th.nn.functional.mse_loss(net(xs), ys).backward()
    for p, o in zip(arena.params, arena.offsets):  # autograd accumulated IN PLACE into the arena views
        assert p.grad.data_ptr() == arena.grad.data_ptr() + 4 * o
    du.allreduce_sum_(arena.grad)
    mean_grad = arena.grad * (1.0 / world)  # the Adam kernel's grad_scale

    # single-process reference: mean loss over the concatenated batch
    th.nn.functional.mse_loss(ref0(x), y).backward()
    for p, o in zip(ref0.parameters(), arena.offsets):
        np.testing.assert_allclose(mean_grad[o:o + p.numel()].view(p.shape).numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-7)
    assert du.shard_seed(5, rank, 4096) == 5 + rank * 4096 and du.shard_seed(None, rank, 4096) is None
    assert abs(du.allreduce_mean_scalar(float(rank)) - (world - 1) / 2) < 1e-12
    th.save(mean_grad, os.path.join(out_dir, f"g{rank}.pt"))
    th.distributed.barrier()
    th.distributed.destroy_process_group()


def test_two_rank_gloo_allreduce_equals_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = th.load(tmp_path / "g0.pt"), th.load(tmp_path / "g1.pt")
    assert th.equal(g0, g1)  # every rank holds the same averaged gradient -> identical Adam steps -> weights stay in sync


def test_shard_seeds_are_the_global_env_seeds():
    """Rank r / env i gets seed + r*N + i: exactly the seeds one process with W*N envs would hand out
    (reference: core/common/vec_env/base_vec_env.py:308), and the rank's sampler stream seed is seed_r + N - 1."""
    from core.common import distributed as du

    seed, N, W = 11, 8, 4
    single = [seed + i for i in range(W * N)]
    sharded = [du.shard_seed(seed, r, N) + i for r in range(W) for i in range(N)]
    assert sharded == single
    assert [du.shard_seed(seed, r, N) + N - 1 for r in range(W)] == [seed + r * N + N - 1 for r in range(W)]


def test_single_process_defaults():
    from core.common import distributed as du

    assert du.rank_world() == (0, 1) and not du.is_distributed() and du.graph_collectives_ok("cpu") is False
    t = th.ones(4)
    assert du.allreduce_sum_(t) is t and th.equal(t, th.ones(4))
