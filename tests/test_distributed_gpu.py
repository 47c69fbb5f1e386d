"""Two data-parallel ranks on ONE MI355X (both on cuda:0, gloo transport for the collectives -- RCCL refuses two ranks
on one device): the full product path with real all-reduces between the captured hipGraph segments.
Checks SURVEY 8e: identical weights on every rank after training (same averaged gradients), different env shards
(seed_r = seed + rank * n_envs), different sampler streams (seed_r + n_envs - 1), per-rank rings."""
import os
import socket

import pytest
import torch as th
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, use_graph):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "pytorch-rl-enhancedstablebaselines_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from core.common import distributed as du
    from core.common import legacy_rng
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    du.init_from_env(backend="gloo")
    N, B, seed, iters = 64, 32, 11, 14
    env = CSTRVecEnv(N, device="cuda:0")
    model = SAC("MlpPolicy", env, seed=seed, batch_size=B, buffer_size=N * 8, learning_starts=100, device="cuda:0",
                policy_kwargs=dict(net_arch=[32, 32]))
    assert model.world_size == world and model.rank == rank and env.seed_offset == rank * N
    assert model.actor.optimizer.grad_scale == 1.0 / world
    model.enable_graph_capture(use_graph)
    model.learn(N * iters)
    th.cuda.synchronize()
    if use_graph:
        (segs,) = model._graph.values()
        assert sum(isinstance(s, th.cuda.CUDAGraph) for s in segs) == 3  # 2 all-reduces (critic + alpha, actor) -> 3 graphs
    th.save(dict(actor=model.policy.actor_arena.flat.cpu(), critic=model.policy.critic_arena.flat.cpu(),
                 target=model.policy.critic_target_arena.flat.cpu(), alpha=model.log_ent_coef.detach().cpu(),
                 obs=env.obs.cpu(), mt=legacy_rng.global_stream(model.device).cpu(), n_updates=model._n_updates,
                 sampler_seed=legacy_rng.last_seed(model.device), ring_obs=model.replay_buffer.observations[0].cpu()),
            os.path.join(out_dir, f"r{rank}_{int(use_graph)}.pt"))
    th.distributed.barrier()
    th.distributed.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_one_gpu_stay_in_sync(tmp_path, use_graph):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), use_graph), nprocs=world, join=True)
    r0, r1 = (th.load(tmp_path / f"r{r}_{int(use_graph)}.pt") for r in range(world))
    assert r0["n_updates"] == r1["n_updates"] == 13
    for k in ("actor", "critic", "target", "alpha"):  # same initial weights + same averaged gradients -> same weights
        assert th.equal(r0[k], r1[k]), k
    assert not th.equal(r0["obs"], r1["obs"]) and not th.equal(r0["ring_obs"], r1["ring_obs"])  # different env shards
    assert not th.equal(r0["mt"], r1["mt"])
    assert r0["sampler_seed"] == 11 + 64 - 1 and r1["sampler_seed"] == 11 + 64 + 64 - 1  # seed_r + n_envs - 1


def _rccl_worker(rank, port, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "pytorch-rl-enhancedstablebaselines_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from core.common import distributed as du
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    th.cuda.set_device(0)
    th.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                      device_id=th.device("cuda", 0))
    du.is_distributed = lambda: True  # a world of one rank still issues its all-reduces (through RCCL)
    assert du.graph_collectives_ok("cuda:0")  # the start-up trial: capture + replay of an RCCL all-reduce
    N, B, iters, out = 64, 32, 14, {}
    for mode in ("eager", "segmented", "in_graph", "in_graph_x4"):
        env = CSTRVecEnv(N, device="cuda:0")
        model = SAC("MlpPolicy", env, seed=5, batch_size=B, buffer_size=N * 8, learning_starts=100, device="cuda:0",
                    policy_kwargs=dict(net_arch=[32, 32]))
        model._force_segment_boundaries = True  # the data-parallel launch structure
        model._graph_collectives = mode.startswith("in_graph")
        model.enable_graph_capture(mode != "eager", unroll=4 if mode == "in_graph_x4" else 1)  # x4: four iterations (8 all-reduces) per graph
        model.learn(N * iters)
        th.cuda.synchronize()
        if mode == "in_graph_x4":
            assert any(key[-1] == 4 for key in model._graph) and all(len(segs) == 1 for segs in model._graph.values())
        elif mode != "eager":
            (segs,) = model._graph.values()
            assert sum(isinstance(s, th.cuda.CUDAGraph) for s in segs) == (1 if mode == "in_graph" else 3)
        out[mode] = dict(actor=model.policy.actor_arena.flat.cpu(), critic=model.policy.critic_arena.flat.cpu(),
                         alpha=model.log_ent_coef.detach().cpu(), obs=env.obs.cpu(), n_updates=model._n_updates)
    th.save(out, os.path.join(out_dir, "rccl.pt"))
    th.distributed.destroy_process_group()


def test_collectives_inside_the_graph_match_segmented_and_eager(tmp_path):
    """The N > 1 launch structure in an RCCL world of ONE rank (all this box can hold): the all-reduces recorded INTO the
    iteration graph (after the start-up trial passed; also four iterations = eight all-reduces per graph) train bit-identically to
    collectives between graph segments and to the eager loop."""
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    out = th.load(tmp_path / "rccl.pt")
    for mode in ("segmented", "in_graph", "in_graph_x4"):
        assert out[mode]["n_updates"] == out["eager"]["n_updates"] == 13
        for k in ("actor", "critic", "alpha", "obs"):
            assert th.equal(out[mode][k], out["eager"][k]), (mode, k)


def _capture_failure_worker(rank, port, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "pytorch-rl-enhancedstablebaselines_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from core.common import distributed as du

    th.cuda.set_device(0)
    th.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                      device_id=th.device("cuda", 0))
    du.is_distributed = lambda: True
    real, calls = du.dist.all_reduce, []

    def flaky(tensor, *a, **kw):
        calls.append(bool(th.cuda.is_current_stream_capturing()))
        if calls[-1]:
            raise RuntimeError("injected: collective refused under capture")
        return real(tensor, *a, **kw)

    du.dist.all_reduce = flaky
    ok = du.graph_collectives_ok("cuda:0")
    du.dist.all_reduce = real
    # the process survived, the stream left capture mode, and ordinary work + collectives still run
    x = th.arange(8, dtype=th.float32, device="cuda:0")
    real(x)
    th.cuda.synchronize()
    th.save(dict(ok=ok, calls=calls, x=x.cpu(), capturing=bool(th.cuda.is_current_stream_capturing())),
            os.path.join(out_dir, "capfail.pt"))
    th.distributed.destroy_process_group()


def test_graph_collectives_trial_survives_a_failed_capture(tmp_path):
    """a collective that raises while being captured must not leave the stream capturing nor destroy the graph
    mid-capture; the trial returns False after the SAME collective sequence as a healthy rank up to the first verdict
    (warm all-reduce, captured attempt, MIN verdict)."""
    mp.spawn(_capture_failure_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    out = th.load(tmp_path / "capfail.pt")
    assert out["ok"] is False and out["capturing"] is False
    assert out["calls"] == [False, True, False]  # warm-up, the captured attempt, the "captured" verdict; no replay verdict
    assert th.equal(out["x"], th.arange(8, dtype=th.float32))


def _full_shape_worker(rank, world, port, out_dir, use_graph):
    import sys

    import numpy as np

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "pytorch-rl-enhancedstablebaselines_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from core.common import distributed as du
    from core.common import legacy_rng
    from core.common.vec_env import CSTRVecEnv
    from core.sac import SAC

    du.init_from_env(backend="gloo")
    N, B, seed, iters = 4096, 256, 21, 50
    env = CSTRVecEnv(N, device="cuda:0")
    model = SAC("MlpPolicy", env, seed=seed, device="cuda:0")  # class defaults: nets [256, 256], ring 244 x 4096, batch 256
    assert model.world_size == world and env.seed_offset == rank * N and model.replay_buffer.buffer_size == 244
    model.enable_graph_capture(use_graph)
    model.learn(N * iters)
    th.cuda.synchronize()
    pcg0 = env.pcg_state.cpu().clone()  # per-env reset generators after the seeded first reset (no auto-reset yet at 50 steps)
    st = model.graph_status()
    if use_graph:
        assert st["active"] and st["graph_collectives"] == "segmented" and st["segments_per_graph"] == [3] and st["error"] is None
    # this rank's sampler stream against numpy's: seed_r + N - 1 with seed_r = seed + rank * N, one randint pair per step
    rs = np.random.RandomState(seed + rank * N + N - 1)
    for k in range(1, iters + 1):
        rs.randint(0, min(k, 244), size=B)
        rs.randint(0, N, size=B)
    want, got = rs.get_state(), legacy_rng.global_stream(model.device).cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:624], want[1]) and int(got[624]) == want[2], "sampler stream differs from numpy's"
    th.save(dict(actor=model.policy.actor_arena.flat.cpu(), critic=model.policy.critic_arena.flat.cpu(),
                 target=model.policy.critic_target_arena.flat.cpu(), alpha=model.log_ent_coef.detach().cpu(), obs=env.obs.cpu(),
                 n_updates=model._n_updates, ring_obs0=model.replay_buffer.observations[0].cpu(), pcg0=pcg0,
                 sampler_seed=legacy_rng.last_seed(model.device)), os.path.join(out_dir, f"full_r{rank}_{int(use_graph)}.pt"))
    th.distributed.barrier()
    th.distributed.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_at_config4_per_rank_shape(tmp_path, use_graph):
    """BASELINE config 4's PER-RANK shape (SAC class defaults, 4096 envs, ring 244 x 4096, batch 256) on two data-parallel ranks
    sharing this box's one MI355X (gloo transport; RCCL refuses two ranks on one device), 50 iterations, eager and captured
    (graph | all-reduce | graph | all-reduce | graph): weights bit-identical across ranks, every rank's MT19937 stream equal to
    numpy's for seed + r N + N - 1 (checked inside the rank), env shards and reset streams disjoint. This is as far as the 8-GPU
    path can be verified without the 8-GPU node: the 1 -> 8 scaling curve itself is the driver's measurement."""
    world, port = 2, _free_port()
    mp.spawn(_full_shape_worker, args=(world, port, str(tmp_path), use_graph), nprocs=world, join=True)
    r0, r1 = (th.load(tmp_path / f"full_r{r}_{int(use_graph)}.pt") for r in range(world))
    assert r0["n_updates"] == r1["n_updates"] == 50
    for k in ("actor", "critic", "target", "alpha"):
        assert th.equal(r0[k], r1[k]), k
    assert not th.equal(r0["obs"], r1["obs"]) and not th.equal(r0["ring_obs0"], r1["ring_obs0"])
    assert r0["sampler_seed"] == 21 + 4096 - 1 and r1["sampler_seed"] == 21 + 4096 + 4096 - 1
    import numpy as np

    both = np.concatenate([r0["pcg0"].numpy(), r1["pcg0"].numpy()])  # per-env reset generators seeded seed_r + i: all 8192 streams distinct
    assert len({tuple(row) for row in both.tolist()}) == 2 * 4096
