"""`OffPolicyAlgorithm`: the collect -> store -> sample -> train runtime
(reference: core/common/off_policy_algorithm.py:27-605), rebuilt around a device-resident loop.

Reference iteration (per vec-step): python loop over N envs, deepcopy of N info dicts, 5 H2D copies per
sample, 4 `.item()` syncs per gradient step. Here, when the env is a `CSTRVecEnv` and the buffer is the HBM
`ReplayBuffer`, one iteration is: actor forward (one whole-network HIP launch above 1024 envs) -> ONE fused HIP launch
(action scaling chain + env step + auto-reset + ring row write + episode statistics) -> `train()`. Nothing
returns to the host except, every `stats_sync_interval` vec-steps, four doubles of episode statistics.
Any other VecEnv goes through the NumPy compatibility path with the reference's exact semantics.
"""
import os
import sys
import time
from typing import Any, Optional, Union

import numpy as np
import torch as th

from core.common import blas
from core.common import distributed as dist_util
from core.common import hip_ops
from core import _native as nv
from core.common.base_class import BaseAlgorithm
from core.common.buffers import ReplayBuffer
from core.common.callbacks import BaseCallback, MaybeCallback, to_callback
from core.common.type_aliases import RolloutReturn, TrainFreq, TrainFrequencyUnit
from core.common.utils import should_collect_more_steps
from core.common.vec_env import CSTRVecEnv, VecEnv

# Data-parallel runs under hipGraph replay. "auto" (default): record the RCCL all-reduces INTO the iteration's graph when a
# start-up trial (distributed.graph_collectives_ok: capture + replay of one all-reduce, result checked on every rank) passes,
# else run them eagerly BETWEEN graph segments; "0": always between segments; "1": always inside (no trial).
GRAPH_COLLECTIVES = os.environ.get("CSTR_GRAPH_COLLECTIVES", "auto")
# the captured iteration's rollout as ONE launch (policy + collect step + replay index draw; hip_ops.rollout_step). "0": the
# separate policy / collect / sampler launches (A/B knob; both forms are bit-identical, tests/test_rollout_step.py)
FUSED_ROLLOUT = os.environ.get("CSTR_FUSED_ROLLOUT", "1") != "0"


class OffPolicyAlgorithm(BaseAlgorithm):
    def __init__(self, policy, env, learning_rate, buffer_size: int = 1_000_000, learning_starts: int = 100,
                 batch_size: int = 256, tau: float = 0.005, gamma: float = 0.99, train_freq: Union[int, tuple] = (1, "step"),
                 gradient_steps: int = 1, action_noise=None, replay_buffer_class=None, replay_buffer_kwargs: Optional[dict] = None,
                 optimize_memory_usage: bool = False, policy_kwargs: Optional[dict] = None, stats_window_size: int = 100,
                 tensorboard_log: Optional[str] = None, verbose: int = 0, device="auto", support_multi_env: bool = False,
                 monitor_wrapper: bool = True, seed: Optional[int] = None, use_sde: bool = False, sde_sample_freq: int = -1,
                 use_sde_at_warmup: bool = False, sde_support: bool = True, supported_action_spaces: Optional[tuple] = None):
        super().__init__(policy=policy, env=env, learning_rate=learning_rate, policy_kwargs=policy_kwargs,
                         stats_window_size=stats_window_size, tensorboard_log=tensorboard_log, verbose=verbose, device=device,
                         support_multi_env=support_multi_env, monitor_wrapper=monitor_wrapper, seed=seed, use_sde=use_sde,
                         sde_sample_freq=sde_sample_freq, supported_action_spaces=supported_action_spaces)
        self.buffer_size = buffer_size
        self.batch_size = batch_size
        self.learning_starts = learning_starts
        self.tau = tau
        self.gamma = gamma
        self.gradient_steps = gradient_steps
        self.action_noise = action_noise
        self.optimize_memory_usage = optimize_memory_usage
        self.replay_buffer: Optional[ReplayBuffer] = None
        self.replay_buffer_class = replay_buffer_class
        self.replay_buffer_kwargs = replay_buffer_kwargs or {}
        self.train_freq = train_freq
        self.use_sde_at_warmup = use_sde_at_warmup
        self._graph_enabled, self._graph, self._graph_key = False, None, None
        self._rng_advance = None  # (rng_ctl, rows) the next fused collect launch owes the rollout policy launch (SAC)
        self._graph_error: Optional[str] = None  # text of the exception that ended hipGraph replay (None = never failed)
        self._graph_replays = 0                  # iterations served by a captured graph / by eager launches (bench.py reports both)
        self._eager_iterations = 0
        self.stats_sync_interval = 100   # vec-steps between host reads of the device episode counters
        self._steps_since_sync = 0
        self._episodes_at_last_dump = 0
        self._ep_window = (0.0, 0.0, 0.0)

    # ---- setup ----------------------------------------------------------------------------------------------------
    def _convert_train_freq(self) -> None:
        """reference: off_policy_algorithm.py:148-170"""
        if not isinstance(self.train_freq, TrainFreq):
            train_freq = self.train_freq
            if not isinstance(train_freq, tuple):
                train_freq = (train_freq, "step")
            try:
                train_freq = (train_freq[0], TrainFrequencyUnit(train_freq[1]))
            except ValueError as e:
                raise ValueError(f"The unit of the `train_freq` must be either 'step' or 'episode' not '{train_freq[1]}'!") from e
            if not isinstance(train_freq[0], int):
                raise ValueError(f"The frequency of `train_freq` must be an integer and not {train_freq[0]}")
            self.train_freq = TrainFreq(*train_freq)

    def _setup_model(self) -> None:
        """reference: off_policy_algorithm.py:172-212"""
        self._chain_cache = {}  # row-chain step objects hold raw pointers of the policy's tensors: rebuilt with the policy
        self._setup_lr_schedule()
        blas.configure()
        if self.world_size > 1 and self._denv is not None:
            self._denv.seed_offset = self.rank * self.n_envs  # seed_r = seed + rank * n_envs
        self.set_random_seed(self.seed)
        if self.replay_buffer_class is None:
            self.replay_buffer_class = ReplayBuffer
        if self.replay_buffer is None:
            kw = dict(self.replay_buffer_kwargs)
            self.replay_buffer = self.replay_buffer_class(self.buffer_size, self.observation_space, self.action_space,
                                                          device=self.device, n_envs=self.n_envs,
                                                          optimize_memory_usage=self.optimize_memory_usage, **kw)
        # built on the CPU generator in the reference's construction order (same seed -> same initial weights),
        # then moved into the HBM arenas by the policy itself
        self.policy = self.policy_class(self.observation_space, self.action_space, self.lr_schedule, **self.policy_kwargs)
        self.policy.to_device_arenas(self.device)
        if self.world_size > 1:
            self.policy.broadcast_from_rank0()
            for opt in self.policy.flat_optimizers():
                opt.grad_scale = 1.0 / self.world_size
            if self.seed is not None:  # same init everywhere, different exploration noise per shard
                th.manual_seed(self.seed + 1000003 * self.rank)
        self._convert_train_freq()
        n = self.n_envs
        self._ep_return = th.zeros(n, dtype=th.float32, device=self.device)
        self._ep_stats = th.zeros(4, dtype=th.float64, device=self.device)

    def save_replay_buffer(self, path) -> None:
        """reference: off_policy_algorithm.py:214-222"""
        from core.common.save_util import save_to_pkl

        assert self.replay_buffer is not None, "The replay buffer is not defined"
        save_to_pkl(path, self.replay_buffer, self.verbose)

    def load_replay_buffer(self, path, truncate_last_traj: bool = True) -> None:
        """reference: off_policy_algorithm.py:224-254 (HerReplayBuffer is out of scope)"""
        from core.common.save_util import load_from_pkl

        self.replay_buffer = load_from_pkl(path, self.verbose)
        assert isinstance(self.replay_buffer, ReplayBuffer), "The replay buffer must inherit from ReplayBuffer class"
        self.replay_buffer.to(self.device)  # :252-253
        self.replay_buffer.normalizer = self._vec_normalize_env
        self._graph = None  # captured graphs hold the old ring's pointers

    def _fast_path(self) -> bool:
        rb = self.replay_buffer
        env = self._denv
        return (env is not None and type(rb) is ReplayBuffer and rb.n_envs == env.num_envs
                and rb.obs_shape[0] == env.obs_dim and rb.action_dim == env.act_dim)

    # ---- learn ----------------------------------------------------------------------------------------------------
    def _setup_learn(self, total_timesteps, callback=None, reset_num_timesteps=True, tb_log_name="run", progress_bar=False):
        from core.common.noise import (DeviceNormalActionNoise, LegacyStreamNormalActionNoise, LegacyStreamOUActionNoise,
                                       NormalActionNoise, OrnsteinUhlenbeckActionNoise, VectorizedActionNoise)

        self.replay_buffer.normalizer = self._vec_normalize_env  # sample(..., env=self._vec_normalize_env), sac.py:215
        opt = getattr(getattr(self.policy, "actor", None), "optimizer", None)
        if hasattr(opt, "refresh_shadow"):  # weights may have been loaded / set since the last learn(): graphs replay, Python does not
            opt.refresh_shadow()
        base = self.action_noise
        if isinstance(base, VectorizedActionNoise) and base.n_envs == self.env.num_envs:
            base = base.base_noise
        if isinstance(base, NormalActionNoise) and self._fast_path() and np.size(base._mu) <= 8:
            # the reference's n_envs sequential np.random.normal draws per vec-step (noise.py:44-45, :141-142) as one
            # kernel on the HBM image of the same legacy stream the replay sampler uses: bit-faithful interleaving
            self.action_noise = LegacyStreamNormalActionNoise(base._mu, base._sigma, self.env.num_envs, self.device,
                                                              lambda: self.replay_buffer.sampler_stream)
        elif isinstance(base, OrnsteinUhlenbeckActionNoise) and self._fast_path() and np.size(base._mu) <= 8:
            # noise.py:84-89 per env on the same stream, float64 state in HBM
            self.action_noise = LegacyStreamOUActionNoise(base._mu, base._sigma, base._theta, base._dt, base.initial_noise,
                                                          self.env.num_envs, self.device, lambda: self.replay_buffer.sampler_stream)
        elif self.action_noise is not None and self.env.num_envs > 1 and not hasattr(self.action_noise, "noises") \
                and not isinstance(self.action_noise, (DeviceNormalActionNoise, LegacyStreamNormalActionNoise, LegacyStreamOUActionNoise)):
            self.action_noise = VectorizedActionNoise(self.action_noise, self.env.num_envs)
        return super()._setup_learn(total_timesteps, callback, reset_num_timesteps, tb_log_name, progress_bar)

    def learn(self, total_timesteps: int, callback: MaybeCallback = None, log_interval: int = 4, tb_log_name: str = "run",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        """reference: off_policy_algorithm.py:309-355"""
        total_timesteps, callback = self._setup_learn(total_timesteps, callback, reset_num_timesteps, tb_log_name, progress_bar)
        callback.on_training_start(locals(), globals())
        assert self.env is not None, "You must set the environment before calling learn()"
        assert isinstance(self.train_freq, TrainFreq)
        while self.num_timesteps < total_timesteps:
            if not self._learn_iteration(callback, log_interval):
                break
        if self._fast_path():
            self._sync_episode_stats(log_interval, force=True)
        callback.on_training_end()
        return self

    def _learn_iteration(self, callback: BaseCallback, log_interval: Optional[int]) -> bool:
        """Body of the reference's `while` loop (off_policy_algorithm.py:331-351): one rollout, then train.
        When the iteration is eligible it is replayed from a captured hipGraph instead (same launches, same
        order, one host call)."""
        if self._graph_enabled and self._graph_eligible(callback):
            self._graph_iteration(log_interval, callback)
            return True
        self._eager_iterations += 1
        rollout = self.collect_rollouts(self.env, train_freq=self.train_freq, action_noise=self.action_noise,
                                        callback=callback, learning_starts=self.learning_starts,
                                        replay_buffer=self.replay_buffer, log_interval=log_interval)
        if not rollout.continue_training:
            return False
        if self.num_timesteps > 0 and self.num_timesteps > self.learning_starts:
            gradient_steps = self.gradient_steps if self.gradient_steps >= 0 else rollout.episode_timesteps
            if gradient_steps > 0:
                self.train(batch_size=self.batch_size, gradient_steps=gradient_steps)
        return True

    # ---- hipGraph capture of the steady-state iteration ------------------------------------------------------------
    def enable_graph_capture(self, enabled: bool = True, unroll: Optional[int] = None) -> None:
        """Replay the steady-state iteration (actor forward, fused collect, `gradient_steps` gradient steps) from a
        captured hipGraph: ~250 launches become one host call. Every per-call control word the kernels need (ring
        position, Adam step, MT19937 stream, learning rate, env / RNG state) lives in HBM, so a replay is exact.
        Falls back to the eager path whenever the iteration is not capturable (warm-up, callbacks, host-side action noise,
        episodic train_freq).

        `unroll` (default 1, env CSTR_GRAPH_UNROLL): consecutive iterations recorded into ONE graph -- the ~10 us the GPU
        idles between two graph launches is paid once per `unroll` iterations. Used on one GPU, and data-parallel when the
        all-reduces are recorded into the graph (every rank replays the same graphs in the same order), with a constant learning
        rate while at least `unroll` iterations remain; the tail of a run replays graphs of unroll / 2, unroll / 4, ... 1 iterations."""
        self._graph_enabled = enabled
        self._graph, self._graph_error, self._abi_launches = None, None, {}
        self.graph_unroll = max(1, int(unroll if unroll is not None else os.environ.get("CSTR_GRAPH_UNROLL", "1")))

    def _graph_eligible(self, callback: BaseCallback) -> bool:
        from core.common.noise import DeviceNormalActionNoise, LegacyStreamNormalActionNoise, LegacyStreamOUActionNoise

        return (self._fast_path() and getattr(callback, "is_noop", False)
                and (self.action_noise is None
                     or isinstance(self.action_noise, (DeviceNormalActionNoise, LegacyStreamNormalActionNoise, LegacyStreamOUActionNoise)))
                and self.train_freq == TrainFreq(1, TrainFrequencyUnit.STEP)
                and self.gradient_steps >= 1 and self.num_timesteps >= self.learning_starts
                and self.num_timesteps + self.n_envs > self.learning_starts and not getattr(self, "debug_capture", False))

    def _graph_body(self) -> None:
        env, rb, vn = self._denv, self.replay_buffer, self._vec_normalize_env
        self.policy.set_training_mode(False)
        noise = None if self.action_noise is None else self.action_noise().contiguous()
        net = self._rollout_net() if (FUSED_ROLLOUT and vn is None and self._use_packed_batch()) else None
        if net is not None:
            # policy network + sampling + collect step + the first gradient step's replay index draw in ONE launch; the gather
            # launch of that gradient step advances the ring position and the policy's Philox offset (hip_ops.rollout_step)
            idx = rb.predraw_indices(self.batch_size)
            hip_ops.rollout_step(env.obs, *net["weights"], net["act"], net["head"], net["out_act"], net["w2_swz"], net["rng_ctl"], env.coef,
                                 env.integrator, rb.ring, env.obs, env.step_count, self._action_mode(False), self.action_space.low,
                                 self.action_space.high, noise=noise, pcg_state=env.pcg_state, static_init=env.static_init,
                                 reward_out=env._rew, done_out=env._done, ep_return=self._ep_return, ep_stats=self._ep_stats,
                                 mt_state=rb.sampler_stream, sample_idx=idx)
            rb.note_predrawn(idx, None if net["rng_ctl"] is None else (net["rng_ctl"], env.num_envs))
            if hasattr(self.action_noise, "reset_done"):
                self.action_noise.reset_done(env._done)
            self.policy.set_training_mode(True)
            self._train_device_only(self.gradient_steps, self.batch_size)
            rb._no_predrawn("the iteration's first gradient step")
            return
        pol = self._policy_out_device(env.obs if vn is None else vn.norm_obs_dev)
        hip_ops.collect_step(env.coef, env.integrator, rb.ring, env.obs, env.step_count, pol, self._action_mode(False),
                             self.action_space.low, self.action_space.high, noise=noise, pcg_state=env.pcg_state, static_init=env.static_init, reward_out=env._rew, done_out=env._done,
                             ep_return=self._ep_return, ep_stats=self._ep_stats, rng_advance=self._take_rng_advance())
        if hasattr(self.action_noise, "reset_done"):
            self.action_noise.reset_done(env._done)  # action_noise.reset(indices of finished envs), :596-599
        if vn is not None:
            vn.after_device_step()
        self.policy.set_training_mode(True)
        self._train_device_only(self.gradient_steps, self.batch_size)

    def _rollout_net(self) -> Optional[dict]:
        """The rollout policy as operands of hip_ops.rollout_step (weights = (w1, b1, w2, b2, w3, b3), act, head, out_act, w2_swz,
        rng_ctl), or None when the one-launch rollout does not cover this algorithm / network (the separate launches run then)."""
        return None

    def _use_packed_batch(self) -> bool:
        return False

    def _graph_host_bookkeeping(self, log_interval: Optional[int]) -> None:
        self.replay_buffer.note_fused_add()
        self._last_obs = self._denv.obs if self._vec_normalize_env is None else self._vec_normalize_env.norm_obs_dev
        self.num_timesteps += self.n_envs
        self._update_current_progress_remaining(self.num_timesteps, self._total_timesteps)
        self._train_host_only(self.gradient_steps)
        self._sync_episode_stats(log_interval)

    def _graph_phase(self) -> int:
        """Iterations that launch different kernel sequences need different graphs (TD3 / MADDPG: the delayed policy
        update happens every `policy_delay`-th gradient step)."""
        return 0

    def _graph_unroll_now(self) -> int:
        u = getattr(self, "graph_unroll", 1)
        if u <= 1 or not isinstance(self.learning_rate, float):
            return 1
        if (self.world_size > 1 or getattr(self, "_force_segment_boundaries", False)) and not self._collectives_in_graph():
            return 1  # data-parallel with the collectives BETWEEN graph segments: one iteration per replay list
        remaining = (self._total_timesteps - self.num_timesteps) // self.n_envs
        while u > 1 and remaining < u:  # the tail of a run: the largest of u, u / 2, u / 4, ... that still fits
            u //= 2
        return max(u, 1)

    def _graph_iteration(self, log_interval: Optional[int], callback: Optional[BaseCallback] = None) -> None:
        vn = self._vec_normalize_env
        opt = getattr(getattr(self.policy, "actor", None), "optimizer", None)
        if getattr(opt, "shadow", None) is not None:  # torch changed the actor's weights (a callback, load_state_dict): the
            opt.refresh_shadow(force=False)           # replayed graph reads their tile-major copy -- one version compare
        unroll = self._graph_unroll_now()
        key = (id(self._denv.coef), self.batch_size, self.gradient_steps, self._graph_phase(), None if vn is None else (id(vn), vn.cfg_key),
               unroll)
        if not isinstance(self._graph, dict):
            self._graph, self._graph_warm = {}, {}
        if key not in self._graph:
            # side-stream warm-up (these are REAL iterations: they advance env, ring, RNG and optimiser state)
            warm = self._graph_warm.get(key, 0)
            if warm < 3:
                self._train_host_pre()
                side = th.cuda.Stream(device=self.device)
                side.wait_stream(th.cuda.current_stream(self.device))
                with th.cuda.stream(side):
                    self._graph_body()
                th.cuda.current_stream(self.device).wait_stream(side)
                self._graph_warm[key] = warm + 1
                self._eager_iterations += 1
                self._graph_host_bookkeeping(log_interval)
                return
            self._train_host_pre()
            try:
                self._graph[key] = self._capture_segments(unroll)
            except Exception as exc:  # something in the iteration is not capturable: run eagerly from now on
                import warnings

                self._graph_error = f"{type(exc).__name__}: {exc}"
                warnings.warn(f"hipGraph capture failed ({self._graph_error}); falling back to eager launches")
                self._graph_enabled, self._graph = False, None
                self._learn_iteration(callback if callback is not None else self._noop_callback(), log_interval)
                return
        self._train_host_pre()
        for item in self._graph[key]:  # hipGraph segments interleaved with the eager collectives that separate them
            item.replay() if isinstance(item, th.cuda.CUDAGraph) else item()
        self._graph_replays += unroll
        for _ in range(unroll):
            self._graph_host_bookkeeping(log_interval)

    def _noop_callback(self) -> BaseCallback:
        cb = to_callback(None)
        cb.init_callback(self)
        return cb

    def graph_status(self) -> dict:
        """What actually runs (not what was requested): bench.py refuses to report a run whose graphs fell back to eager."""
        graphs = self._graph if isinstance(self._graph, dict) else {}
        segs = [sum(isinstance(i, th.cuda.CUDAGraph) for i in items) for items in graphs.values()]
        mode = "none"
        if self.world_size > 1 or getattr(self, "_force_segment_boundaries", False):
            mode = "in-graph" if getattr(self, "_graph_collectives", False) else "segmented"
        return dict(requested=bool(self._graph_enabled or self._graph_error), active=bool(self._graph_enabled and len(graphs) > 0),
                    graphs=len(graphs), segments_per_graph=segs, replays=self._graph_replays, eager_iterations=self._eager_iterations,
                    error=self._graph_error, graph_collectives=mode,
                    abi_launches_per_iteration={int(k): v for k, v in sorted(getattr(self, "_abi_launches", {}).items())})

    def _capture_segments(self, unroll: int = 1) -> list:
        """`_record_segments`, and if recording WITH the collectives inside the graph raises (every rank runs the same code,
        so every rank gets here), once more with the collectives between graph segments."""
        try:
            return self._record_segments(unroll)
        except Exception as exc:  # noqa: BLE001
            if not (self.world_size > 1 and getattr(self, "_graph_collectives", False) and GRAPH_COLLECTIVES == "auto"):
                raise
            print(f"[graph] recording the collectives into the graph failed ({exc!r}); keeping them between graph segments", file=sys.stderr)
            self._graph_collectives = False
            th.cuda.synchronize(self.device)
            return self._record_segments(unroll)

    def _record_segments(self, unroll: int = 1) -> list:
        """Record the iteration as hipGraph segments. A data-parallel run has an RCCL all-reduce between backward and
        the optimiser step (two per SAC gradient step). When the start-up trial passes (`_collectives_in_graph`) they are
        recorded into the graph; otherwise collectives stay OUTSIDE the captured graphs -- every `_eager_boundary` closes the
        current segment, runs the collective eagerly and opens the next segment in the same memory pool (activations saved
        for a later segment's backward stay alive). Single-GPU runs have no boundary and get one graph."""
        import gc

        # like torch.cuda.graph(): collect garbage BEFORE recording and keep the collector off while recording -- a cycle
        # collection that destroys another model's CUDAGraph (or frees device memory) in the middle of a capture aborts
        self._collectives_in_graph()  # decided (start-up trial, world > 1) before anything is being recorded
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        th.cuda.synchronize(self.device)
        side = th.cuda.Stream(device=self.device)
        side.wait_stream(th.cuda.current_stream(self.device))
        items: list = []
        with th.cuda.stream(side):
            self._cap = dict(pool=th.cuda.graph_pool_handle(), graph=th.cuda.CUDAGraph(), items=items)
            self._cap["graph"].capture_begin(pool=self._cap["pool"], capture_error_mode="thread_local")
            n_updates = self._n_updates
            calls0 = nv.ABI_CALLS[0]
            try:
                for _ in range(unroll):
                    self._graph_body()
                    self._n_updates += self.gradient_steps  # the next body sees its own policy-delay phase
                self._cap["graph"].capture_end()
                items.append(self._cap["graph"])
                # launches recorded per iteration of this policy-delay phase (every launch of the captured body goes through the
                # C ABI; bench.py reports it, tools/count_launches.sh is the rocprofv3 cross-check)
                self._abi_launches[self._graph_phase()] = (nv.ABI_CALLS[0] - calls0) / unroll
            except Exception:
                try:  # leave capture mode before the graph object is destroyed
                    self._cap["graph"].capture_end()
                except Exception:
                    pass
                # nothing of the recorded body ran: host-side debts of the one-launch rollout (indices "drawn" by a launch that was
                # only recorded, a Philox advance handed to a consumer that was never reached) must not reach the eager fallback
                self._drop_recording_debts()
                raise
            finally:
                self._cap = None
                self._n_updates = n_updates  # nothing ran while recording
                if gc_was_enabled:
                    gc.enable()
        th.cuda.current_stream(self.device).wait_stream(side)
        th.cuda.synchronize(self.device)
        return items

    def _drop_recording_debts(self) -> None:
        rb = getattr(self, "replay_buffer", None)
        if rb is not None and hasattr(rb, "_predrawn"):
            rb._predrawn = None
        self._rng_advance = None

    def _collectives_in_graph(self) -> bool:
        if getattr(self, "_graph_collectives", None) is None:
            if GRAPH_COLLECTIVES in ("0", "1"):
                self._graph_collectives = GRAPH_COLLECTIVES == "1"
            else:
                self._graph_collectives = self.world_size > 1 and dist_util.graph_collectives_ok(self.device)
        return self._graph_collectives

    def _eager_boundary(self, fn) -> None:
        """Run `fn` (a collective) eagerly; when a capture is in progress, split the graph around it."""
        cap = getattr(self, "_cap", None)
        if cap is None or self._collectives_in_graph():
            # no capture in progress, or the collective is recorded into the graph like any other launch (RCCL issues a
            # blocking collective on the current stream) and the iteration stays ONE graph
            fn()
            return
        cap["graph"].capture_end()
        cap["items"].append(cap["graph"])
        fn()
        cap["items"].append(fn)
        cap["graph"] = th.cuda.CUDAGraph()
        cap["graph"].capture_begin(pool=cap["pool"], capture_error_mode="thread_local")

    # ---- action selection -----------------------------------------------------------------------------------------
    def _action_mode(self, warmup: bool) -> int:
        """`squashed` bit field of cstr_collect_step_f32: bit 0 = the input is the actor's tanh output (predict()
        unscales it first); bit 1 = multi-agent behaviour (no scale/unscale round trip, no noise)."""
        return 0 if warmup else 1

    def _take_rng_advance(self):
        adv, self._rng_advance = self._rng_advance, None
        return adv

    def _policy_out_device(self, obs: th.Tensor) -> th.Tensor:
        """Actor output for the fused collect kernel: squashed ([-1,1]) action, device tensor [N, A], no grad."""
        with th.no_grad():
            return self.policy._predict(obs, deterministic=False).contiguous()

    def _sample_action(self, learning_starts: int, action_noise=None, n_envs: int = 1):
        """reference: off_policy_algorithm.py:364-411 (NumPy compatibility path)"""
        if self.num_timesteps < learning_starts:
            unscaled_action = self.action_space.sample_batch(n_envs)
        else:
            assert self._last_obs is not None, "self._last_obs was not set"
            unscaled_action, _ = self.predict(self._last_obs, deterministic=False)
        scaled_action = self.policy.scale_action(unscaled_action)
        if action_noise is not None:
            scaled_action = np.clip(scaled_action + action_noise(), -1, 1)
        buffer_action = scaled_action
        action = self.policy.unscale_action(scaled_action)
        return action, buffer_action

    # ---- logging ---------------------------------------------------------------------------------------------------
    def _dump_logs(self) -> None:
        """reference: off_policy_algorithm.py:413-438"""
        time_elapsed = max((time.time_ns() - self.start_time) / 1e9, sys.float_info.epsilon)
        fps = int((self.num_timesteps - self._num_timesteps_at_start) / time_elapsed)
        if self.world_size > 1:
            fps *= self.world_size  # whole-job env-steps/s: every rank advances n_envs per vec-step
        self.logger.record("time/episodes", self._episode_num, exclude="tensorboard")
        n_ep, ret_sum, len_sum = self._ep_window
        if n_ep > 0:
            self.logger.record("rollout/ep_rew_mean", ret_sum / n_ep)
            self.logger.record("rollout/ep_len_mean", len_sum / n_ep)
        self.logger.record("time/fps", fps)
        self.logger.record("time/time_elapsed", int(time_elapsed), exclude="tensorboard")
        self.logger.record("time/total_timesteps", self.num_timesteps, exclude="tensorboard")
        self.logger.dump(step=self.num_timesteps)

    def _sync_episode_stats(self, log_interval: Optional[int], force: bool = False) -> None:
        """One blocking read of four doubles (episodes, sum of returns, sum of lengths). The reference learns about
        finished episodes from the host-side `dones` every vec-step (:590-602); here the counters live in HBM."""
        self._steps_since_sync += 1
        if not force and self._steps_since_sync < self.stats_sync_interval:
            return
        self._steps_since_sync = 0
        n_ep, ret_sum, len_sum, _ = self._ep_stats.cpu().tolist()
        self._episode_num = int(n_ep)
        done_since = self._episode_num - self._episodes_at_last_dump
        if log_interval is not None and done_since >= log_interval:
            w0, w1, w2 = getattr(self, "_ep_totals_at_dump", (0.0, 0.0, 0.0))
            self._ep_window = (n_ep - w0, ret_sum - w1, len_sum - w2)
            self._ep_totals_at_dump = (n_ep, ret_sum, len_sum)
            self._episodes_at_last_dump = self._episode_num
            self._dump_logs()

    def _on_step(self) -> None:
        pass

    # ---- storage (compatibility path) ------------------------------------------------------------------------------
    def _store_transition(self, replay_buffer, buffer_action, new_obs, reward, dones, infos) -> None:
        """reference: off_policy_algorithm.py:445-508"""
        next_obs = np.array(new_obs, copy=True)
        for i, done in enumerate(dones):
            if done and infos[i].get("terminal_observation") is not None:
                next_obs[i] = infos[i]["terminal_observation"]
        replay_buffer.add(self._last_obs, next_obs, buffer_action, reward, dones, infos)
        self._last_obs = new_obs

    # ---- rollouts --------------------------------------------------------------------------------------------------
    def collect_rollouts(self, env: VecEnv, callback: BaseCallback, train_freq: TrainFreq, replay_buffer: ReplayBuffer,
                         action_noise=None, learning_starts: int = 0, log_interval: Optional[int] = None) -> RolloutReturn:
        """reference: off_policy_algorithm.py:510-605"""
        self.policy.set_training_mode(False)
        num_collected_steps, num_collected_episodes = 0, 0
        assert train_freq.frequency > 0, "Should at least collect one step or episode."
        if env.num_envs > 1:
            assert train_freq.unit == TrainFrequencyUnit.STEP, "You must use only one env when doing episodic training."
        fast = self._fast_path() and train_freq.unit == TrainFrequencyUnit.STEP
        noop_cb = getattr(callback, "is_noop", False)
        callback.on_rollout_start()
        continue_training = True
        while should_collect_more_steps(train_freq, num_collected_steps, num_collected_episodes):
            if fast:
                self._collect_one_fused(env.unwrapped, replay_buffer, action_noise, learning_starts)
                new_obs, rewards, dones = self._last_obs, env._rew, env._done  # device tensors (for callbacks)
                infos: Any = None
            else:
                if isinstance(self._last_obs, th.Tensor):
                    self._last_obs = self._last_obs.cpu().numpy()
                actions, buffer_actions = self._sample_action(learning_starts, action_noise, env.num_envs)
                new_obs, rewards, dones, infos = env.step(actions)
            self.num_timesteps += env.num_envs
            num_collected_steps += 1
            if not noop_cb:
                callback.update_locals(locals())
            if not callback.on_step():
                return RolloutReturn(num_collected_steps * env.num_envs, num_collected_episodes, continue_training=False)
            if not fast:
                self._store_transition(replay_buffer, buffer_actions, new_obs, rewards, dones, infos)
            self._update_current_progress_remaining(self.num_timesteps, self._total_timesteps)
            self._on_step()
            if fast:
                self._sync_episode_stats(log_interval)
            else:
                for idx, done in enumerate(dones):
                    if done:
                        num_collected_episodes += 1
                        self._episode_num += 1
                        if action_noise is not None:
                            kwargs = dict(indices=[idx]) if env.num_envs > 1 else {}
                            action_noise.reset(**kwargs)
                        if log_interval is not None and self._episode_num % log_interval == 0:
                            self._dump_logs()
        callback.on_rollout_end()
        return RolloutReturn(num_collected_steps * env.num_envs, num_collected_episodes, continue_training)

    def _collect_one_fused(self, env: CSTRVecEnv, rb: ReplayBuffer, action_noise, learning_starts: int) -> None:
        """One vec-step entirely in HBM: reference statements :561 (_sample_action), :564 (env.step), :580
        (_store_transition -> ReplayBuffer.add) in one HIP launch after the actor forward."""
        n, vn = env.num_envs, self._vec_normalize_env
        if self.num_timesteps < learning_starts:
            # warm-up: uniform actions from the action space's own generator (:386-388); drawn on the host
            pol = th.as_tensor(self.action_space.sample_batch(n)).to(self.device)
            squashed = self._action_mode(warmup=True)
        else:
            pol = self._policy_out_device(env.obs if vn is None else vn.norm_obs_dev)
            squashed = self._action_mode(warmup=False)
        noise = None
        if action_noise is not None:
            z = action_noise()
            noise = z if isinstance(z, th.Tensor) else th.as_tensor(np.asarray(z, np.float32))
            noise = noise.to(self.device, th.float32).reshape(n, -1).contiguous()
        with th.cuda.device(self.device):
            hip_ops.collect_step(env.coef, env.integrator, rb.ring, env.obs, env.step_count, pol, squashed,
                                 self.action_space.low, self.action_space.high, noise=noise, pcg_state=env.pcg_state, static_init=env.static_init,
                                 reward_out=env._rew, done_out=env._done, ep_return=self._ep_return, ep_stats=self._ep_stats,
                                 rng_advance=self._take_rng_advance())
            if hasattr(action_noise, "reset_done"):
                action_noise.reset_done(env._done)  # action_noise.reset(indices of finished envs), :596-599
            if vn is not None:
                vn.after_device_step()  # VecNormalize.step_wait on the raw outputs (vec_normalize.py:174-204)
        rb.note_fused_add()
        self._last_obs = env.obs if vn is None else vn.norm_obs_dev

    # ---- data-parallel helper used by train() ----------------------------------------------------------------------
    def _allreduce_grads(self, arena) -> None:
        if self.world_size > 1 or getattr(self, "_force_segment_boundaries", False):
            buf = getattr(arena, "grad_full", None)
            buf = arena.grad if buf is None else buf
            self._eager_boundary(lambda: dist_util.allreduce_sum_(buf))
