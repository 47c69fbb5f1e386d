"""VecEnv protocol (reference: core/common/vec_env/base_vec_env.py:50-335) and the device-resident
`CSTRVecEnv` that replaces `DummyVecEnv([lambda: TwoSeriesCSTREnv()] * N)`: N reactor trains live in HBM
and one HIP launch advances all of them (reference loop: core/common/vec_env/dummy_vec_env.py:56-73).

Two faces:
  * compatibility path  -- `reset()/step(actions)` take and return NumPy exactly like the reference
    (copies, bool dones, infos with "TimeLimit.truncated"/"terminal_observation");
  * fast path           -- `step_device(actions)` and the fused `collect_step` used by the off-policy
    loop keep everything in HBM and never synchronise with the host.
"""
from typing import Optional, Sequence

import numpy as np
import torch as th

from core import _native as nv
from core.common import hip_ops
from core.common.spaces import Box
from core.common.vec_env.base_vec_env import VecEnv


class CSTRVecEnv(VecEnv):
    """N independent two-series CSTR environments (reference: twoseriescstr.py:15-503) stepped by one HIP kernel.

    :param num_envs: number of reactor trains held by this process / GPU
    :param obs_dim: 4 = the reference's observation [C1,T1,C2,T2] normalised to [-1,1];
                    8 = [normalised | raw] (both halves are emitted by the reference's `info`)
    :param twin: with obs_dim=8: TWO reactor trains side by side per env -- observation [train A | train B] (both
                    normalised), 4 actions [F1A,F2A,F1B,F2B], reward rA + rB, one step counter / reset stream per env.
                    The 8-obs/4-act environment a 4-agent MADDPG needs (constructed, not in the reference)
    :param integrator: "euler" (reference, parity-pinned) or "rk4" (north_star's ask; not in the reference)
    :param default_target, min_concentration, max_concentration, init_mode: TwoSeriesCSTREnv ctor args
    :param seed_offset: added to env seeds, used by data-parallel shards (rank * num_envs)
    """

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}
    max_steps = 400

    def __init__(self, num_envs: int, obs_dim: int = 4, integrator: str = "euler", device="cuda",
                 default_target: float = 0.20, min_concentration: float = 0.05, max_concentration: float = 0.45,
                 init_mode: str = "random", seed_offset: int = 0, twin: bool = False):
        if obs_dim not in (4, 8):
            raise ValueError(f"obs_dim must be 4 or 8, got {obs_dim}")
        if twin and obs_dim != 8:
            raise ValueError("twin=True needs obs_dim=8 (two 4-dim reactor trains per env)")
        if integrator not in nv.INTEGRATORS:
            raise ValueError(f"integrator must be one of {list(nv.INTEGRATORS)}, got {integrator!r}")
        if init_mode not in ("random", "static"):
            raise ValueError(f"init_mode={init_mode} is not supported, please choose 'random' or 'static'")  # twoseriescstr.py:257
        from core.common.utils import get_device

        self.device = get_device(device)
        nv.lib()  # fail loudly if the HIP extension is missing
        one = np.ones(obs_dim, np.float32)
        self.twin, self.act_dim = twin, (4 if twin else 2)
        if obs_dim == 8 and not twin:
            lo = np.concatenate([-one[:4], np.array([0.0, 273.15, 0.0, 273.15], np.float32)])
            hi = np.concatenate([one[:4], np.array([0.7, 400.0, 0.7, 400.0], np.float32)])
        else:
            lo, hi = -one, one
        a1 = np.ones(self.act_dim, np.float32)
        super().__init__(num_envs, Box(lo, hi, dtype=np.float32), Box(-a1, a1, dtype=np.float32))
        self.obs_dim, self.integrator, self.seed_offset = obs_dim, integrator, seed_offset
        self.target_C2, self.min_concentration, self.max_concentration = default_target, min_concentration, max_concentration
        self.init_mode = init_mode
        self.coef = nv.default_coef(default_target, min_concentration, max_concentration, self.max_steps)
        n, dev = num_envs, self.device
        with th.cuda.device(dev):
            self.obs = th.zeros(n, obs_dim, dtype=th.float32, device=dev)          # TwoSeriesCSTREnv.state (+ raw half)
            self.step_count = th.zeros(n, dtype=th.int32, device=dev)              # TwoSeriesCSTREnv.current_step
            self.pcg_state = th.zeros(n, nv.PCG_STATE_WORDS, dtype=th.int64, device=dev)  # per-env np_random (PCG64)
            self._next_obs = th.zeros(n, obs_dim, dtype=th.float32, device=dev)
            self._reset_buf = th.zeros(n, obs_dim, dtype=th.float32, device=dev)
            self._rew, self._done, self._timeout = (th.zeros(n, dtype=th.float32, device=dev) for _ in range(3))
            # init_mode="static": every env's f64 `init_state`, which each reset perturbs in place (twoseriescstr.py:94-96,
            # :246-255); None = init_mode="random"
            self.static_init = None
            if init_mode == "static":
                base = th.tensor([0.45, 310.0, 0.25, 290.0], dtype=th.float64, device=dev)
                self.static_init = base.repeat(n, 2 if twin else 1).contiguous()
        self._rng_seeded = False
        self._pending_actions: Optional[th.Tensor] = None
        self.numpy_reseed: Optional[int] = None  # last `np.random.seed` a seeded reset performed (twoseriescstr.py:164)
        self._has_reset = False

    # ---- seeding / reset -----------------------------------------------------------------------------
    def set_target(self, target: float) -> bool:
        """reference: twoseriescstr.py:114-127"""
        if self.min_concentration <= target <= self.max_concentration:
            self.target_C2 = target
            self.coef = nv.default_coef(target, self.min_concentration, self.max_concentration, self.max_steps)
            return True
        return False

    def _seed_rng(self, seeds: Sequence[Optional[int]]) -> None:
        """Per-env `seeding.np_random(seed)` = Generator(PCG64(SeedSequence(seed))) (twoseriescstr.py:162). The
        SeedSequence hashing runs in NumPy on the host (once); only the 4 state words go to the GPU."""
        m = (1 << 64) - 1
        words = np.empty((self.num_envs, 4), np.uint64)
        for i, s in enumerate(seeds):
            st = np.random.PCG64(np.random.SeedSequence(None if s is None else int(s))).state["state"]
            words[i] = (st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m)
        self.pcg_state.copy_(th.from_numpy(words.view(np.int64)))
        self._rng_seeded = True

    def reset_device(self) -> th.Tensor:
        """reference: dummy_vec_env.py:75-83 + TwoSeriesCSTREnv.reset (twoseriescstr.py:226-269)."""
        seeds = list(self._seeds)
        if any(s is not None for s in seeds):
            # a seeded reset re-creates every seeded env's generator and reseeds the GLOBAL numpy stream with the
            # env's seed (twoseriescstr.py:163-164): the last env wins -> seed + N - 1 (SURVEY a-6)
            if not all(s is not None for s in seeds):
                raise ValueError("CSTRVecEnv.reset: either all or none of the envs must be seeded")
            self._seed_rng([s + self.seed_offset for s in seeds])
            self.numpy_reseed = int(seeds[-1] + self.seed_offset) & 0xFFFFFFFF
        elif not self._rng_seeded:
            self._seed_rng([None] * self.num_envs)  # unseeded envs draw OS entropy, like gymnasium does
        with th.cuda.device(self.device):
            hip_ops.reset_draw(self.pcg_state, None, self.obs, self.act_dim, static_init=self.static_init)
            self.step_count.zero_()
        self._reset_seeds()
        self._reset_options()
        self._has_reset = True
        return self.obs

    def reset(self) -> np.ndarray:
        return self.reset_device().cpu().numpy()

    def set_state(self, obs, step_count=None) -> None:
        """Inject observations / step counters (tests, fixtures). obs: [N, 4] normalised or [N, obs_dim]."""
        obs = th.as_tensor(np.asarray(obs, np.float32))
        if obs.shape == (self.num_envs, 4) and self.obs_dim == 8 and not self.twin:
            lo = th.tensor([0.0, 273.15, 0.0, 273.15])
            hi = th.tensor([0.7, 400.0, 0.7, 400.0])
            raw = th.minimum(th.maximum(lo + (obs + 1.0) * (hi - lo) / 2.0, lo), hi)
            obs = th.cat([obs, raw], dim=1)
        if tuple(obs.shape) != (self.num_envs, self.obs_dim):
            raise ValueError(f"obs shape {tuple(obs.shape)} != {(self.num_envs, self.obs_dim)}")
        self.obs.copy_(obs)
        if step_count is not None:
            self.step_count.copy_(th.as_tensor(np.asarray(step_count, np.int32)))
        if not self._rng_seeded:
            self._seed_rng([None] * self.num_envs)
        self._has_reset = True

    # ---- stepping ------------------------------------------------------------------------------------
    def step_device(self, actions: th.Tensor):
        """One launch for all envs, no host sync. Returns device views
        (obs_after, reward, done, timeout, next_obs): `next_obs` is the terminal observation where done."""
        if not self._has_reset:
            raise ValueError("Please call env.reset() to reset the env first!")  # twoseriescstr.py:401-402
        with th.cuda.device(self.device):
            # reset source for envs that finish now: draw lazily for ALL envs would advance every stream, so the
            # draw kernel runs masked AFTER the step on a scratch copy of the observations
            hip_ops.vec_step(self.coef, self.integrator, self.obs, actions, self.step_count, self.obs, self._next_obs,
                             self._reset_buf, self._rew, self._done, self._timeout)
            # _reset_buf now holds obs_after with the OLD obs where done; overwrite those rows with fresh draws
            hip_ops.reset_draw(self.pcg_state, self._done.to(th.uint8), self._reset_buf, self.act_dim, static_init=self.static_init)
            self.obs.copy_(self._reset_buf)
        return self.obs, self._rew, self._done, self._timeout, self._next_obs

    def step_async(self, actions) -> None:
        a = th.as_tensor(np.asarray(actions, dtype=np.float32)) if not isinstance(actions, th.Tensor) else actions
        if tuple(a.shape) != (self.num_envs, self.act_dim):
            raise ValueError(f"actions shape {tuple(a.shape)} != {(self.num_envs, self.act_dim)}")
        self._pending_actions = a.to(self.device, th.float32).contiguous()

    def step_wait(self):
        obs, rew, done, timeout, next_obs = self.step_device(self._pending_actions)
        obs_h, rew_h = obs.cpu().numpy(), rew.cpu().numpy()
        done_h, tout_h = done.cpu().numpy().astype(bool), timeout.cpu().numpy().astype(bool)
        infos: list = [{"TimeLimit.truncated": bool(t)} for t in tout_h]
        if done_h.any():
            term = next_obs.cpu().numpy()
            for i in np.nonzero(done_h)[0]:
                infos[i]["terminal_observation"] = term[i].copy()
        return obs_h, rew_h, done_h, infos
