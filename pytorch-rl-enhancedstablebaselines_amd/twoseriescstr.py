"""`TwoSeriesCSTREnv` with the reference's constructor and gym-style API (reference: twoseriescstr.py:15-503),
backed by the HIP environment kernel. A single instance is a 1-env view of `CSTRVecEnv`; building a
`DummyVecEnv` out of N instances collapses them into ONE batched device environment (core/common/vec_env.py),
which is what the off-policy algorithms step.
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np

from core.common.spaces import Box


class TwoSeriesCSTREnv:
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}
    # physical ranges (twoseriescstr.py:56-61)
    raw_state_low = np.array([0.0, 273.15, 0.0, 273.15], dtype=np.float32)
    raw_state_high = np.array([0.7, 400.0, 0.7, 400.0], dtype=np.float32)
    raw_action_low = np.array([30.0, 30.0], dtype=np.float32)
    raw_action_high = np.array([250.0, 250.0], dtype=np.float32)
    dt = 0.1

    def __init__(self, render_mode: Optional[str] = None, default_target: float = 0.20, min_concentration: float = 0.05,
                 max_concentration: float = 0.45, init_mode: str = "random"):
        self.render_mode = render_mode
        self.observation_space = Box(-np.ones(4, np.float32), np.ones(4, np.float32), dtype=np.float32)
        self.action_space = Box(-np.ones(2, np.float32), np.ones(2, np.float32), dtype=np.float32)
        self.init_mode = init_mode
        self.max_steps = 400
        self.target_C2 = default_target
        self.min_concentration, self.max_concentration = min_concentration, max_concentration
        self.state = None
        self._vec = None
        self._seed: Optional[int] = None
        # memory of the zero-weight reward diagnostics (twoseriescstr.py:108-112): they never reach the reward, only `info`
        self.last_concentration = None
        self.last_action = None
        self.stable_counter = 0
        self.last_error = None

    # ctor arguments a batched env must share
    def vec_kwargs(self) -> dict:
        return dict(default_target=self.target_C2, min_concentration=self.min_concentration,
                    max_concentration=self.max_concentration, init_mode=self.init_mode)

    def _backend(self):
        if self._vec is None:
            from core.common.vec_env import CSTRVecEnv

            self._vec = CSTRVecEnv(1, **self.vec_kwargs())
        return self._vec

    @property
    def unwrapped(self):
        return self

    @property
    def current_step(self) -> int:
        return 0 if self._vec is None else int(self._vec.step_count[0])

    def set_target(self, target) -> bool:
        if self.min_concentration <= target <= self.max_concentration:
            self.target_C2 = target
            if self._vec is not None:
                self._vec.set_target(target)
            return True
        return False

    def seed(self, seed: Optional[int] = None):
        self._seed = seed
        return [seed]

    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None) -> Tuple[np.ndarray, dict]:
        v = self._backend()
        if seed is not None:
            self._seed = seed
        if self._seed is not None:
            v.seed(self._seed)
            self._seed = None
        self.last_concentration, self.last_action, self.stable_counter, self.last_error = None, None, 0, None  # :233-236
        obs = v.reset()[0]
        raw = self.raw_state_low + (obs + 1.0) * (self.raw_state_high - self.raw_state_low) / 2.0
        self.state = obs
        info = {"initial_concentration_1": raw[0], "initial_temperature_1": raw[1], "initial_concentration_2": raw[2],
                "initial_temperature_2": raw[3]}
        return obs.astype(np.float32), info

    def step(self, action: np.ndarray):
        """reference: twoseriescstr.py:394-454 -> (obs, reward, terminated=False, truncated, info). A single gym env does
        not auto-reset, so the VecEnv backend's post-step reset is undone here."""
        if self.state is None:
            raise ValueError("Please call env.reset() to reset the env first!")
        import torch as th

        v = self._backend()
        a = np.asarray(action, np.float32).reshape(1, 2)
        _, rew, _, timeout, nxt = v.step_device(th.as_tensor(a).to(v.device))
        obs = nxt.cpu().numpy()[0].copy()
        truncated = bool(timeout.cpu().numpy()[0])
        v.obs.copy_(nxt)  # keep the true next state (the VecEnv face would already hold the reset observation)
        if truncated:
            v.step_count.fill_(self.max_steps)
        self.state = obs
        reward = float(rew.cpu().numpy()[0])
        norm_a = np.clip(a[0], -1.0, 1.0).astype(np.float32)
        raw_action = (self.raw_action_low + (norm_a + 1.0) * (self.raw_action_high - self.raw_action_low) / 2.0).astype(np.float32)
        if np.isnan(a).any():  # the reference's exception path returns an EMPTY info dict (twoseriescstr.py:413-421)
            return obs, reward, False, truncated, {}
        original_state = (self.raw_state_low + (obs + 1.0) * (self.raw_state_high - self.raw_state_low) / 2.0).astype(np.float32)
        info = {"reward": reward, "raw_action": raw_action, "truncated": truncated, "state": obs, "original_state": original_state,
                "target_C2": self.target_C2, "step": self.current_step}
        info.update(self._reward_info(original_state, norm_a))
        return obs, reward, False, truncated, info

    def _reward_info(self, raw_state: np.ndarray, action: np.ndarray) -> dict:
        """The nine `info` entries of compute_reward (twoseriescstr.py:271-392) for the NEW state. The reward itself comes from the
        device step (concentration term + 0.5 x temperature penalty); the other five terms carry weight 0.0 in the reference and exist
        only here, with their per-env memory (last concentration / error / action, stable counter), on the NumPy compatibility face."""
        f = np.float32
        C1, T1, C2, T2 = (f(x) for x in raw_state)
        err = np.abs(C2 - self.target_C2)
        ne = err / (self.max_concentration - self.min_concentration)
        conc = -5.0 * (ne ** 2) - 2.0 * ne
        prox = (1.0 - err / 0.05) if err < 0.05 else 0.0
        if self.last_concentration is not None and self.last_error is not None:
            cur, prev = C2 - self.target_C2, self.last_concentration - self.target_C2
            trend = 0.5 if np.abs(cur) < np.abs(prev) else (-0.2 if np.abs(cur) > np.abs(prev) else 0.0)
        else:
            trend = 0.0
        self.last_concentration, self.last_error = C2, C2 - self.target_C2
        if err < 0.02:
            self.stable_counter += 1
            stab = min(2.0, 0.05 * self.stable_counter)
        else:
            self.stable_counter = max(0, self.stable_counter - 1)
            stab = 0.0
        temp = 0.0
        for T in (T1, T2):
            if T < 280:
                temp -= 0.2 * ((280 - T) / 280)
            elif T > 350:
                temp -= 0.5 * ((T - 350) / 350)
        if self.last_action is not None:
            smooth = max(-1.0, -0.05 * np.sum((action - self.last_action) ** 2))
        else:
            smooth = 0.0
        self.last_action = action.copy()
        extreme = 0.0
        if C2 < 0.005:
            extreme -= 1.0 * (1.0 - C2 / 0.005)
        elif C2 > 0.95 * self.max_concentration:
            extreme -= 1.0 * ((C2 - 0.95 * self.max_concentration) / (0.05 * self.max_concentration))
        return {"concentration_reward": conc, "concentration_proximity_reward": prox, "concentration_trend_reward": trend,
                "stability_reward": stab, "temp_penalty": temp, "action_smoothness_penalty": smooth, "extreme_penalty": extreme,
                "concentration_error": err, "stable_steps": self.stable_counter}

    def render(self):
        if self.render_mode == "human" and self.state is not None:
            raw = self.raw_state_low + (self.state + 1.0) * (self.raw_state_high - self.raw_state_low) / 2.0
            print(f"Step: {self.current_step}\nReactor 1: C1={raw[0]:.4f} mol/L, T1={raw[1]:.2f} K\n"
                  f"Reactor 2: C2={raw[2]:.4f} mol/L, T2={raw[3]:.2f} K\nTarget C2: {self.target_C2:.4f} mol/L")

    def close(self):
        self._vec = None
