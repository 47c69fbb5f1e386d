"""Host-side helpers with the reference's names and semantics (core/common/utils.py)."""
import random
from collections import deque
from itertools import zip_longest
from typing import Callable, Iterable, Union

import numpy as np
import torch as th

from core.common.type_aliases import TrainFreq, TrainFrequencyUnit

Schedule = Callable[[float], float]


def set_random_seed(seed: int, using_cuda: bool = False, device=None) -> None:
    """reference: core/common/utils.py:36-53. `np.random.seed` also seeds the device image of the legacy
    global stream (the one the replay sampler consumes)."""
    random.seed(seed)
    np.random.seed(seed)
    th.manual_seed(seed)
    if device is not None and th.device(device).type == "cuda":
        from core.common import legacy_rng

        legacy_rng.seed(seed, device)


def get_device(device: Union[th.device, str] = "auto") -> th.device:
    """reference: core/common/utils.py:147-167, minus the silent CPU fallback: this stack is MI355X-only."""
    if device == "auto":
        device = "cuda"
    device = th.device(device)
    if device.type != "cuda":
        raise ValueError(
            f"device={device}: this build runs the env / replay / update kernels in HIP on an MI355X and has no CPU "
            "path. Use device='cuda' (or 'auto').")
    if not th.cuda.is_available():
        raise RuntimeError("No HIP device is visible (torch.cuda.is_available() is False); this stack has no CPU fallback.")
    if device.index is None:
        device = th.device("cuda", th.cuda.current_device())
    return device


def constant_fn(val: float) -> Schedule:
    def func(_):
        return val

    return func


def get_schedule_fn(value_schedule: Union[Schedule, float]) -> Schedule:
    """reference: core/common/utils.py:88-105"""
    if isinstance(value_schedule, (float, int)):
        return constant_fn(float(value_schedule))
    assert callable(value_schedule)
    return lambda progress_remaining: float(value_schedule(progress_remaining))


def get_linear_fn(start: float, end: float, end_fraction: float) -> Schedule:
    def func(progress_remaining: float) -> float:
        if (1 - progress_remaining) > end_fraction:
            return end
        return start + (1 - progress_remaining) * (end - start) / end_fraction

    return func


def update_learning_rate(optimizer, learning_rate: float) -> None:
    """reference: core/common/utils.py:56-66 (works for torch optimisers and FlatAdam alike)"""
    for param_group in optimizer.param_groups:
        param_group["lr"] = learning_rate


def zip_strict(*iterables: Iterable) -> Iterable:
    sentinel = object()
    for combo in zip_longest(*iterables, fillvalue=sentinel):
        if sentinel in combo:
            raise ValueError("Iterables have different lengths")
        yield combo


def polyak_update(params: Iterable[th.Tensor], target_params: Iterable[th.Tensor], tau: float) -> None:
    """reference: core/common/utils.py:457-481. Tensor-list form kept for API compatibility: each pair goes through
    the HIP kernel (bit-identical to mul_ + add(alpha=)); the algorithms call it once per FLAT arena instead."""
    from core.common import hip_ops

    with th.no_grad():
        for param, target_param in zip_strict(params, target_params):
            if not (param.is_contiguous() and target_param.is_contiguous()):
                raise ValueError("polyak_update needs contiguous tensors")
            hip_ops.polyak(param.detach().view(-1), target_param.detach().view(-1), tau)


def should_collect_more_steps(train_freq: TrainFreq, num_collected_steps: int, num_collected_episodes: int) -> bool:
    """reference: core/common/utils.py:500-525"""
    if train_freq.unit == TrainFrequencyUnit.STEP:
        return num_collected_steps < train_freq.frequency
    elif train_freq.unit == TrainFrequencyUnit.EPISODE:
        return num_collected_episodes < train_freq.frequency
    raise ValueError(f"The unit of the `train_freq` must be either TrainFrequencyUnit.STEP "
                     f"or TrainFrequencyUnit.EPISODE not '{train_freq.unit}'!")


def safe_mean(arr) -> float:
    return float("nan") if len(arr) == 0 else float(np.mean(arr))


def get_parameters_by_name(model: th.nn.Module, included_names: Iterable[str]) -> list:
    return [param for name, param in model.state_dict().items() if any(key in name for key in included_names)]


__all__ = ["set_random_seed", "get_device", "get_schedule_fn", "constant_fn", "get_linear_fn", "update_learning_rate",
           "polyak_update", "zip_strict", "should_collect_more_steps", "safe_mean", "get_parameters_by_name", "deque"]
