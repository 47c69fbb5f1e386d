"""Host-side logic of the facade that needs no GPU: spaces, schedules, train_freq conversion, logger, callbacks,
policy aliases / error conventions (reference: SURVEY 8b "Error conventions")."""
import numpy as np
import pytest
import torch as th

from core.common.spaces import Box, IndexedBox, as_box, split_spaces
from core.common.type_aliases import ReplayBufferSamples, TrainFreq, TrainFrequencyUnit
from core.common.utils import get_linear_fn, get_schedule_fn, should_collect_more_steps, zip_strict


def test_box_and_split_spaces():
    b = Box(-np.ones(4, np.float32), np.ones(4, np.float32))
    assert b.shape == (4,) and b.dtype == np.float32 and b.contains(np.zeros(4, np.float32)) and b.is_bounded()
    b.seed(3)
    s1 = b.sample_batch(5)
    b.seed(3)
    s2 = np.array([b.sample() for _ in range(5)])  # N sequential draws == one batched draw (row-major fill)
    np.testing.assert_array_equal(s1, s2)
    assert s1.dtype == np.float32 and np.all(np.abs(s1) <= 1)
    obs_l, act_l = split_spaces(b, Box(-1, 1, (2,)), [[0, 1], [2, 3]], [[0], [1]])
    assert isinstance(obs_l[1], IndexedBox) and list(obs_l[1].indices) == [2, 3] and act_l[0].shape == (1,)

    class Duck:  # any object with low/high/shape/dtype (e.g. a real gymnasium Box)
        low, high, shape, dtype = -np.ones(2), np.ones(2), (2,), np.float32

    assert as_box(Duck()) == Box(-1, 1, (2,))
    with pytest.raises(ValueError):
        as_box(object())


def test_named_tuple_field_order_is_api():
    assert ReplayBufferSamples._fields == ("observations", "actions", "next_observations", "dones", "rewards")


def test_schedules_and_collect_predicate():
    assert get_schedule_fn(3e-4)(0.1) == 3e-4
    f = get_linear_fn(1.0, 0.1, 0.5)
    assert f(1.0) == 1.0 and abs(f(0.75) - 0.55) < 1e-12 and f(0.2) == 0.1
    assert should_collect_more_steps(TrainFreq(2, TrainFrequencyUnit.STEP), 1, 0)
    assert not should_collect_more_steps(TrainFreq(2, TrainFrequencyUnit.STEP), 2, 0)
    assert should_collect_more_steps(TrainFreq(1, TrainFrequencyUnit.EPISODE), 99, 0)
    with pytest.raises(ValueError):
        list(zip_strict([1, 2], [1]))


def test_train_freq_conversion_errors():
    from core.common.off_policy_algorithm import OffPolicyAlgorithm

    class Probe(OffPolicyAlgorithm):
        def __init__(self, tf):
            self.train_freq = tf

    p = Probe(4)
    p._convert_train_freq()
    assert p.train_freq == TrainFreq(4, TrainFrequencyUnit.STEP)
    p = Probe((2, "episode"))
    p._convert_train_freq()
    assert p.train_freq == TrainFreq(2, TrainFrequencyUnit.EPISODE)
    with pytest.raises(ValueError, match="must be either 'step' or 'episode'"):
        Probe((1, "minutes"))._convert_train_freq()
    with pytest.raises(ValueError, match="must be an integer"):
        Probe((0.5, "step"))._convert_train_freq()


def test_logger_lazy_device_mean_and_callbacks():
    from core.common.callbacks import CallbackList, ConvertCallback, NoopCallback, to_callback
    from core.common.logger import DeviceMean, Logger

    lg = Logger()
    lg.record("train/critic_loss", DeviceMean(th.tensor(6.0), 3))
    lg.record("time/fps", 123)
    assert float(lg.name_to_value["train/critic_loss"]) == 2.0
    lg.dump(step=7)
    assert lg.last_dump["train/critic_loss"] == 2.0 and lg.last_dump["step"] == 7 and not lg.name_to_value
    assert getattr(to_callback(None), "is_noop", False) and isinstance(to_callback(None), NoopCallback)
    calls = []
    cb = to_callback([ConvertCallback(lambda l, g: calls.append(1) or True), ConvertCallback(lambda l, g: False)])
    assert isinstance(cb, CallbackList)

    class M:
        num_timesteps = 0

    cb.init_callback(M())
    assert cb.on_step() is False and calls == [1]  # returning False aborts training


def test_policy_alias_and_multi_env_errors():
    from core.sac import SAC

    with pytest.raises(ValueError, match="Policy CnnPolicy unknown"):
        SAC._get_policy_from_name(SAC, "CnnPolicy")
    assert SAC._get_policy_from_name(SAC, "MlpPolicy").__name__ == "SACPolicy"


def test_dummy_vec_env_shim_rejects_foreign_envs():
    from core.common.vec_env import DummyVecEnv

    class Other:
        pass

    with pytest.raises(ValueError, match="vectorises TwoSeriesCSTREnv only"):
        DummyVecEnv([Other, Other])
    o = Other()
    with pytest.raises(ValueError, match="returned the same"):
        DummyVecEnv([lambda: o, lambda: o])
