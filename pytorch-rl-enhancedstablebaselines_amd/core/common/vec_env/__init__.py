"""`core.common.vec_env` (reference package: core/common/vec_env/__init__.py): the VecEnv protocol, the device-resident
CSTR environment and the DummyVecEnv drop-in. SubprocVecEnv / VecNormalize / video / frame-stack wrappers are out of
scope."""
from core.common.vec_env.base_vec_env import VecEnv
from core.common.vec_env.cstr_vec_env import CSTRVecEnv
from core.common.vec_env.dummy_vec_env import DummyVecEnv

__all__ = ["VecEnv", "CSTRVecEnv", "DummyVecEnv"]
