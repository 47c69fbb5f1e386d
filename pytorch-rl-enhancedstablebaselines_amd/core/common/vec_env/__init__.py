"""`core.common.vec_env` (reference package: core/common/vec_env/__init__.py): the VecEnv protocol, the device-resident
CSTR environment, the DummyVecEnv drop-in and the device VecNormalize. SubprocVecEnv / video / frame-stack wrappers are out
of scope."""
from core.common.vec_env.base_vec_env import VecEnv
from core.common.vec_env.cstr_vec_env import CSTRVecEnv
from core.common.vec_env.dummy_vec_env import DummyVecEnv
from core.common.vec_env.vec_normalize import VecNormalize, unwrap_vec_normalize

__all__ = ["VecEnv", "CSTRVecEnv", "DummyVecEnv", "VecNormalize", "unwrap_vec_normalize"]
