"""`DummyVecEnv` drop-in (reference: core/common/vec_env/dummy_vec_env.py:16-140)."""
from core.common.vec_env.base_vec_env import VecEnv
from core.common.vec_env.cstr_vec_env import CSTRVecEnv


def DummyVecEnv(env_fns) -> VecEnv:
    """Drop-in for `DummyVecEnv([make_env] * N)` (reference: core/common/vec_env/dummy_vec_env.py:30-54): when every
    factory returns a `TwoSeriesCSTREnv` with the same constructor arguments the N Python envs collapse into one
    device-resident `CSTRVecEnv(N)`. Anything else is outside this stack's scope."""
    from twoseriescstr import TwoSeriesCSTREnv

    envs = [fn() for fn in env_fns]
    if len(set(id(e) for e in envs)) != len(envs):
        raise ValueError("You tried to create multiple environments, but the function to create them returned the same "
                         "instance instead of creating different objects.")  # dummy_vec_env.py:32-42
    if not envs or not all(isinstance(e, TwoSeriesCSTREnv) for e in envs):
        raise ValueError("This MI355X build vectorises TwoSeriesCSTREnv only; got " + ", ".join(sorted({type(e).__name__ for e in envs})))
    kw = envs[0].vec_kwargs()
    if any(e.vec_kwargs() != kw for e in envs[1:]):
        raise ValueError("All TwoSeriesCSTREnv instances of one vectorised env must share their constructor arguments")
    return CSTRVecEnv(len(envs), **kw)
