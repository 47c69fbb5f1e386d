"""`core` -- MI355X-native drop-in for the reference's `core` package on the SAC/TD3/MADDPG + two-series
CSTR path (reference: core/__init__.py:1-40). Algorithms are imported lazily so that host-only
utilities (and the CPU test-suite) do not need a GPU."""
import os

with open(os.path.join(os.path.dirname(__file__), "version.txt")) as _fh:  # the reference forgot to ship this file
    __version__ = _fh.read().strip()

__all__ = ["SAC", "TD3", "MADDPG", "IDDPG", "DDPG", "__version__"]


def __getattr__(name):
    if name == "SAC":
        from core.sac import SAC
        return SAC
    if name in ("TD3", "DDPG"):
        import core.td3 as m
        return getattr(m, name)
    if name == "IDDPG":
        from core.iddpg import IDDPG
        return IDDPG
    if name == "MADDPG":
        from core.maddpg import MADDPG
        return MADDPG
    raise AttributeError(f"module 'core' has no attribute {name!r}")
