"""Dev-harness stand-in for the `gymnasium` package (NOT shipped, NOT used by the product).

The reference (`/root/reference`) imports gymnasium for class scaffolding only
(`gym.Env` base class, `spaces.Box` container, `seeding.np_random`). gymnasium is
absent from this image and cannot be installed, so the golden-vector generator
under tools/refharness loads the unmodified reference code against this stand-in.

What this weakens (documented in DESIGN.md / SURVEY.md 8c): any arithmetic that
lives inside real gymnasium follows this file instead -- `Box.sample()` (warm-up
actions) and `seeding.np_random` (reset draws). Fixtures therefore inject initial
states and actions explicitly; parity of those two pieces is "unpinned".
Everything else on the path is the reference's own code on real numpy/torch.
"""
from . import spaces, utils, error, logger  # noqa: F401
from .core import Env, Wrapper, ObservationWrapper, RewardWrapper, ActionWrapper  # noqa: F401
from .spaces import Space  # noqa: F401
from . import envs  # noqa: F401

__version__ = "0.29.1-standin"


def make(*args, **kwargs):
    raise error.DependencyNotInstalled("gymnasium stand-in: gym.make is not available")
