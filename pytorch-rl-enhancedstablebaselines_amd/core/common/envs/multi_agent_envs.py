"""Import path of the reference (core/common/envs/multi_agent_envs.py:7-61): `IndexedBox`, `split_spaces`."""
from core.common.spaces import IndexedBox, split_spaces

__all__ = ["IndexedBox", "split_spaces"]
