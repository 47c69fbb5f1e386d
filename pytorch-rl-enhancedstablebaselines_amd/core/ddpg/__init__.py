"""Import path of the reference (core/ddpg/__init__.py): DDPG = TD3 with policy_delay 1, one critic, no target smoothing."""
from core.td3.policies import MlpPolicy
from core.td3.td3 import DDPG

__all__ = ["DDPG", "MlpPolicy"]
