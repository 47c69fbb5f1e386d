from core.td3.policies import MlpPolicy, TD3Policy
from core.td3.td3 import DDPG, TD3

__all__ = ["TD3", "DDPG", "MlpPolicy", "TD3Policy"]
