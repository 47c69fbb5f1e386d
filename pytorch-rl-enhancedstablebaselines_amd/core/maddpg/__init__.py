from core.maddpg.maddpg import MADDPG
from core.maddpg.policies import MADDPGPolicy, MlpPolicy

__all__ = ["MADDPG", "MlpPolicy", "MADDPGPolicy"]
