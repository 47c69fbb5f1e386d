from core.sac.policies import MlpPolicy, SACPolicy
from core.sac.sac import SAC

__all__ = ["SAC", "MlpPolicy", "SACPolicy"]
