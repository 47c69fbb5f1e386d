from core.iddpg.iddpg import IDDPG, IDDPGPolicy, MlpPolicy

__all__ = ["IDDPG", "IDDPGPolicy", "MlpPolicy"]
