"""IDDPG -- independent DDPG/TD3 learners per agent (reference: core/iddpg/iddpg.py:20-244, core/iddpg/policies.py).
The algorithm file of the reference is MADDPG's train() verbatim; the only difference is the critic: every agent's
twin Q networks see that agent's own observation slice and action slice instead of the joint vector."""
from core.maddpg.maddpg import MADDPG
from core.maddpg.policies import MADDPGPolicy


class IDDPGPolicy(MADDPGPolicy):
    local_critics = True


MlpPolicy = IDDPGPolicy


class IDDPG(MADDPG):
    policy_aliases = {"MlpPolicy": MlpPolicy}

    def learn(self, total_timesteps: int, callback=None, log_interval: int = 4, tb_log_name: str = "IDDPG",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        return super().learn(total_timesteps, callback, log_interval, tb_log_name, reset_num_timesteps, progress_bar)
