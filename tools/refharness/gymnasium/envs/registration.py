class EnvSpec:
    def __init__(self, id="standin-v0", max_episode_steps=None):
        self.id = id
        self.max_episode_steps = max_episode_steps
