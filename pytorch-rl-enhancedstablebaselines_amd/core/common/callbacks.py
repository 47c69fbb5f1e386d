"""Callback protocol of the reference (core/common/callbacks.py:30-170): hook points only."""
from typing import Callable, Optional, Union


class BaseCallback:
    def __init__(self, verbose: int = 0):
        self.model = None
        self.n_calls = 0
        self.num_timesteps = 0
        self.verbose = verbose
        self.locals: dict = {}
        self.globals: dict = {}
        self.parent = None

    @property
    def training_env(self):
        return self.model.get_env()

    @property
    def logger(self):
        return self.model.logger

    def init_callback(self, model) -> None:
        self.model = model
        self._init_callback()

    def _init_callback(self) -> None:
        pass

    def on_training_start(self, locals_: dict, globals_: dict) -> None:
        self.locals, self.globals = locals_, globals_
        self.num_timesteps = self.model.num_timesteps
        self._on_training_start()

    def _on_training_start(self) -> None:
        pass

    def on_rollout_start(self) -> None:
        self._on_rollout_start()

    def _on_rollout_start(self) -> None:
        pass

    def _on_step(self) -> bool:
        return True

    def on_step(self) -> bool:
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    def on_training_end(self) -> None:
        self._on_training_end()

    def _on_training_end(self) -> None:
        pass

    def on_rollout_end(self) -> None:
        self._on_rollout_end()

    def _on_rollout_end(self) -> None:
        pass

    def update_locals(self, locals_: dict) -> None:
        self.locals.update(locals_)
        self.update_child_locals(locals_)

    def update_child_locals(self, locals_: dict) -> None:
        pass


class NoopCallback(BaseCallback):
    """What `callback=None` becomes; lets the loop skip `locals()` snapshots."""
    is_noop = True


class CallbackList(BaseCallback):
    def __init__(self, callbacks: list):
        super().__init__()
        self.callbacks = callbacks

    def _init_callback(self) -> None:
        for cb in self.callbacks:
            cb.init_callback(self.model)
            cb.parent = self.parent

    def _on_training_start(self) -> None:
        for cb in self.callbacks:
            cb.on_training_start(self.locals, self.globals)

    def _on_rollout_start(self) -> None:
        for cb in self.callbacks:
            cb.on_rollout_start()

    def _on_step(self) -> bool:
        cont = True
        for cb in self.callbacks:
            cont = cb.on_step() and cont
        return cont

    def _on_rollout_end(self) -> None:
        for cb in self.callbacks:
            cb.on_rollout_end()

    def _on_training_end(self) -> None:
        for cb in self.callbacks:
            cb.on_training_end()

    def update_child_locals(self, locals_: dict) -> None:
        for cb in self.callbacks:
            cb.update_locals(locals_)


class ConvertCallback(BaseCallback):
    def __init__(self, callback: Optional[Callable[[dict, dict], bool]], verbose: int = 0):
        super().__init__(verbose)
        self.callback = callback

    def _on_step(self) -> bool:
        if self.callback is not None:
            return self.callback(self.locals, self.globals)
        return True


MaybeCallback = Union[None, Callable, list, BaseCallback]


def to_callback(callback: MaybeCallback) -> BaseCallback:
    """reference: core/common/base_class.py:382-404"""
    if callback is None:
        return NoopCallback()
    if isinstance(callback, list):
        return CallbackList(callback)
    if not isinstance(callback, BaseCallback):
        return ConvertCallback(callback)
    return callback
