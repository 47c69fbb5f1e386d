"""MLP builders (reference: core/common/torch_layers.py:33-46, :110-183, :316-373). Module construction order
is the reference's, so `th.manual_seed(s)` yields the same initial weights (tests/golden/policy_init_kat.npz)."""
from typing import Union

import torch as th
from torch import nn


class FlattenExtractor(nn.Module):
    def __init__(self, features_dim: int):
        super().__init__()
        self.features_dim = features_dim
        self.flatten = nn.Flatten()

    def forward(self, observations: th.Tensor) -> th.Tensor:
        return self.flatten(observations)


def create_mlp(input_dim: int, output_dim: int, net_arch: list, activation_fn=nn.ReLU, squash_output: bool = False,
               with_bias: bool = True) -> list:
    modules: list = []
    last = input_dim
    for width in net_arch:
        modules.append(nn.Linear(last, width, bias=with_bias))
        modules.append(activation_fn())
        last = width
    if output_dim > 0:
        modules.append(nn.Linear(last, output_dim, bias=with_bias))
    if squash_output:
        modules.append(nn.Tanh())
    return modules


def get_actor_critic_arch(net_arch: Union[list, dict]) -> tuple:
    if isinstance(net_arch, list):
        return net_arch, net_arch
    assert isinstance(net_arch, dict), "Error: the net_arch can only contain be a list of ints or a dict"
    assert "pi" in net_arch, "Error: no key 'pi' was provided in net_arch for the actor network"
    assert "qf" in net_arch, "Error: no key 'qf' was provided in net_arch for the critic network"
    return net_arch["pi"], net_arch["qf"]
