"""Fused learner path: the MLPs' GEMMs stay in PyTorch-ROCm (`torch.mm` -> rocBLAS), everything around them is
a hand-written HIP kernel, and gradients are written STRAIGHT into the flat arena:

  * `linear(x, W, b, act)`   = 1 GEMM + 1 epilogue launch (bias + ReLU/Tanh);
    backward                 = 1 epilogue-backward launch (activation gradient + bias gradient into the arena)
                               + `torch.mm(gz^T, x, out=<arena view of W.grad>)` + `torch.mm(gz, W)`.
    No AccumulateGrad adds, no zero_grad memsets: every parameter's gradient is (over)written exactly once per
    backward; frozen parameters (critic during the actor loss) skip the weight/bias GEMMs entirely.
  * `squashed_gaussian(mean, log_std, eps)` = 1 launch forward, 1 launch backward (analytic).

The nn.Modules keep owning the parameters (state_dict / API); `FastMLP` only reads their tensors. Arithmetic per
element is the reference's (core/common/torch_layers.py:110-183, core/common/distributions.py:161-260).
"""
from typing import List, Optional, Tuple

import torch as th
from torch import nn

from core.common import hip_ops

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


class _LinearFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act: int, train_params: bool):
        y = th.mm(x, weight.t())
        hip_ops.bias_act_fwd_(y, bias, act)
        ctx.act, ctx.train_params = act, train_params
        ctx.save_for_backward(x, weight, y)
        ctx.wgrad, ctx.bgrad = (weight.grad, bias.grad) if train_params else (None, None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        gbias = ctx.bgrad if ctx.train_params else None
        if ctx.act != ACT_NONE:
            gz = th.empty_like(gy)
            hip_ops.bias_act_bwd(gy, y, ctx.act, gz, gbias)
        else:
            gz = gy
            if gbias is not None:
                hip_ops.bias_act_bwd(gy, None, ACT_NONE, gy, gbias)
        if ctx.train_params:
            th.mm(gz.t(), x, out=ctx.wgrad)  # dW lands in the flat gradient arena
        dx = th.mm(gz, weight) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None


def linear(x: th.Tensor, weight: th.Tensor, bias: th.Tensor, act: int, train_params: bool) -> th.Tensor:
    """y = act(x @ W^T + b). With grad mode off this is just the two launches."""
    if not th.is_grad_enabled() or not (x.requires_grad or (train_params and weight.requires_grad)):
        y = th.mm(x, weight.t())
        return hip_ops.bias_act_fwd_(y, bias, act)
    if train_params and (weight.grad is None or bias.grad is None):
        raise RuntimeError("fused linear: parameter gradients must be views of a ParamArena gradient buffer")
    return _LinearFn.apply(x, weight, bias, act, train_params)


class FastMLP:
    """Reads an nn.Sequential built by create_mlp (Linear [+ ReLU | Tanh] ...) and evaluates it with `linear`."""

    def __init__(self, seq: nn.Sequential):
        self.layers: List[Tuple[nn.Linear, int]] = []
        mods = list(seq)
        i = 0
        while i < len(mods):
            lin = mods[i]
            if not isinstance(lin, nn.Linear) or lin.bias is None:
                raise NotImplementedError(f"FastMLP: unsupported module {lin}")
            act = ACT_NONE
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                act, i = ACT_RELU, i + 1
            elif i + 1 < len(mods) and isinstance(mods[i + 1], nn.Tanh):
                act, i = ACT_TANH, i + 1
            elif i + 1 < len(mods) and not isinstance(mods[i + 1], nn.Linear):
                raise NotImplementedError(f"FastMLP: unsupported activation {mods[i + 1]}")
            self.layers.append((lin, act))
            i += 1

    @staticmethod
    def supported(seq: nn.Sequential) -> bool:
        try:
            FastMLP(seq)
            return True
        except NotImplementedError:
            return False

    def __call__(self, x: th.Tensor, train_params: bool = True) -> th.Tensor:
        for lin, act in self.layers:
            x = linear(x, lin.weight, lin.bias, act, train_params)
        return x


class _SquashedGaussianFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, mean, log_std_raw, eps):
        mean, log_std_raw = mean.contiguous(), log_std_raw.contiguous()
        action = th.empty_like(mean)
        logp = th.empty(mean.shape[0], dtype=mean.dtype, device=mean.device)
        hip_ops.squashed_gaussian_fwd(mean, log_std_raw, eps, action, logp)
        ctx.save_for_backward(action, log_std_raw, eps)
        ctx.set_materialize_grads(False)
        return action, logp

    @staticmethod
    def backward(ctx, g_action, g_logp):
        action, log_std_raw, eps = ctx.saved_tensors
        g_mean, g_ls = th.empty_like(action), th.empty_like(action)
        hip_ops.squashed_gaussian_bwd(None if g_action is None else g_action.contiguous(),
                                      None if g_logp is None else g_logp.contiguous(), action, log_std_raw, eps, g_mean, g_ls)
        return g_mean, g_ls, None


def squashed_gaussian(mean: th.Tensor, log_std_raw: th.Tensor, eps: th.Tensor, want_logp: bool = True):
    """(action, logp) of SAC's squashed Gaussian with the log_std clamp folded in; logp is None when not wanted."""
    eps = eps.contiguous()
    if th.is_grad_enabled() and (mean.requires_grad or log_std_raw.requires_grad):
        return _SquashedGaussianFn.apply(mean, log_std_raw, eps)
    mean, log_std_raw = mean.contiguous(), log_std_raw.contiguous()
    action = th.empty_like(mean)
    logp = th.empty(mean.shape[0], dtype=mean.dtype, device=mean.device) if want_logp else None
    hip_ops.squashed_gaussian_fwd(mean, log_std_raw, eps, action, logp)
    return action, logp


class FastSacActor:
    """core/sac/policies.py:147-175 on the fused path."""

    def __init__(self, actor):
        self.actor = actor
        self.latent = FastMLP(actor.latent_pi)
        self.mu, self.log_std = actor.mu, actor.log_std

    def dist_params(self, obs: th.Tensor, train_params: bool = True):
        h = self.latent(obs, train_params)
        return (linear(h, self.mu.weight, self.mu.bias, ACT_NONE, train_params),
                linear(h, self.log_std.weight, self.log_std.bias, ACT_NONE, train_params))

    def action_log_prob(self, obs: th.Tensor, eps: Optional[th.Tensor] = None, train_params: bool = True, want_logp: bool = True):
        mean, ls = self.dist_params(obs, train_params)
        if eps is None:
            eps = self.actor.action_dist.draw_eps(mean.shape, mean.device)
        return squashed_gaussian(mean, ls, eps, want_logp)


class FastTwinCritic:
    """core/common/policies.py:960-987 on the fused path (n_critics Q networks on cat(obs, action)).

    The two Q networks are independent chains of ~5-us launches. With `two_streams` the second network is issued
    on a side HIP stream (fork after the concat, join before the outputs are used): under hipGraph capture the two
    chains become parallel branches of the graph, forward AND backward (autograd replays each node on the stream
    of its forward), so the pair costs about one chain's latency. Same kernels, same arithmetic."""

    def __init__(self, critic, two_streams: bool = True):
        self.nets = [FastMLP(q) for q in critic.q_networks]
        self.two_streams = two_streams and len(self.nets) == 2
        self._side: Optional[th.cuda.Stream] = None

    def __call__(self, obs: th.Tensor, actions: th.Tensor, train_params: bool = True, only_first: bool = False):
        x = th.cat([obs, actions], dim=1)
        if only_first or not self.two_streams:
            nets = self.nets[:1] if only_first else self.nets
            return tuple(net(x, train_params) for net in nets)
        cur = th.cuda.current_stream(x.device)
        if self._side is None:
            self._side = th.cuda.Stream(device=x.device)
        side = self._side
        side.wait_stream(cur)
        x.record_stream(side)
        q1 = self.nets[0](x, train_params)
        with th.cuda.stream(side):
            q2 = self.nets[1](x, train_params)
        cur.wait_stream(side)
        q2.record_stream(cur)
        return q1, q2

    def join(self) -> None:
        """After a backward pass through this critic: make the caller's stream wait for the side stream. The second
        network's weight/bias gradients are written into the arena by its backward nodes ON THE SIDE STREAM; autograd
        only synchronises streams along tensor edges it knows about, and those writes are side effects."""
        if self._side is not None:
            th.cuda.current_stream(self._side.device).wait_stream(self._side)
