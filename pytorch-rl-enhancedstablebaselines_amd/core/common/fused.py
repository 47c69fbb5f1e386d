"""Fused learner path: every Linear layer of the learners at batch size is ONE hand-written f32-MFMA launch with its bias and
activation (backward: the input gradient times the activation gradient of the layer below in one launch, dW + db of ALL layers of
an MLP in one pointer-table launch), everything around them is a hand-written HIP kernel too, and gradients are written STRAIGHT
into the flat arena (no AccumulateGrad adds, no zero_grad memsets: every parameter's gradient is (over)written exactly once per
backward; frozen parameters -- the critic during the actor loss -- skip the weight / bias launches entirely). CSTR_FUSED_LINEAR=0
leaves every GEMM to PyTorch-ROCm (`torch.mm` -> rocBLAS) with the HIP epilogues around it (the north_star-literal form).

  * twin critics = ONE chain of grouped launches over stacked arena views; the critic and the TARGET critic of a gradient step =
    ONE four-network pointer-table chain (`_TwinPairFn`); the actor's mu / log_std heads = one head kernel;
  * SAC's pi(obs) and pi(next_obs) = ONE 2B-row actor pass, backward on the B differentiated rows (`_ActorPairFn`);
  * a Q network's last hidden layer + scalar head (`hidden_head`) = 1 GEMM launch + 1 launch; its backward's first launch can
    carry the loss that roots it (`set_loss_root`: TD critic loss, SAC actor loss);
  * deterministic actors write their action into the critic input and read its gradient columns in place (`_LinearXbufFn`,
    `_ActorGroupFn`): no torch.cat, no gather copies;
  * `squashed_gaussian([mean | log_std], eps)` = 1 launch forward, 1 launch backward (analytic).

The nn.Modules keep owning the parameters (state_dict / API); `FastMLP` only reads their tensors. Arithmetic per
element is the reference's (core/common/torch_layers.py:110-183, core/common/distributions.py:161-260).
"""
import contextlib
import os
from typing import List, Optional, Tuple

import torch as th
from torch import nn

from core.common import hip_ops

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2

# inference passes with more rows than this run the whole policy network in one launch (cstr_policy_rows_fwd_f32): a workgroup
# per 16 rows; below it the per-layer kernels spread the same work over more CUs and win (tools/policy_probe.py)
WHOLE_NET_MIN_ROWS = 1024

# Linear + bias + activation forward in ONE launch (cstr_linear_act_fwd_f32, f32 matrix cores) where it beats the rocBLAS GEMM
# + epilogue pair: measured on MI355X (tools/linear_probe.py) for every K at batch-sized M and for narrow inputs (K <= 32) at
# any M; the 4096-row x 256 x 256 layers of the collect-time actor stay on rocBLAS. CSTR_FUSED_LINEAR=0 turns it off.
USE_FUSED_LINEAR = os.environ.get("CSTR_FUSED_LINEAR", "1") != "0"
# development A/B knobs: the 2B-row actor pass (_ActorPairFn) and the four-network critic / target chain (_TwinPairFn)
USE_ACTOR_PAIR = os.environ.get("CSTR_ACTOR_PAIR", "1") != "0"
# SAC: when the rollout launch has drawn the batch indices, the 2B-row actor pass's first layer gathers the sampled rows itself
# (hip_ops.linear_act_fwd_gather) instead of a gather launch in front of it; "0" keeps the gather launch (A/B, bit-identical)
USE_GATHER_IN_FIRST_LAYER = os.environ.get("CSTR_GATHER_IN_L1", "1") != "0"
# TD3: the target actor's last layer adds the target policy smoothing itself (hip_ops.linear_smooth_fwd); "0" = separate launch
USE_SMOOTH_IN_LAST_LAYER = os.environ.get("CSTR_SMOOTH_IN_LAST", "1") != "0"
USE_TWIN_PAIR = os.environ.get("CSTR_TWIN_PAIR", "1") != "0"
USE_LOSS_ROOT = os.environ.get("CSTR_LOSS_ROOT", "1") != "0"  # the loss launches ride in the backward's first launch


def _fused_linear_ok(x: th.Tensor) -> bool:
    if not USE_FUSED_LINEAR or x.stride(-1) != 1:
        return False
    rows = x.shape[-2] * (x.shape[0] if x.dim() == 3 else 1)
    return x.shape[-1] <= 32 or rows <= 1024


def _linear_fwd(x: th.Tensor, weight: th.Tensor, bias: th.Tensor, act: int) -> th.Tensor:
    """act(x @ W^T + b), plain or stacked ([G, M, K] x [G, N, K]): one fused launch or GEMM + epilogue"""
    if _fused_linear_ok(x):
        return hip_ops.linear_act_fwd(x, weight, bias, act)
    y = th.bmm(x, weight.transpose(1, 2)) if x.dim() == 3 else th.mm(x, weight.t())
    return hip_ops.bias_act_fwd_(y, bias, act)


# Inside a chain of fused layers the gradient that travels DOWN between two Linears is dz (w.r.t. the lower layer's
# pre-activation), not dy: the upper layer's input-gradient kernel applies the lower layer's activation gradient while the
# tile is still in registers (cstr_linear_bwd_input_f32), so the lower layer's backward is ONE more launch: dW and db
# together (cstr_linear_bwd_weight_f32). `below` = (activation, None) of the fused layer whose OUTPUT this layer's input is;
# `grad_is_dz` marks a layer whose consumer does that for it. Both are set by the chain builders below (FastMLP,
# FastTwinCritic, FastSacActor) -- the tensors in between are private to the chain.
def _input_grad(gz: th.Tensor, weight: th.Tensor, x: th.Tensor, below) -> th.Tensor:
    """d(loss)/d(input) of a Linear given gz = d(loss)/d(pre-activation); with `below`, d(loss)/d(lower pre-activation)."""
    if below is None or not USE_FUSED_LINEAR:
        dx = th.bmm(gz, weight) if gz.dim() == 3 else th.mm(gz, weight)
        if below is not None and below[0] != ACT_NONE:  # unfused arithmetic of the same contract
            out = th.empty_like(dx)
            hip_ops.bias_act_bwd(dx, x, below[0], out, None)
            return out
        return dx
    return hip_ops.linear_bwd_input(gz, weight, x, below[0])


def _own_grad(ctx, gy: th.Tensor, y: th.Tensor) -> th.Tensor:
    """gz of this layer from what arrived: already dz when the consumer is a fused layer, otherwise dy -> dz here."""
    gy = gy.contiguous()
    if ctx.grad_is_dz or ctx.act == ACT_NONE:
        return gy
    gz = th.empty_like(gy)
    hip_ops.bias_act_bwd(gy, y, ctx.act, gz, None)
    return gz


_pending_wgrads: Optional[list] = None


@contextlib.contextmanager
def deferred_weight_grads():
    """Inside this context the fused layers' backward passes only QUEUE their (2-D) weight / bias gradients; leaving it
    computes all of them in one launch (`cstr_linear_bwd_weight_sets_f32`). Parameter gradients are leaves of the backward
    pass -- nothing but the optimiser waits for them -- so an MLP's backward becomes its dz chain + ONE launch."""
    global _pending_wgrads
    if not USE_FUSED_LINEAR or _pending_wgrads is not None:
        yield
        return
    _pending_wgrads = []
    try:
        yield
    finally:
        pending, _pending_wgrads = _pending_wgrads, None
    for i in range(0, len(pending), hip_ops.nv.MAX_LINEAR_SETS):
        chunk = pending[i:i + hip_ops.nv.MAX_LINEAR_SETS]
        if len(chunk) == 1:
            hip_ops.linear_bwd_weight(*chunk[0])
        else:
            hip_ops.linear_bwd_weight_sets(chunk)


def _weight_grad(gz: th.Tensor, x: th.Tensor, wgrad: th.Tensor, bgrad: Optional[th.Tensor]) -> None:
    """dW (+ db) of one fused Linear: one launch now, or queued when a `deferred_weight_grads` context is open."""
    if _pending_wgrads is None:
        hip_ops.linear_bwd_weight(gz, x, wgrad, bgrad)
    elif gz.dim() == 2:
        _pending_wgrads.append((gz, x, wgrad, bgrad))
    else:  # stacked networks: one set per group (a 2-D x is the input the groups share)
        for g in range(gz.shape[0]):
            _pending_wgrads.append((gz[g], x[g] if x.dim() == 3 else x, wgrad[g], None if bgrad is None else bgrad[g]))


def _param_grads(ctx, gz: th.Tensor, x: th.Tensor) -> None:
    """dW = gz^T x and db = column sums of gz, written into the gradient arena views."""
    if not ctx.train_params:
        return
    if USE_FUSED_LINEAR:
        _weight_grad(gz, x, ctx.wgrad, ctx.bgrad)  # one launch for both
        return
    th.bmm(gz.transpose(1, 2), x, out=ctx.wgrad) if gz.dim() == 3 else th.mm(gz.t(), x, out=ctx.wgrad)
    hip_ops.bias_act_bwd(gz, None, ACT_NONE, gz, ctx.bgrad)


class _LinearFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act: int, train_params: bool, below, grad_is_dz: bool):
        y = _linear_fwd(x, weight, bias, act)
        ctx.act, ctx.train_params, ctx.below, ctx.grad_is_dz = act, train_params, below, grad_is_dz
        ctx.save_for_backward(x, weight, y)
        ctx.wgrad, ctx.bgrad = (weight.grad, bias.grad) if train_params else (None, None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gz = _own_grad(ctx, gy, y)
        _param_grads(ctx, gz, x)  # dW / db land in the flat gradient arena
        dx = _input_grad(gz, weight, x, ctx.below) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, None, None


class _LinearXbufFn(th.autograd.Function):
    """`_LinearFn` whose output lands in the LAST N columns of a wider row-major buffer (a critic input whose observation columns
    are already filled: torch.cat((obs, action)) without the launch) and whose gradient is read from those columns of the buffer's
    gradient in place (cstr_bias_act_bwd_rows_f32: no gather copy). The buffer is the output."""

    @staticmethod
    def forward(ctx, x, weight, bias, act: int, train_params: bool, below, xbuf):
        n = weight.shape[0]
        y = xbuf[:, xbuf.shape[1] - n:]
        hip_ops.linear_act_fwd_sets([(x, weight, bias, y)], act)
        ctx.act, ctx.train_params, ctx.below, ctx.n = act, train_params, below, n
        ctx.save_for_backward(x, weight)
        ctx.y = y.detach()
        ctx.wgrad, ctx.bgrad = (weight.grad, bias.grad) if train_params else (None, None)
        ctx.mark_dirty(xbuf)
        return xbuf

    @staticmethod
    def backward(ctx, g_buf):
        x, weight = ctx.saved_tensors
        if g_buf.stride(1) != 1:
            g_buf = g_buf.contiguous()
        gy = g_buf[:, g_buf.shape[1] - ctx.n:]
        if ctx.act != ACT_NONE:
            gz = hip_ops.bias_act_bwd_rows(gy, ctx.y, ctx.act, th.empty(gy.shape, dtype=gy.dtype, device=gy.device))
        else:
            gz = gy.contiguous()
        _param_grads(ctx, gz, x)
        dx = _input_grad(gz, weight, x, ctx.below) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, None, None


def linear(x: th.Tensor, weight: th.Tensor, bias: th.Tensor, act: int, train_params: bool, below=None,
           grad_is_dz: bool = False) -> th.Tensor:
    """y = act(x @ W^T + b). With grad mode off this is just the forward launch(es)."""
    if not th.is_grad_enabled() or not (x.requires_grad or (train_params and weight.requires_grad)):
        return _linear_fwd(x, weight, bias, act)
    if train_params and (weight.grad is None or bias.grad is None):
        raise RuntimeError("fused linear: parameter gradients must be views of a ParamArena gradient buffer")
    return _LinearFn.apply(x, weight, bias, act, train_params, below if x.requires_grad else None, grad_is_dz)


class FastMLP:
    """Reads an nn.Sequential built by create_mlp (Linear [+ ReLU | Tanh] ...) and evaluates it with `linear`."""

    def __init__(self, seq: nn.Sequential, optimizer=None):
        """`optimizer`: the FlatAdam that updates these parameters, if the many-row inference pass may keep a tile-major
        copy of the second layer with it (see _weight_shadow)."""
        self.optimizer = optimizer
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

    def _whole_net_ok(self, x: th.Tensor, train_params: bool) -> bool:
        """No gradient wanted, many rows, and the network is create_mlp(k0, A <= 8, [H1, H2]) with one hidden activation."""
        if not USE_FUSED_LINEAR or (th.is_grad_enabled() and (train_params or x.requires_grad)):
            return False
        layers = self.layers
        if len(layers) != 3 or x.dim() != 2 or x.shape[0] <= WHOLE_NET_MIN_ROWS or x.stride(1) != 1 or layers[0][1] != layers[1][1]:
            return False
        l1, l2, l3 = (lin for lin, _ in layers)
        return (l3.out_features <= 2 * hip_ops.nv.MAX_HEAD_ACT
                and hip_ops.policy_rows_supported(l1.in_features, l1.out_features, l2.out_features, l3.out_features)
                and all(lin.weight.is_contiguous() and lin.weight.data_ptr() % 16 == 0 for lin in (l2, l3)) and l1.weight.is_contiguous())

    def rollout_operands(self, obs: th.Tensor) -> Optional[dict]:
        """This network (a deterministic actor) as operands of hip_ops.rollout_step, or None if that launch does not cover it."""
        with th.no_grad():
            if not self._whole_net_ok(obs, train_params=False):
                return None
        (l1, act), (l2, _), (l3, out_act) = self.layers
        swz = _weight_shadow(self.optimizer, l2.weight)
        if not hip_ops.rollout_step_supported(l1.in_features, l1.out_features, l2.out_features, l3.out_features, swz is not None):
            return None
        if obs.stride(0) % 4 or obs.data_ptr() % 16 or l1.weight.data_ptr() % 16 or not l3.weight.is_contiguous():
            return None
        return dict(weights=(l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias), act=act, head=1, out_act=out_act, w2_swz=swz,
                    rng_ctl=None)

    def _tail(self, x: th.Tensor, first: int, smooth: Optional[dict] = None) -> th.Tensor:
        """layers[first:] of an inference pass (no gradients kept). `smooth`: see `__call__`."""
        last = len(self.layers) - 1
        for li, (lin, act) in enumerate(self.layers[first:], first):
            if li == last and smooth is not None:
                return hip_ops.linear_smooth_fwd(x, lin.weight, lin.bias, act, smooth.get("noise"), smooth.get("rng_ctl"), smooth["sigma"],
                                                 smooth["clip"], smooth["out"])
            x = linear(x, lin.weight, lin.bias, act, False, None, grad_is_dz=False)
        return x

    def smooth_supported(self, x: th.Tensor) -> bool:
        """`__call__(x, train_params=False, smooth=...)` applies: a per-layer inference pass whose last Linear can add the target
        policy smoothing itself (hip_ops.linear_smooth_fwd)."""
        layers = self.layers
        with th.no_grad():
            whole = self._whole_net_ok(x, train_params=False)
        lin = layers[-1][0]
        return (USE_FUSED_LINEAR and not whole and len(layers) >= 2 and x.dim() == 2 and x.shape[0] <= 1024
                and hip_ops.linear_smooth_supported(x.shape[0], lin.out_features, lin.in_features) and lin.weight.is_contiguous())

    def tail_below(self, train_params: bool):
        """what a fused consumer of this MLP's output needs to run the last layer's activation / bias gradient itself"""
        return (self.layers[-1][1], None)

    def gather_supported(self, x: th.Tensor) -> bool:
        """`__call__(x, train_params=False, gather=...)` applies: an inference pass through the per-layer kernels whose first layer
        can read the sampled next observations from the ring itself (hip_ops.linear_act_fwd_gather)."""
        layers = self.layers
        with th.no_grad():
            whole = self._whole_net_ok(x, train_params=False)
        lin = layers[0][0]
        return (USE_FUSED_LINEAR and not whole and len(layers) >= 2 and x.dim() == 2 and lin.in_features == x.shape[1]
                and lin.weight.is_contiguous() and lin.weight.data_ptr() % 16 == 0 and not (len(layers) == 2 and layers[-1][0].out_features == 1))

    def __call__(self, x: th.Tensor, train_params: bool = True, out_grad_is_dz: bool = False, xbuf: Optional[th.Tensor] = None,
                 gather=None, smooth: Optional[dict] = None) -> th.Tensor:
        """`out_grad_is_dz`: the consumer is a fused layer built with `below=self.tail_below(...)` (see _input_grad).
        `xbuf` [M, W]: the output is written into its last out_features columns and the BUFFER is returned (a deterministic actor's
        action straight into the critic input, differentiable).
        `gather` = ReplayBuffer.take_predrawn(pb) (inference only, `gather_supported`): x is the not-yet-gathered next-observation
        block of `pb`; the first layer fetches the sampled rows from the ring and writes `pb` for the launches behind it.
        `smooth` = dict(noise | rng_ctl, sigma, clip, out) (inference only, `smooth_supported`): the last layer adds the target policy
        smoothing and writes the result into `out` (the action columns of the target critic's input), which is returned."""
        layers = self.layers
        if smooth is not None and (th.is_grad_enabled() or train_params or xbuf is not None or not self.smooth_supported(x)):
            raise NotImplementedError("FastMLP smooth: a no-grad inference pass through the per-layer kernels only")
        if gather is None and smooth is not None:
            return FastMLP._tail(self, x, 0, smooth)
        if gather is not None:
            if th.is_grad_enabled() or train_params or xbuf is not None or not self.gather_supported(x):
                raise NotImplementedError("FastMLP gather: a no-grad inference pass through the per-layer kernels only")
            ring, idx, rng_advance, pb = gather
            lin, act = layers[0]
            h = hip_ops.linear_act_fwd_gather(ring, idx, x.shape[0], False, lin.weight, lin.bias, act, pb.x_data, pb.x_next, pb.x_pi,
                                              pb.samples.dones, pb.samples.rewards, advance_ring=True, rng_advance=rng_advance)
            return FastMLP._tail(self, h, 1, smooth)
        if xbuf is not None:
            if not USE_FUSED_LINEAR or not th.is_grad_enabled() or layers[-1][0].out_features == 1 or x.stride(-1) != 1:
                raise NotImplementedError("FastMLP xbuf: the fused, differentiated, non-scalar-head form only")
            below = None
            for i, (lin, act) in enumerate(layers[:-1]):
                x = linear(x, lin.weight, lin.bias, act, train_params, below, grad_is_dz=True)
                below = (act, None)
            lin, act = layers[-1]
            if train_params and (lin.weight.grad is None or lin.bias.grad is None):
                raise RuntimeError("fused linear: parameter gradients must be views of a ParamArena gradient buffer")
            return _LinearXbufFn.apply(x, lin.weight, lin.bias, act, train_params, below if x.requires_grad else None, xbuf)
        if self._whole_net_ok(x, train_params):  # a deterministic actor's rollout pass: ONE launch, nothing kept
            (l1, act), (l2, _), (l3, out_act) = layers
            out = th.empty(x.shape[0], l3.out_features, dtype=x.dtype, device=x.device)
            return hip_ops.policy_rows_fwd(x, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, act, 1, out_act, out,
                                           w2_swz=_weight_shadow(self.optimizer, l2.weight))
        scalar_head = len(layers) >= 2 and layers[-1][0].out_features == 1 and layers[-1][1] == ACT_NONE
        plain = layers[:-2] if scalar_head else layers
        below = None
        for i, (lin, act) in enumerate(plain):
            inner = scalar_head or i < len(plain) - 1 or out_grad_is_dz
            x = linear(x, lin.weight, lin.bias, act, train_params, below, grad_is_dz=inner)
            below = (act, None)
        if scalar_head:
            (l1, act), (l2, _) = layers[-2:]
            grads = (l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad) if train_params else None
            if train_params and th.is_grad_enabled() and any(g is None for g in grads):
                raise RuntimeError("fused linear: parameter gradients must be views of a ParamArena gradient buffer")
            x = hidden_head(x, l1.weight, l1.bias, l2.weight, l2.bias, grads, act, train_params, (l1.weight, l2.weight), below)
        return x


class _ActorGroupFn(th.autograd.Function):
    """Grouped forward of several agents' actors where ONE agent is differentiated (MADDPG's actor loss: the other agents'
    actions enter the joint critic input as constants, core/maddpg/maddpg.py:167-177). The buffer `out` receives every
    agent's action in its column block and carries the differentiated agent's history."""

    @staticmethod
    def forward(ctx, out, group, inputs, col_ranges, agent: int, *owners):
        acts_i = group._run(inputs, out, col_ranges, keep=agent)  # agent's per-layer outputs; the last one is its column block of `out`
        ctx.group, ctx.agent, ctx.cols = group, agent, col_ranges[agent]
        ctx.save_for_backward(inputs[agent], *acts_i[:-1])
        ctx.y_last = acts_i[-1]
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x0, *ys = ctx.saved_tensors
        ys.append(ctx.y_last)
        layers = ctx.group.mlps[ctx.agent].layers
        lo, hi = ctx.cols
        if g_out.stride(1) != 1:
            g_out = g_out.contiguous()
        gz = g_out[:, lo:hi]
        last_act = layers[-1][1]
        if last_act != ACT_NONE:
            gz = hip_ops.bias_act_bwd_rows(gz, ys[-1], last_act, th.empty(gz.shape, dtype=gz.dtype, device=gz.device))
        else:
            gz = gz.contiguous()
        for li in range(len(layers) - 1, -1, -1):
            lin = layers[li][0]
            x = x0 if li == 0 else ys[li - 1]
            _weight_grad(gz, x, lin.weight.grad, lin.bias.grad)
            if li > 0:
                gz = hip_ops.linear_bwd_input(gz, lin.weight, ys[li - 1], layers[li - 1][1])
        return (None,) * len(ctx.needs_input_grad)


class FastActorGroup:
    """Every agent's deterministic actor MLP (identical architectures, parameters in per-agent arena slices) evaluated layer by
    layer with ONE launch per layer (cstr_linear_act_fwd_sets_f32) instead of one chain per agent; the last layer writes each
    agent's action into its column block of the caller's buffer (joint action / joint critic input): no torch.cat."""

    def __init__(self, mlps: List["FastMLP"]):
        self.mlps = mlps

    @staticmethod
    def supported(mlps: List["FastMLP"]) -> bool:
        if not USE_FUSED_LINEAR or not 1 < len(mlps) <= hip_ops.nv.MAX_LINEAR_SETS:
            return False
        sig = lambda m: [(l.in_features, l.out_features, a) for l, a in m.layers]  # noqa: E731
        return all(sig(m) == sig(mlps[0]) for m in mlps[1:])

    def _run(self, inputs, out, col_ranges, keep: Optional[int] = None):
        n_layers = len(self.mlps[0].layers)
        hs, kept = list(inputs), []
        m = inputs[0].shape[0]
        for li in range(n_layers):
            act = self.mlps[0].layers[li][1]
            n = self.mlps[0].layers[li][0].out_features
            last = li == n_layers - 1
            if last:  # (the differentiated agent's backward reads its column block in place: cstr_bias_act_bwd_rows_f32)
                ys = [out[:, lo:hi] for lo, hi in col_ranges]
            else:
                buf = th.empty(len(self.mlps), m, n, dtype=out.dtype, device=out.device)
                ys = [buf[j] for j in range(len(self.mlps))]
            hip_ops.linear_act_fwd_sets([(hs[j], self.mlps[j].layers[li][0].weight, self.mlps[j].layers[li][0].bias, ys[j])
                                         for j in range(len(self.mlps))], act)
            if keep is not None:
                kept.append(ys[keep].detach() if last else ys[keep])
            hs = ys
        return kept

    def forward(self, inputs, out: th.Tensor, col_ranges, grad_agent: Optional[int] = None) -> th.Tensor:
        """inputs[j] [M, K0] (row-strided views ok); out [M, W]; agent j's action -> out[:, col_ranges[j][0]:col_ranges[j][1]].
        grad_agent = i: agent i's parameters receive gradients (written into their arena views) through the returned buffer."""
        if grad_agent is None or not th.is_grad_enabled():
            with th.no_grad():
                self._run(inputs, out, col_ranges)
            return out
        owners = [l.weight for l, _ in self.mlps[grad_agent].layers]
        return _ActorGroupFn.apply(out, self, list(inputs), list(col_ranges), grad_agent, *owners)


class _SquashedGaussianFn(th.autograd.Function):
    """`params` is either (mean, log_std_raw) as two [B, A] tensors or ONE merged-head GEMM output [B, 2A]."""

    @staticmethod
    def forward(ctx, params, eps, act_dim: int):
        mean, ls = params[:, :act_dim], params[:, act_dim:]
        action = th.empty(params.shape[0], act_dim, dtype=params.dtype, device=params.device)
        logp = th.empty(params.shape[0], dtype=params.dtype, device=params.device)
        hip_ops.squashed_gaussian_fwd(mean, ls, eps, action, logp)
        ctx.save_for_backward(action, params, eps)
        ctx.act_dim = act_dim
        ctx.set_materialize_grads(False)
        return action, logp

    @staticmethod
    def backward(ctx, g_action, g_logp):
        action, params, eps = ctx.saved_tensors
        a = ctx.act_dim
        g_params = th.empty_like(params)
        if g_action is not None and g_action.stride(1) != 1:
            g_action = g_action.contiguous()
        hip_ops.squashed_gaussian_bwd(g_action, None if g_logp is None else g_logp.contiguous(), action, params[:, a:], eps,
                                      g_params[:, :a], g_params[:, a:])
        return g_params, None, None


def squashed_gaussian(params: th.Tensor, eps: th.Tensor, act_dim: int, want_logp: bool = True):
    """(action, logp) of SAC's squashed Gaussian (log_std clamp folded in) from the merged [B, 2A] head output
    [mean | log_std_raw]; logp is None when not wanted."""
    eps = eps.contiguous()
    if th.is_grad_enabled() and params.requires_grad:
        return _SquashedGaussianFn.apply(params, eps, act_dim)
    action = th.empty(params.shape[0], act_dim, dtype=params.dtype, device=params.device)
    logp = th.empty(params.shape[0], dtype=params.dtype, device=params.device) if want_logp else None
    hip_ops.squashed_gaussian_fwd(params[:, :act_dim], params[:, act_dim:], eps, action, logp)
    return action, logp


class _StackedLinearFn(th.autograd.Function):
    """G independent Linear layers as ONE batched GEMM: x [G, M, K] (may be a stride-0 expand of a shared input),
    W [G, N, K], b [G, N] are stacked VIEWS of the parameter arena, wgrad / bgrad the same views of the gradient arena."""

    @staticmethod
    def forward(ctx, x, weight, bias, wgrad, bgrad, act: int, train_params: bool, below, grad_is_dz: bool, *owners):
        ctx.shared = x.dim() == 2  # ONE input for all groups (the critics' first layer): kept 2-D so its gradient comes back
        if ctx.shared:             # as the [M, K] sum over groups from a single launch, not as an expand + reduce
            x = x.unsqueeze(0).expand(weight.shape[0], -1, -1)
        y = _linear_fwd(x, weight, bias, act)
        ctx.act, ctx.train_params, ctx.below, ctx.grad_is_dz = act, train_params, below, grad_is_dz
        ctx.save_for_backward(x, weight, y)
        ctx.wgrad, ctx.bgrad = wgrad, bgrad
        ctx.n_owners = len(owners)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gz = _own_grad(ctx, gy, y)
        _param_grads(ctx, gz, x)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.shared and USE_FUSED_LINEAR:
                dx = hip_ops.linear_bwd_input(gz, weight, None, ACT_NONE, sum_groups=True)
            else:
                dx = _input_grad(gz, weight, x, ctx.below)
                if ctx.shared:
                    dx = dx.sum(0)
        return (dx,) + (None,) * (8 + ctx.n_owners)


def stacked_linear(x, weight, bias, wgrad, bgrad, act: int, train_params: bool, owners=(), below=None, grad_is_dz: bool = False):
    """`owners`: the nn.Parameters whose storage `weight` / `bias` alias; passing them makes the output require grad when
    only the parameters do (first layer on replay data)."""
    if not th.is_grad_enabled() or not (x.requires_grad or train_params):
        return _linear_fwd(x if x.dim() == 3 else x.unsqueeze(0).expand(weight.shape[0], -1, -1), weight, bias, act)
    return _StackedLinearFn.apply(x, weight, bias, wgrad, bgrad, act, train_params, below if x.requires_grad else None, grad_is_dz,
                                  *(owners if train_params else ()))


_ZERO_BIAS: dict = {}

# A loss whose gradient w.r.t. the twin Q values is a per-row function of the batch (the TD critic loss, SAC's actor loss) can ride
# in the FIRST launch of the backward it roots (cstr_hidden_head_bwd_root_f32) instead of being a launch of its own:
# `set_loss_root(root)` right before the backward call; the twin hidden-head backward that runs next consumes it and ignores the
# values of the gradient tensor it is handed.
_pending_root: Optional[dict] = None


def set_loss_root(root: Optional[dict]) -> None:
    global _pending_root
    _pending_root = root


def loss_root_pending() -> bool:
    return _pending_root is not None


@contextlib.contextmanager
def loss_root(root: Optional[dict]):
    """`set_loss_root(root)` for the backward call inside the block; on leaving, the root must have been consumed by that backward
    (else the graph did not start with the twin hidden-head launch: a programming error) and is cleared in any case, so an
    exception inside the block cannot leave a stale root for an unrelated backward."""
    set_loss_root(root)
    try:
        yield
        if root is not None and _pending_root is not None:
            raise RuntimeError("loss root not consumed: the backward did not pass through the twin hidden-head launch")
    finally:
        set_loss_root(None)


LOSS_ROOT_MAX_ROWS = 1024  # cstr_hidden_head_bwd_root_f32 keeps the per-row gradients in LDS


def _take_root(groups: int) -> Optional[dict]:
    global _pending_root
    if _pending_root is None or groups != (1 if _pending_root["mode"] == "neg_mean" else 2):
        return None
    root, _pending_root = _pending_root, None
    return root


def loss_root_supported(critic: "FastTwinCritic") -> bool:
    """The critic's backward starts with the twin hidden-head launch: stacked, two Q networks, scalar head."""
    st = critic.stack
    return (USE_FUSED_LINEAR and USE_LOSS_ROOT and st is not None and len(st) >= 2 and st[0][0].shape[0] == 2 and st[-1][0].shape[1] == 1
            and critic.acts[-1] == ACT_NONE)


def _head_bwd(gq, y, act, w2, dz, gb1, gw2, gb2) -> None:
    root = _take_root(y.shape[0] if y.dim() == 3 else 1)
    if root is not None:
        hip_ops.hidden_head_bwd_root(root, y, act, w2, dz, gb1, gw2, gb2)
    else:
        hip_ops.hidden_head_bwd(gq.contiguous(), y, act, w2, dz, gb1, gw2, gb2)


def _hidden_gemm(x: th.Tensor, w1: th.Tensor) -> th.Tensor:
    """z = x @ w1^T of a Q network's last hidden layer (its bias + activation ride in the head kernel): at batch size the f32-MFMA
    Linear kernel with a zero bias (x . w + 0 is exact) instead of the rocBLAS batched GEMM -- same launch count, no Tensile
    dispatch gap behind it (tools/graph_timeline.sh: 6.7 us start-to-next-start for the 4.6 us Cijk kernel)."""
    if _fused_linear_ok(x):
        n = w1.shape[-2]
        key = (w1.shape[0] if w1.dim() == 3 else 0, n, x.device)
        zb = _ZERO_BIAS.get(key)
        if zb is None:
            zb = _ZERO_BIAS[key] = th.zeros((key[0], n) if key[0] else (n,), dtype=x.dtype, device=x.device)
        return hip_ops.linear_act_fwd(x, w1, zb, ACT_NONE)
    return th.bmm(x, w1.transpose(1, 2)) if x.dim() == 3 else th.mm(x, w1.t())


class _HiddenHeadFn(th.autograd.Function):
    """q = Linear_2(act(Linear_1(x))) with out_features(Linear_2) == 1, plain ([M, K]) or stacked ([G, M, K]) operands.
    w1 [.., N, K], b1 [.., N], w2 [.., 1, N], b2 [.., 1] and their gradient views (None when frozen)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, grads, act: int, train_params: bool, below, *owners):
        batched = x.dim() == 3
        z = _hidden_gemm(x, w1)
        q = th.empty(*z.shape[:-1], 1, dtype=z.dtype, device=z.device)
        hip_ops.hidden_head_fwd_(z, b1, act, w2, b2, q)  # z now holds y = act(z + b1)
        ctx.act, ctx.train_params, ctx.batched, ctx.grads, ctx.n_owners = act, train_params, batched, grads, len(owners)
        ctx.below = below
        ctx.save_for_backward(x, w1, z, w2)
        return q

    @staticmethod
    def backward(ctx, gq):
        x, w1, y, w2 = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = ctx.grads if ctx.train_params else (None, None, None, None)
        dz = th.empty_like(y)
        _head_bwd(gq, y, ctx.act, w2, dz, gb1, gw2, gb2)
        if ctx.train_params:
            if USE_FUSED_LINEAR:
                _weight_grad(dz, x, gw1, None)  # gb1 came out of the head kernel
            else:
                th.bmm(dz.transpose(1, 2), x, out=gw1) if ctx.batched else th.mm(dz.t(), x, out=gw1)
        dx = _input_grad(dz, w1, x, ctx.below) if ctx.needs_input_grad[0] else None
        return (dx,) + (None,) * (8 + ctx.n_owners)


def hidden_head(x, w1, b1, w2, b2, grads, act: int, train_params: bool, owners=(), below=None):
    """The last two layers of a Q network in one GEMM + one launch. `grads` = (w1.grad, b1.grad, w2.grad, b2.grad) views of
    the gradient arena (or None when the parameters are frozen)."""
    if th.is_grad_enabled() and (x.requires_grad or train_params):
        return _HiddenHeadFn.apply(x, w1, b1, w2, b2, grads, act, train_params, below if x.requires_grad else None,
                                   *(owners if train_params else ()))
    z = _hidden_gemm(x, w1)
    q = th.empty(*z.shape[:-1], 1, dtype=z.dtype, device=z.device)
    return hip_ops.hidden_head_fwd_(z, b1, act, w2, b2, q)


def _weight_shadow(optimizer, weight: nn.Parameter) -> Optional[th.Tensor]:
    """The tile-major copy of `weight` that `optimizer` (a FlatAdam over the arena holding it) keeps current with every step
    (arena.FlatAdam.add_weight_shadow), or None. Changes made by torch since the last look (load_state_dict, ...) are caught
    through the parameter's version counter; OffPolicyAlgorithm._setup_learn refreshes it before captured graphs replay."""
    from core.common.arena import FlatAdam

    if not isinstance(optimizer, FlatAdam) or id(weight) not in optimizer.arena.offset_of or weight.shape[1] % 4:
        return None
    if optimizer.shadow is not None and optimizer.shadow[4] is not weight:
        return None
    shadow = optimizer.add_weight_shadow(weight)
    optimizer.refresh_shadow(force=False)
    return shadow


class FastSacActor:
    """core/sac/policies.py:147-175 on the fused path. With `head` = (stacked [2, A, H] weight view, [2, A] bias view and
    their gradient views) the mu and log_std heads are ONE GEMM with N = 2A."""

    def __init__(self, actor, head=None):
        self.actor = actor
        self.latent = FastMLP(actor.latent_pi)
        self.mu, self.log_std = actor.mu, actor.log_std
        self.act_dim = actor.mu.weight.shape[0]
        self.head = head
        self.rng_ctl = None  # in-kernel Philox stream of the sampling head, seeded from torch's seed at first use
        self.deferred_rng = None  # (rng_ctl, rows) of a whole-network launch that left the stream offset to its caller (`defer_rng`)
        if head is not None:
            w, wg, b, bg = head
            a2 = 2 * self.act_dim
            self._hw, self._hb = w.view(a2, -1), b.view(a2)
            self._hwg, self._hbg = wg.view(a2, -1), bg.view(a2)

    def seed_rng(self, seed: int) -> None:
        """Restart the sampling head's in-kernel Philox stream (keeps the tensor: captured graphs hold its address)."""
        if self.rng_ctl is not None:
            self.rng_ctl.copy_(hip_ops.new_rng_ctl(seed, self.rng_ctl.device))

    def _whole_net_ok(self, obs: th.Tensor, train_params: bool) -> bool:
        """No gradient wanted and the network is create_mlp(obs, ., [H1, H2]) with one activation: cstr_policy_rows_fwd_f32."""
        if not USE_FUSED_LINEAR or (th.is_grad_enabled() and (train_params or obs.requires_grad)):
            return False
        if obs.shape[0] <= WHOLE_NET_MIN_ROWS:  # batch-sized passes: the per-layer kernels spread over more CUs (tools/policy_probe.py)
            return False
        layers = self.latent.layers
        if len(layers) != 2 or layers[0][1] != layers[1][1] or obs.dim() != 2 or obs.stride(1) != 1:
            return False
        (l1, _), (l2, _) = layers
        return (hip_ops.policy_rows_supported(l1.in_features, l1.out_features, l2.out_features, 2 * self.act_dim)
                and l1.weight.is_contiguous() and l2.weight.is_contiguous() and l2.weight.data_ptr() % 16 == 0
                and self._hw.is_contiguous() and self._hw.data_ptr() % 16 == 0)

    def rollout_operands(self, obs: th.Tensor) -> Optional[dict]:
        """The sampling actor as operands of hip_ops.rollout_step (Philox noise from this actor's stream), or None."""
        if self.head is None or self.act_dim > hip_ops.nv.MAX_HEAD_ACT or self.actor.action_dist.eps_queue:
            return None
        with th.no_grad():
            if not self._whole_net_ok(obs, train_params=False):
                return None
        (l1, act), (l2, _) = self.latent.layers
        swz = _weight_shadow(getattr(self.actor, "optimizer", None), l2.weight)
        if not hip_ops.rollout_step_supported(l1.in_features, l1.out_features, l2.out_features, 2 * self.act_dim, swz is not None):
            return None
        if obs.stride(0) % 4 or obs.data_ptr() % 16 or l1.weight.data_ptr() % 16:
            return None
        if self.rng_ctl is None:
            self.rng_ctl = hip_ops.new_rng_ctl(th.initial_seed(), obs.device)
        return dict(weights=(l1.weight, l1.bias, l2.weight, l2.bias, self._hw, self._hb), act=act, head=0, out_act=ACT_NONE, w2_swz=swz,
                    rng_ctl=self.rng_ctl)

    def pair_supported(self, pb) -> bool:
        """`action_log_prob_pair` applies: fused Linear kernels, two hidden layers with one activation, merged head, a packed
        batch whose x_pi / x_next are the halves of one buffer, and zero or two teacher-forced noise tensors queued."""
        layers = self.latent.layers
        q = self.actor.action_dist.eps_queue
        return (USE_FUSED_LINEAR and USE_ACTOR_PAIR and self.head is not None and self.act_dim <= hip_ops.nv.MAX_HEAD_ACT and len(layers) == 2
                and layers[0][1] == layers[1][1] and getattr(pb, "x_pn", None) is not None and 2 * pb.x_pi.shape[0] <= 1024
                and layers[1][0].out_features % 4 == 0 and len(q) in (0, 2) and self._hw.is_contiguous()
                and all(lin.weight.grad is not None and lin.bias.grad is not None for lin, _ in layers))

    def action_log_prob_pair(self, pb, gather=None):
        """(x_pi, log pi(a|obs)) with gradients and (x_next, log pi(a'|next_obs)) without, from ONE 2B-row pass (_ActorPairFn).
        `gather` = ReplayBuffer.take_predrawn(pb): the batch has not been gathered yet -- the pass's first layer reads the sampled
        rows from the ring itself and writes `pb` for the launches behind it (hip_ops.linear_act_fwd_gather)."""
        dist = self.actor.action_dist
        b, a = pb.x_pi.shape[0], self.act_dim
        eps2 = None
        if dist.eps_queue:  # teacher-forced draws (tests): the pi(obs) tensor was queued first
            eps2 = th.cat((dist.draw_eps((b, a), pb.x_pn.device), dist.draw_eps((b, a), pb.x_pn.device)), dim=0).contiguous()
        elif self.rng_ctl is None:
            self.rng_ctl = hip_ops.new_rng_ctl(th.initial_seed(), pb.x_pn.device)
        (l1, act), (l2, _) = self.latent.layers
        grads = (l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad, self._hwg, self._hbg)
        x2 = pb.x_pn[:, :pb.obs_dim]
        return _ActorPairFn.apply(x2, l1.weight, l1.bias, l2.weight, l2.bias, self._hw, self._hb, grads, act, eps2, self.rng_ctl,
                                  pb.x_pi.detach(), pb.x_next.detach(), True, gather, l1.weight, l2.weight, self.mu.weight, self.log_std.weight)

    def dist_params(self, obs: th.Tensor, train_params: bool = True) -> th.Tensor:
        """[B, 2A] = [mean | log_std_raw]"""
        h = self.latent(obs, train_params)
        if self.head is None:
            return th.cat((linear(h, self.mu.weight, self.mu.bias, ACT_NONE, train_params),
                           linear(h, self.log_std.weight, self.log_std.bias, ACT_NONE, train_params)), dim=1)
        if not th.is_grad_enabled() or not (h.requires_grad or train_params):
            return hip_ops.bias_act_fwd_(th.mm(h, self._hw.t()), self._hb, ACT_NONE)
        return _MergedHeadFn.apply(h, self._hw, self._hb, self._hwg, self._hbg, train_params, self.mu.weight, self.log_std.weight)

    def action_log_prob(self, obs: th.Tensor, eps: Optional[th.Tensor] = None, train_params: bool = True, want_logp: bool = True,
                        xbuf: Optional[th.Tensor] = None, defer_rng: bool = False):
        """`xbuf`: a critic input buffer [B, D + A] (observation columns already filled): the action is written into its last
        A columns and the BUFFER is returned in place of the action (torch.cat((obs, action)) without the launch).
        `defer_rng`: the rollout's caller advances the Philox offset itself (the fused collect launch that consumes the action
        does it in its last-workgroup epilogue): when the whole-network launch runs it skips its own 256-workgroup ticket and
        `self.deferred_rng` = (rng_ctl, rows) tells the caller what to pass on; otherwise `self.deferred_rng` stays None."""
        dist = self.actor.action_dist
        if eps is None and dist.eps_queue:  # teacher-forced draw (tests)
            eps = dist.draw_eps((obs.shape[0], self.act_dim), obs.device)
        if self.head is None or self.act_dim > hip_ops.nv.MAX_HEAD_ACT:
            params = self.dist_params(obs, train_params)
            if eps is None:
                eps = dist.draw_eps((params.shape[0], self.act_dim), params.device)
            action, logp = squashed_gaussian(params, eps, self.act_dim, want_logp)
            if xbuf is not None:
                return th.cat((xbuf[:, :xbuf.shape[1] - self.act_dim], action), dim=1), logp
            return action, logp
        if self.rng_ctl is None:
            self.rng_ctl = hip_ops.new_rng_ctl(th.initial_seed(), obs.device)
        if self._whole_net_ok(obs, train_params):
            # inference (rollout, target pass): latent net + head + sampling in ONE launch, nothing kept for a backward
            n = obs.shape[0]
            (l1, act), (l2, _) = self.latent.layers
            action = xbuf[:, xbuf.shape[1] - self.act_dim:] if xbuf is not None else th.empty(n, self.act_dim, dtype=obs.dtype, device=obs.device)
            logp = th.empty(n, dtype=obs.dtype, device=obs.device) if want_logp else None
            defer = defer_rng and eps is None
            hip_ops.policy_rows_fwd(obs, l1.weight, l1.bias, l2.weight, l2.bias, self._hw, self._hb, act, 0, ACT_NONE, action,
                                    eps=eps, rng_ctl=None if eps is not None else self.rng_ctl, logp=logp,
                                    w2_swz=_weight_shadow(getattr(self.actor, "optimizer", None), l2.weight), defer_rng_advance=defer)
            if defer:
                self.deferred_rng = (self.rng_ctl, n)
            return (xbuf if xbuf is not None else action), logp
        h = self.latent(obs, train_params, out_grad_is_dz=True)  # this head runs the latent net's last activation gradient
        grad = th.is_grad_enabled() and (h.requires_grad or train_params)
        tp = train_params and grad
        below = self.latent.tail_below(train_params) if h.requires_grad else None
        args = (h, self._hw, self._hb, self._hwg if tp else None, self._hbg if tp else None, eps, self.rng_ctl, xbuf, tp, want_logp,
                below)
        if grad:
            out = _GaussianHeadFn.apply(*args, *((self.mu.weight, self.log_std.weight) if tp else ()))
        else:
            with th.no_grad():
                out = _GaussianHeadFn.apply(*args)
        return (out[0], out[1]) if want_logp else (out[0], None)


class _MergedHeadFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, b, wg, bg, train_params: bool, *owners):
        y = hip_ops.bias_act_fwd_(th.mm(h, w.t()), b, ACT_NONE)
        ctx.save_for_backward(h, w)
        ctx.wg, ctx.bg, ctx.train_params = wg, bg, train_params
        return y

    @staticmethod
    def backward(ctx, gy):
        h, w = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.train_params:
            hip_ops.bias_act_bwd(gy, None, ACT_NONE, gy, ctx.bg)
            th.mm(gy.t(), h, out=ctx.wg)
        dx = th.mm(gy, w) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, None, None, None


class _GaussianHeadFn(th.autograd.Function):
    """Merged (mu | log_std) head GEMM + ONE launch for bias, rsample, tanh and log-prob (noise drawn in the kernel unless
    `eps` is given); backward = ONE launch (d params + bias gradient) + the head's two GEMMs."""

    @staticmethod
    def forward(ctx, h, w, b, wg, bg, eps, rng_ctl, xbuf, train_params: bool, want_logp: bool, below, *owners):
        # at batch size the 2A-output head Linear rides in the sampling kernel; for the 4096-row collect-time batch the
        # rocBLAS GEMM + thread-per-row kernel pair is faster (tools/linear_probe.py: 6.8 vs 9.3 us)
        gemm_inside = (USE_FUSED_LINEAR and h.shape[0] <= 1024 and h.shape[1] % 4 == 0 and h.stride(1) == 1 and h.stride(0) % 4 == 0
                       and h.data_ptr() % 16 == 0 and w.is_contiguous())
        n, a = h.shape[0], w.shape[0] // 2
        params = th.empty(n, 2 * a, dtype=h.dtype, device=h.device) if gemm_inside else th.mm(h, w.t())
        # xbuf: a critic input [B, D + A] whose LAST A columns receive the action (no torch.cat); the whole buffer is the
        # output then, so the critic's input gradient comes back as one tensor and its action columns are read in place
        action = xbuf[:, xbuf.shape[1] - a:] if xbuf is not None else th.empty(n, a, dtype=params.dtype, device=params.device)
        logp = th.empty(n, dtype=params.dtype, device=params.device) if want_logp else None
        if eps is None:
            eps = th.empty(n, a, dtype=params.dtype, device=params.device)
        else:
            rng_ctl = None
        if gemm_inside:  # the 2A-output head Linear is 2A dot products per row: done inside the sampling kernel
            hip_ops.gaussian_head_gemm_fwd(h, w, b, params, eps, rng_ctl, action, logp)
        else:
            hip_ops.gaussian_head_fwd_(params, b, eps, rng_ctl, action, logp)
        ctx.train_params, ctx.wg, ctx.bg, ctx.n_owners, ctx.below = train_params, wg, bg, len(owners), below
        ctx.save_for_backward(h, w, params, eps)
        ctx.action, ctx.a = action.detach(), a
        ctx.set_materialize_grads(False)
        out = action
        if xbuf is not None:
            ctx.mark_dirty(xbuf)
            out = xbuf
        return (out, logp) if want_logp else (out,)

    @staticmethod
    def backward(ctx, g_action, g_logp=None):
        h, w, params, eps = ctx.saved_tensors
        action = ctx.action
        g_params = th.empty_like(params)
        if g_action is not None:
            if g_action.stride(1) != 1:
                g_action = g_action.contiguous()
            g_action = g_action[:, g_action.shape[1] - ctx.a:]  # the action columns of d(loss)/d(critic input)
        g_logp = None if g_logp is None else g_logp.contiguous()
        if USE_FUSED_LINEAR and ctx.below is not None and ctx.needs_input_grad[0] and w.is_contiguous() and h.stride(1) == 1:
            # two launches: (d params, carried through the head's weights and the hidden activation) + (head dW, db)
            dx = th.empty(h.shape[0], w.shape[1], dtype=h.dtype, device=h.device)
            hip_ops.gaussian_head_bwd_input(g_action, g_logp, action, params, eps, w, h, ctx.below[0], g_params, dx)
            if ctx.train_params:
                _weight_grad(g_params, h, ctx.wg, ctx.bg)
            return (dx,) + (None,) * (10 + ctx.n_owners)
        hip_ops.gaussian_head_bwd(g_action, g_logp, action, params, eps, g_params, ctx.bg if ctx.train_params else None)
        if ctx.train_params:
            th.mm(g_params.t(), h, out=ctx.wg)
        dx = _input_grad(g_params, w, h, ctx.below) if ctx.needs_input_grad[0] else None
        return (dx,) + (None,) * (10 + ctx.n_owners)


class _ActorPairFn(th.autograd.Function):
    """SAC's two actor passes of a gradient step -- pi(obs) with gradients (core/sac/sac.py:222) and pi(next_obs) without (:247) --
    as ONE pass over 2B rows: rows [0, B) = obs, rows [B, 2B) = next_obs of one row-strided input, three forward launches instead
    of six. Only the first B rows are kept for the backward, which is the chain's usual three launches on B rows (head backward
    carried through its Linear, layer 2's input gradient, the deferred dW / db launch). The Philox stream positions are those
    of the two separate launches (counter = offset + row, obs rows first). Two hidden layers + the merged head only."""

    @staticmethod
    def forward(ctx, x2, w1, b1, w2, b2, hw, hb, grads, act: int, eps2, rng_ctl, xbuf_pi, xbuf_next, train_params: bool, gather, *owners):
        n2, a = x2.shape[0], hw.shape[0] // 2
        n = n2 // 2
        if gather is not None:  # the sampled rows come straight from the ring; x2 (a view of pb.x_pn) is written by this launch
            ring, idx, rng_advance, pb = gather
            h1 = hip_ops.linear_act_fwd_gather(ring, idx, n, True, w1, b1, act, pb.x_data, pb.x_next, pb.x_pi, pb.samples.dones,
                                               pb.samples.rewards, advance_ring=True, rng_advance=rng_advance)
        else:
            h1 = hip_ops.linear_act_fwd(x2, w1, b1, act)
        h2 = hip_ops.linear_act_fwd(h1, w2, b2, act)
        params = th.empty(n2, 2 * a, dtype=h2.dtype, device=h2.device)
        logp = th.empty(n2, dtype=h2.dtype, device=h2.device)
        if eps2 is None:
            eps2 = th.empty(n2, a, dtype=h2.dtype, device=h2.device)
        else:
            rng_ctl = None
        d = xbuf_pi.shape[1] - a
        # xbuf_pi / xbuf_next are the two halves of ONE [2B, D + A] buffer (PackedBatch.x_pn): one action pointer, one row stride
        action2 = th.as_strided(xbuf_pi, (n2, a), (xbuf_pi.stride(0), 1), xbuf_pi.storage_offset() + d)
        hip_ops.gaussian_head_gemm_fwd(h2, hw, hb, params, eps2, rng_ctl, action2, logp)
        ctx.train_params, ctx.grads, ctx.act, ctx.a, ctx.n_owners = train_params, grads, act, a, len(owners)
        ctx.save_for_backward(x2[:n], h1[:n], h2[:n], params[:n], eps2[:n], w2, hw)
        ctx.action = xbuf_pi.detach()[:, d:]
        ctx.set_materialize_grads(False)
        ctx.mark_dirty(xbuf_pi, xbuf_next)
        lp_pi, lp_next = logp[:n], logp[n:]
        ctx.mark_non_differentiable(xbuf_next, lp_next)
        return xbuf_pi, lp_pi, xbuf_next, lp_next

    @staticmethod
    def backward(ctx, g_x, g_logp, _gxn=None, _glpn=None):
        x, h1, h2, params, eps, w2, hw = ctx.saved_tensors
        g_action = None
        if g_x is not None:
            if g_x.stride(1) != 1:
                g_x = g_x.contiguous()
            g_action = g_x[:, g_x.shape[1] - ctx.a:]
        g_logp = None if g_logp is None else g_logp.contiguous()
        g_params = th.empty_like(params)
        dz2 = th.empty(h2.shape[0], hw.shape[1], dtype=h2.dtype, device=h2.device)
        hip_ops.gaussian_head_bwd_input(g_action, g_logp, ctx.action, params, eps, hw, h2, ctx.act, g_params, dz2)
        dz1 = hip_ops.linear_bwd_input(dz2, w2, h1, ctx.act)
        if ctx.train_params:
            w1g, b1g, w2g, b2g, hwg, hbg = ctx.grads
            _weight_grad(g_params, h2, hwg, hbg)
            _weight_grad(dz2, h1, w2g, b2g)
            _weight_grad(dz1, x, w1g, b1g)
        return (None,) * (15 + ctx.n_owners)


class _TwinPairFn(th.autograd.Function):
    """The two critic passes in front of the critic loss -- Q_k(obs, act) with gradients (core/sac/sac.py:258, core/td3/td3.py:179)
    and Q_k^target(next_obs, a') without (:250, :173) -- as ONE chain over four networks: they are independent until the loss, so
    every layer is one pointer-table launch (cstr_linear_act_fwd_sets_f32: sets 0, 1 = the critic's networks on x_data, sets 2, 3 =
    the target's on x_next) -- three launches instead of six. Only the critic's half is kept for the backward, which is the stacked
    chain's usual launches. Networks = Linear-act-Linear-act-Linear(., 1), twin."""

    @staticmethod
    def chain_sets(x_data, x_next, cs, ts):
        """The chain's buffers and, per layer, its four pointer-table sets (critic nets on x_data, target nets on x_next)."""
        (w1, b1), (w2, b2), (w3, b3) = cs
        (tw1, tb1), (tw2, tb2), (tw3, tb3) = ts
        m, n1, n2 = x_data.shape[0], w1.shape[1], w2.shape[1]
        e = lambda *sh: th.empty(*sh, dtype=x_data.dtype, device=x_data.device)  # noqa: E731
        h1, y2, q = e(4, m, n1), e(4, m, n2), e(4, m, 1)
        xs = (x_data, x_data, x_next, x_next)
        layers = ([(xs[g], (w1, tw1)[g >> 1][g & 1], (b1, tb1)[g >> 1][g & 1], h1[g]) for g in range(4)],
                  [(h1[g], (w2, tw2)[g >> 1][g & 1], (b2, tb2)[g >> 1][g & 1], y2[g]) for g in range(4)],
                  [(y2[g], (w3, tw3)[g >> 1][g & 1], (b3, tb3)[g >> 1][g & 1], q[g]) for g in range(4)])
        return (h1, y2, q), layers

    @staticmethod
    def forward(ctx, x_data, x_next, cs, ts, grads, act: int, pre, *owners):
        """`pre` = (h1, y2, q) already computed by `twin_pair_forward_many` (several chains per launch), or None."""
        w2, w3 = cs[1][0], cs[2][0]
        if pre is None:
            (h1, y2, q), layers = _TwinPairFn.chain_sets(x_data, x_next, cs, ts)
            for li, sets in enumerate(layers):
                hip_ops.linear_act_fwd_sets(sets, act if li < 2 else ACT_NONE)
        else:
            h1, y2, q = pre
        ctx.act, ctx.grads, ctx.n_owners = act, grads, len(owners)
        ctx.save_for_backward(x_data, h1[:2], w2, y2[:2], w3)
        ctx.set_materialize_grads(False)  # no zero-fill launch for the (undefined) gradient of the target's output
        q_c, q_t = q[:2], q[2:]
        ctx.mark_non_differentiable(q_t)
        return q_c, q_t

    @staticmethod
    def backward(ctx, gq, _gt=None):
        x, h1, w2, y2, w3 = ctx.saved_tensors
        (gw1, gb1), (gw2, gb2), (gw3, gb3) = ctx.grads
        dz2 = th.empty_like(y2)
        _head_bwd(gq, y2, ctx.act, w3, dz2, gb2, gw3, gb3)
        _weight_grad(dz2, h1, gw2, None)  # gb2 came out of the head kernel
        dz1 = hip_ops.linear_bwd_input(dz2, w2, h1, ctx.act)
        _weight_grad(dz1, x, gw1, gb1)
        return (None,) * (7 + ctx.n_owners)


class QOut(tuple):
    """Tuple of per-network Q tensors; `.stacked` is the [G, B, 1] batched-GEMM output they are views of (or None):
    backward from the stacked tensor directly (one root) instead of through G select nodes."""
    stacked: Optional[th.Tensor] = None


def backward_q(qs: "QOut", grads: th.Tensor) -> None:
    """autograd.backward for a critic output with d(loss)/dQ given as a [G, B, 1] tensor."""
    with deferred_weight_grads():
        if qs.stacked is not None:
            th.autograd.backward([qs.stacked], [grads[:qs.stacked.shape[0]]])
        else:
            th.autograd.backward(list(qs), [grads[i] for i in range(len(qs))])


class FastTwinCritic:
    """core/common/policies.py:960-987 on the fused path: the n_critics Q networks on cat(obs, action).

    `stack` (per layer: stacked [G, N, K] weight view, [G, N] bias view and their gradient views, from
    `ParamArena.stacked`) evaluates all Q networks as ONE chain of batched GEMMs -- half the launches of the
    per-network chains; without it each network is its own `FastMLP`."""

    def __init__(self, critic, stack=None):
        self.nets = [FastMLP(q) for q in critic.q_networks]
        self.stack = stack
        self.acts = [act for _, act in self.nets[0].layers]
        self.owners = [[net.layers[li][0].weight for net in self.nets] for li in range(len(self.acts))]

    def __call__(self, obs: th.Tensor, actions: th.Tensor, train_params: bool = True, only_first: bool = False):
        return self.forward_input(th.cat([obs, actions], dim=1), train_params, only_first)

    def forward_input(self, x: th.Tensor, train_params: bool = True, only_first: bool = False):
        """Q values for an already assembled critic input [B, features + actions]."""
        if self.stack is None:
            nets = self.nets[:1] if only_first else self.nets
            return QOut(net(x, train_params) for net in nets)
        g = 1 if only_first else len(self.nets)
        h = x  # 2-D: shared by the g groups of the first stacked layer
        stack = self.stack
        scalar_head = len(stack) >= 2 and stack[-1][0].shape[1] == 1 and self.acts[-1] == ACT_NONE
        if train_params and g != stack[0][0].shape[0]:
            raise RuntimeError("stacked critic: parameter gradients need all Q networks in the pass")
        cut = lambda t: None if t is None else t[:g]  # noqa: E731
        plain = stack[:-2] if scalar_head else stack
        below = None
        for li, (w, wg, b, bg) in enumerate(plain):
            inner = scalar_head or li < len(plain) - 1  # its consumer is a fused layer that takes over the activation gradient
            h = stacked_linear(h, w[:g], b[:g], cut(wg), cut(bg), self.acts[li], train_params, self.owners[li][:g], below, inner)
            below = (self.acts[li], None)
        if scalar_head:
            if h.dim() == 2:  # no plain layer in front: the head pair itself reads the shared input
                h = h.unsqueeze(0).expand(g, -1, -1)
            (w1, wg1, b1, bg1), (w2, wg2, b2, bg2) = stack[-2:]
            grads = (cut(wg1), cut(bg1), cut(wg2), cut(bg2)) if train_params else None
            h = hidden_head(h, w1[:g], b1[:g], w2[:g], b2[:g], grads, self.acts[-2], train_params,
                            self.owners[-2][:g] + self.owners[-1][:g], below)
        out = QOut(h[i] for i in range(g))
        out.stacked = h
        return out


def twin_pair_supported(critic: "FastTwinCritic", target: "FastTwinCritic") -> bool:
    """`twin_pair_forward` applies: fused Linear kernels, both stacks present, two Q networks of Linear-act-Linear-act-Linear(., 1)."""
    cs, ts = critic.stack, target.stack
    return (USE_FUSED_LINEAR and USE_TWIN_PAIR and cs is not None and ts is not None and len(cs) == 3 and len(ts) == 3 and cs[0][0].shape[0] == 2
            and ts[0][0].shape[0] == 2 and cs[-1][0].shape[1] == 1 and critic.acts[-1] == ACT_NONE and critic.acts[0] == critic.acts[1]
            and all(wg is not None and bg is not None for _, wg, _, bg in cs) and 4 <= hip_ops.nv.MAX_LINEAR_SETS)


def twin_pair_forward(critic: "FastTwinCritic", target: "FastTwinCritic", x_data: th.Tensor, x_next: th.Tensor):
    """(QOut of the critic on x_data, with gradients; (q1_target, q2_target) on x_next, without) from ONE four-network chain."""
    cs = [(w, b) for w, _, b, _ in critic.stack]
    ts = [(w.detach(), b.detach()) for w, _, b, _ in target.stack]
    grads = [(wg, bg) for _, wg, _, bg in critic.stack]
    owners = [p for layer in critic.owners for p in layer]
    q_c, q_t = _TwinPairFn.apply(x_data, x_next.detach(), cs, ts, grads, critic.acts[0], None, *owners)
    out = QOut(q_c[i] for i in range(2))
    out.stacked = q_c
    return out, (q_t[0], q_t[1])


def twin_pair_forward_many(critics, targets, x_data: th.Tensor, x_next: th.Tensor) -> list:
    """`twin_pair_forward` for SEVERAL (critic, target) pairs that read the same inputs and do not depend on each other (MADDPG's
    per-agent centralised critics on a step without a policy update, core/maddpg/maddpg.py:146-164): every layer of all pairs in
    ceil(4 pairs / MAX_LINEAR_SETS) pointer-table launches instead of one per pair. Returns [(QOut, (q1_target, q2_target))]."""
    x_next = x_next.detach()
    chains = []
    for critic, target in zip(critics, targets):
        cs = [(w, b) for w, _, b, _ in critic.stack]
        ts = [(w.detach(), b.detach()) for w, _, b, _ in target.stack]
        chains.append((cs, ts) + _TwinPairFn.chain_sets(x_data, x_next, cs, ts))
    cap = hip_ops.nv.MAX_LINEAR_SETS
    for li in range(3):
        sets = [st for _, _, _, layers in chains for st in layers[li]]
        for i in range(0, len(sets), cap):
            hip_ops.linear_act_fwd_sets(sets[i:i + cap], critics[0].acts[0] if li < 2 else ACT_NONE)
    outs = []
    for (cs, ts, pre, _), critic in zip(chains, critics):
        grads = [(wg, bg) for _, wg, _, bg in critic.stack]
        owners = [p for layer in critic.owners for p in layer]
        q_c, q_t = _TwinPairFn.apply(x_data, x_next, cs, ts, grads, critic.acts[0], pre, *owners)
        out = QOut(q_c[i] for i in range(2))
        out.stacked = q_c
        outs.append((out, (q_t[0], q_t[1])))
    return outs


def twin_groups(q_networks) -> list:
    """Arena groups that make `FastTwinCritic`'s stacked views possible: for every layer, the weights of all Q networks
    back to back, then their biases. Empty for a single Q network or non-identical architectures."""
    seqs = [[m for m in q if isinstance(m, nn.Linear)] for q in q_networks]
    if len(seqs) < 2 or len({tuple((l.in_features, l.out_features) for l in sq) for sq in seqs}) != 1:
        return []
    groups = []
    for li in range(len(seqs[0])):
        groups.append([sq[li].weight for sq in seqs])
        groups.append([sq[li].bias for sq in seqs])
    return groups


def twin_stack(arena, first_group: int = 0, n_layers: Optional[int] = None):
    """[(W [G,N,K], W.grad view | None, b [G,N], b.grad view | None)] per layer from an arena built with `twin_groups`
    (`first_group` / `n_layers`: the slice of groups of one agent when several critics share an arena)."""
    if n_layers is None:
        n_layers = len(arena.group_spans) // 2
    out = []
    for li in range(n_layers):
        w, wg = arena.stacked(first_group + 2 * li)
        b, bg = arena.stacked(first_group + 2 * li + 1)
        out.append((w, wg, b.view(b.shape[0], -1), None if bg is None else bg.view(bg.shape[0], -1)))
    return out or None
