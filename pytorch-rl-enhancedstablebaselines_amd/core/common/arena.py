"""Flat parameter arenas: the learner-side HBM layout.

Each optimiser group (actor, critic, ...) owns ONE contiguous fp32 buffer for its parameters, one for its
gradients and two for the Adam moments; a target network owns a buffer with the identical layout. The
nn.Module parameters are views into the arena, `.grad`s are views into the gradient buffer, so:

  * `polyak_update` over a whole network is one HIP launch on two flat buffers
    (reference: one mul_ + one add per tensor, core/common/utils.py:478-481),
  * the optimiser step is one HIP launch (reference: torch.optim.Adam, core/common/policies.py:96-117),
  * the data-parallel gradient exchange is one RCCL all-reduce per group on the flat gradient buffer, with
    the 1/world scale folded into the Adam kernel.

Tensors start on 256-byte boundaries inside the arena (GEMM-friendly); the padding stays zero under both
kernels (0 grad, 0 moment -> 0 update).
"""
from typing import Iterable, List, Optional

import torch as th
from torch import nn

from core.common import hip_ops

_ALIGN = 64  # floats (256 B)


class ParamArena:
    """`groups`: lists of same-shaped parameters that must sit back to back WITHOUT padding (the group starts on a
    256-B boundary), so that `stacked(i)` is a plain [G, *shape] view of the arena -- twin critics' layer-l weights as
    one batched-GEMM operand, the actor's mu / log_std heads as one [2A, H] weight. Grouped parameters are placed at
    their first member's position in the parameter order; the layout is a pure function of (shapes, groups), so a
    network and its target network built with the same grouping have identical layouts (polyak is one launch)."""

    def __init__(self, params: Iterable[nn.Parameter], device, with_grad: bool = True, groups: Optional[list] = None,
                 extra_grad: int = 0, grad_storage: Optional[th.Tensor] = None):
        """extra_grad: floats appended to the GRADIENT buffer only (`grad_tail`), not covered by the optimiser: a place
        for another small arena's gradient so that one all-reduce of `grad_full` serves both (SAC: the entropy
        coefficient rides on the critic's collective). grad_storage: adopt such a tail as this arena's gradient buffer."""
        params = list(params)
        self.input_params: List[nn.Parameter] = params  # the caller's order (= the torch optimiser's parameter indices)
        groups = [list(g) for g in (groups or [])]
        in_group = {id(p): gi for gi, g in enumerate(groups) for p in g}
        for g in groups:
            if len({tuple(p.shape) for p in g}) != 1:
                raise ValueError("ParamArena group members must have the same shape")
        ordered, placed = [], set()
        for p in params:  # a group is emitted where its first member appears
            if id(p) in placed:
                continue
            members = groups[in_group[id(p)]] if id(p) in in_group else [p]
            ordered.append(members)
            placed.update(id(q) for q in members)
        self.params: List[nn.Parameter] = [q for members in ordered for q in members]
        self.device = th.device(device)
        self.offsets, self.group_spans, off = [], {}, 0
        for members in ordered:
            start = off
            for q in members:
                self.offsets.append(off)
                off += q.numel()
            if len(members) > 1:
                self.group_spans[in_group[id(members[0])]] = (start, len(members), tuple(members[0].shape))
            off = -(-off // _ALIGN) * _ALIGN
        self.numel = max(off, _ALIGN)
        self.offset_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        self.flat = th.zeros(self.numel, dtype=th.float32, device=self.device)
        self.grad = self.grad_full = self.grad_tail = None
        if grad_storage is not None:
            if grad_storage.numel() < self.numel or not grad_storage.is_contiguous():
                raise ValueError("grad_storage too small")
            self.grad = self.grad_full = grad_storage[:self.numel]
        elif with_grad:
            self.grad_full = th.zeros(self.numel + extra_grad, dtype=th.float32, device=self.device)
            self.grad = self.grad_full[:self.numel]
            self.grad_tail = self.grad_full[self.numel:] if extra_grad else None
        with th.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.detach().to(self.device, th.float32))
                p.data = view
                if with_grad and p.requires_grad:
                    p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def stacked(self, group_index: int):
        """([G, *shape] view of the parameters, same view of the gradient arena or None)."""
        start, g, shape = self.group_spans[group_index]
        n = g * int(th.tensor(shape).prod()) if shape else g
        w = self.flat[start:start + n].view(g, *shape)
        return w, (None if self.grad is None else self.grad[start:start + n].view(g, *shape))

    def zero_grad(self) -> None:
        """One memset; `.grad` views stay attached so autograd keeps accumulating in place."""
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def same_layout(self, other: "ParamArena") -> bool:
        return self.offsets == other.offsets and self.numel == other.numel and \
            [tuple(p.shape) for p in self.params] == [tuple(p.shape) for p in other.params]

    def polyak_from(self, source: "ParamArena", tau: float) -> None:
        """self (target) <- polyak(source, tau): reference core/common/utils.py:457-481, one launch."""
        if not self.same_layout(source):
            raise ValueError("Iterables have different lengths")  # zip_strict's error (utils.py:447)
        with th.cuda.device(self.device):
            hip_ops.polyak(source.flat, self.flat, tau)


class ArenaSlice:
    """A contiguous run of whole tensors inside a `ParamArena` (one agent's parameters inside the arena that holds
    every agent's): same duck type as ParamArena for FlatAdam and the gradient all-reduce."""

    def __init__(self, arena: ParamArena, params: Iterable[nn.Parameter]):
        given = list(params)
        pos = {id(q): i for i, q in enumerate(arena.params)}
        idx = sorted(pos[id(p)] for p in given)  # arena order (grouping may have permuted the caller's order)
        if idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError("ArenaSlice needs consecutive arena tensors")
        params = [arena.params[i] for i in idx]
        start = arena.offsets[idx[0]]
        end = arena.offsets[idx[-1] + 1] if idx[-1] + 1 < len(arena.params) else arena.numel
        self.parent, self.params, self.device = arena, params, arena.device
        self.input_params = given
        self.offsets = [arena.offsets[i] - start for i in idx]
        self.offset_of = {id(p): o for p, o in zip(params, self.offsets)}
        self.numel = end - start
        self.flat = arena.flat[start:end]
        self.grad = arena.grad[start:end]

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FlatAdam:
    """torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay / amsgrad -- the reference's defaults) over one
    `ParamArena`, one HIP launch per step, step counter and learning rate resident in HBM."""

    def __init__(self, arena: ParamArena, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.arena = arena
        self.defaults = dict(lr=lr, betas=betas, eps=eps)
        self.param_groups = [dict(params=arena.params, lr=lr, betas=betas, eps=eps)]
        dev = arena.device
        self.exp_avg = th.zeros_like(arena.flat)
        self.exp_avg_sq = th.zeros_like(arena.flat)
        self.ctl = hip_ops.new_adam_ctl(dev, 0, betas[0], betas[1])
        self.lr_dev = th.tensor([lr], dtype=th.float64, device=dev)
        self._lr_on_device = float(lr)
        self.grad_scale = 1.0  # 1 / world_size after a summing all-reduce
        self.shadow = None  # (tile-major copy, offset in the arena, rows, cols, parameter) of ONE weight matrix, see add_weight_shadow

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad()

    def sync_lr(self) -> None:
        """Push `param_groups[0]["lr"]` (set by update_learning_rate, reference core/common/utils.py:56-66) to HBM.
        Called outside captured regions; a constant schedule never copies."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.lr_dev.fill_(lr)
            self._lr_on_device = lr

    def step(self, closure=None) -> None:
        g = self.param_groups[0]
        with th.cuda.device(self.arena.device):
            if self.shadow is not None:  # the multi-segment kernel is the one that keeps a shadow copy current
                hip_ops.adam_multi([self._segment()])
                return
            hip_ops.adam(self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, self.ctl, self.lr_dev,
                         g["betas"][0], g["betas"][1], g["eps"], self.grad_scale)

    def moments_of(self, param: nn.Parameter):
        """(exp_avg, exp_avg_sq) views of one parameter of this arena (the fused weight-gradient + Adam launch updates them per tile)."""
        o = self.arena.offset_of[id(param)]
        return self.exp_avg[o:o + param.numel()], self.exp_avg_sq[o:o + param.numel()]

    def _segment(self) -> tuple:
        g = self.param_groups[0]
        seg = (self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, self.ctl, self.lr_dev, g["betas"][0], g["betas"][1],
               g["eps"], self.grad_scale)
        return seg if self.shadow is None else seg + (self.shadow[:4],)

    def add_weight_shadow(self, param: nn.Parameter) -> th.Tensor:
        """Keep a tile-major copy (hip_ops.policy_swizzle) of ONE [N, K] weight matrix of this arena for the policy kernel's
        operand loads: every step rewrites it together with the weights; `refresh_shadow()` after anything else changed them."""
        if self.shadow is not None:
            if self.shadow[4] is param:
                return self.shadow[0]
            raise ValueError("one weight shadow per optimiser")
        o = self.arena.offset_of[id(param)]
        n, k = param.shape
        if o % 4 or k % 4:
            raise ValueError("shadowed matrix: offset and width must be multiples of 4")
        with th.cuda.device(self.arena.device):
            self.shadow = (hip_ops.policy_swizzle(param.detach()), o, n, k, param)
        self._shadow_version = param._version
        return self.shadow[0]

    def refresh_shadow(self, force: bool = True) -> None:
        """Re-derive the shadow copy from the weights (after load_state_dict / set_parameters / a broadcast: anything that
        changed them other than step()). force=False: only when torch's version counter of the parameter moved."""
        if self.shadow is None or (not force and self.shadow[4]._version == self._shadow_version):
            return
        with th.cuda.device(self.arena.device):
            hip_ops.policy_swizzle(self.shadow[4].detach(), self.shadow[0])
        self._shadow_version = self.shadow[4]._version

    def step_with(self, *others: "FlatAdam", polyak=None, own_target=None) -> None:
        """This optimiser's step, the others' and -- polyak=(source arena, target arena, tau), or a list of (source flat, target
        flat, tau) runs -- soft target updates of parameters none of them touches, in ONE launch (same arithmetic as the separate
        calls). own_target=(target flat tensor, tau): the soft update of THIS optimiser's parameters' target as well, by the threads
        that have just computed the new values (cstr_adam_seg_t.own_target)."""
        seg = self._segment()
        if own_target is not None:
            if own_target[0].numel() != self.arena.flat.numel():
                raise ValueError("Iterables have different lengths")  # zip_strict's error (utils.py:447)
            seg = seg[:10] + (seg[10] if len(seg) > 10 else None, own_target)
        segs = [seg] + [o._segment() for o in others]
        if polyak is not None:
            if isinstance(polyak, tuple):
                source, target, tau = polyak
                if not target.same_layout(source):
                    raise ValueError("Iterables have different lengths")
                polyak = [(source.flat, target.flat, tau)]
            for source_flat, target_flat, tau in polyak:
                if source_flat.numel():
                    segs.append(("polyak", source_flat, target_flat, tau))
        with th.cuda.device(self.arena.device):
            hip_ops.adam_multi(segs)

    @property
    def step_count(self) -> int:
        return int(self.ctl[0])

    def state_dict(self) -> dict:
        """torch.optim.Adam's state_dict layout (per-parameter `step` / `exp_avg` / `exp_avg_sq`), so checkpoints are
        interchangeable with the reference's `actor.optimizer.pth` etc. (core/common/base_class.py:827-840)."""
        step = float(self.ctl[0])
        state = {}
        for i, p in enumerate(self.arena.input_params):  # torch.optim indexes parameters in the order they were given
            o = self.arena.offset_of[id(p)]
            sl = slice(o, o + p.numel())
            state[i] = {"step": th.tensor(step), "exp_avg": self.exp_avg[sl].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[sl].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {"lr": float(g["lr"]), "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.arena.input_params)))}
        return {"state": state if step > 0 else {}, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        state = sd.get("state", {})
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, p in enumerate(self.arena.input_params):
            o = self.arena.offset_of[id(p)]
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            sl = slice(o, o + p.numel())
            self.exp_avg[sl].view(p.shape).copy_(st["exp_avg"])
            self.exp_avg_sq[sl].view(p.shape).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FlatAdam needs one common step count, checkpoint has {sorted(steps)}")
        g0 = self.param_groups[0]
        hip_ops.set_adam_step(self.ctl, steps.pop() if steps else 0, g0["betas"][0], g0["betas"][1])
        groups = sd.get("param_groups") or [{}]
        if "lr" in groups[0]:
            self.param_groups[0]["lr"] = float(groups[0]["lr"])
        self.sync_lr()


def make_optimizer(module_params: Iterable[nn.Parameter], device, lr: float, optimizer_class=None,
                   optimizer_kwargs: Optional[dict] = None, groups: Optional[list] = None, extra_grad: int = 0):
    """Default (optimizer_class None or torch.optim.Adam with default kwargs) -> arena + FlatAdam. Any other
    optimiser class is honoured with the stock torch implementation on the arena's parameter views."""
    optimizer_kwargs = dict(optimizer_kwargs or {})
    arena = ParamArena(module_params, device, groups=groups, extra_grad=extra_grad)
    if optimizer_class in (None, th.optim.Adam) and set(optimizer_kwargs) <= {"betas", "eps"}:
        return arena, FlatAdam(arena, lr=lr, **optimizer_kwargs)
    return arena, optimizer_class(arena.params, lr=lr, **optimizer_kwargs)
