"""Fused optimizer side of a training iteration (row N3): multi-tensor gradient-norm clipping and AdamW in HIP.

Drop-in for the two lines of the reference loop (Utils/training.py:195-196)

    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)   ->  hdmoe_hip.optim.clip_grad_norm_(model.parameters(), 1.0)
    optimizer.step()                                           ->  FusedAdamW(...).step()

`FusedAdamW` subclasses ``torch.optim.Optimizer`` (param_groups, ``state_dict()`` in torch.optim.AdamW's layout -- so the
reference's ``save_checkpoint`` output stays loadable by either implementation -- and LR schedulers keep working).  The whole
update is one launch over a (tensor, chunk) table; ``step(clip=(params, max_norm))`` additionally fuses the clip coefficient
into the update so norm -> clip -> update never syncs with the host.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import numpy as np
import torch

from ._lib import call, lib

_DESC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("step", "<u8"), ("use", "<u8"), ("numel", "<i8"), ("group", "<i4"),
                  ("pad0", "<i4")])
_CHUNK = 4096


class _Table:
    """Device-side (tensor, chunk) table over a list of parameters that currently have gradients."""

    def __init__(self, entries, steps=None, uses=None):
        """entries: list of (param, grad, m or None, v or None, group); steps: per-entry device float step counters; uses: per-entry device
        address of a float "received a gradient this step" flag (0 = always)."""
        assert lib().hdmoe_opt_desc_bytes() == _DESC.itemsize
        dev = entries[0][0].device
        descs = np.zeros(len(entries), dtype=_DESC)
        chunks = []
        for i, (p, g, m, v, grp) in enumerate(entries):
            d = descs[i]
            d["p"], d["g"] = p.data_ptr(), g.data_ptr()
            d["m"], d["v"] = (0 if m is None else m.data_ptr()), (0 if v is None else v.data_ptr())
            d["step"] = 0 if steps is None else steps[i].data_ptr()
            d["use"] = 0 if uses is None else uses[i]
            d["numel"], d["group"] = p.numel(), grp
            chunks.extend((i, c) for c in range((p.numel() + _CHUNK - 1) // _CHUNK))
        self.descs = torch.from_numpy(descs.view(np.uint8).copy()).to(dev)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)
        self.n = len(chunks)
        self.ntensors = len(entries)
        self.ws = torch.empty(max(self.n, 1), dtype=torch.float32, device=dev)          # per-chunk partial sums of the deterministic norm
        self.sig = tuple((p.data_ptr(), g.data_ptr()) for p, g, _, _, _ in entries) + tuple(uses or ())


_norm_tables = {}


def _grad_entries(params):
    out = []
    for p in params:
        if p.grad is not None:
            if p.dtype != torch.float32 or not p.grad.is_contiguous():
                raise TypeError("hdmoe_hip.optim handles contiguous float32 parameters/gradients")
            out.append((p, p.grad, None, None, 0))
    return out


def grad_norm_sq(params: Iterable[torch.nn.Parameter], key=None) -> Tuple[torch.Tensor, Optional[_Table]]:
    ents = _grad_entries(list(params))
    dev = ents[0][0].device if ents else torch.device("cuda")
    out = torch.empty(1, dtype=torch.float32, device=dev)
    if not ents:
        out.zero_()
        return out, None
    sig = tuple((p.data_ptr(), g.data_ptr()) for p, g, _, _, _ in ents)
    tab = _norm_tables.get(key) if key is not None else None
    if tab is None or tab.sig != sig:
        tab = _Table(ents)
        if key is not None:
            _norm_tables[key] = tab
    call("hdmoe_mt_sumsq", out, tab.descs, tab.chunks, tab.n, tab.ws)
    return out, tab


def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) (L2): scales the gradients in place, returns the total norm.
    Two launches, no host sync."""
    params = list(parameters) if not isinstance(parameters, torch.Tensor) else [parameters]
    ss, tab = grad_norm_sq(params, key=("clip", id(params[0]) if params else 0, len(params)))
    if tab is not None:
        call("hdmoe_mt_clip_scale", tab.descs, tab.chunks, tab.n, ss, float(max_norm))
    return _sqrt_scalar(ss)


def _sqrt_scalar(ss: torch.Tensor) -> torch.Tensor:
    # the returned total norm is only reported/logged; a 1-element sqrt on the device keeps the call sync-free
    return torch.sqrt(ss)[0]


class FusedAdamW(torch.optim.Optimizer):
    """AdamW over all parameters in two launches (global-norm clip + update), ``torch.optim.AdamW``'s state_dict layout.

    Every tensor keeps its own step counter (``state[p]["step"]``, a device float as in torch's capturable mode) and the bias corrections
    are computed from it on the device.  An expert that received no sample in a step has ``.grad is None`` in the reference
    (`if not mask.any(): continue`, models/model_config1.py:26-29) and torch.optim.AdamW then skips its tensors: no weight decay, no
    moment decay, no step increment (Utils/training.py:195-197).  Here such an expert's gradient is an exact zero in its flat-bucket
    view instead, so the skip is driven by the dispatch plan: ``track_expert_usage`` registers the expert lists, the model's forward
    leaves the per-expert routed-row counts of the step in ``<ModuleList>._hdmoe_usage`` (device floats; averaged over the ranks
    with the gradients, hdmoe_hip/dp.py), and the update kernel skips a tensor whose expert's count is 0."""
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdamW supports up to 8 parameter groups")
        if len({tuple(g["betas"]) for g in self.param_groups}) > 1 or len({g["eps"] for g in self.param_groups}) > 1:
            raise ValueError("FusedAdamW needs the same betas/eps in every group (lr and weight_decay may differ)")
        self._table = None
        self._usage = {}                                     # id(param) -> (expert ModuleList, expert index)

    def track_expert_usage(self, expert_lists) -> None:
        """``expert_lists``: ModuleLists of experts (HDMOEM.Unet_experts / VIT_experts).  Their parameters are updated only in steps in
        which the expert received at least one sample (on any rank)."""
        for lst in expert_lists:
            for e, expert in enumerate(lst):
                for p in expert.parameters():
                    self._usage[id(p)] = (lst, e)
        self._table = None

    def _use_ptr(self, p) -> int:
        src = self._usage.get(id(p))
        if src is None:
            return 0
        u = getattr(src[0], "_hdmoe_usage", None)            # written by the model's forward (models/_assembly.py _dispatch_nhwc)
        return 0 if u is None else u.data_ptr() + 4 * src[1]

    def zero_grad(self, set_to_none: bool = False):
        """Zero the gradients IN PLACE.  torch's default (set_to_none=True) would detach the flat DP bucket views / the weight bank's
        gradient buffers that the backward kernels accumulate into (the all-reduce would then run on stale buffers), so it is
        only honoured when asked for explicitly."""
        if set_to_none:
            return super().zero_grad(set_to_none=True)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.grad.zero_()

    def _ensure_state(self):
        ents = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st["step"].device != p.device or st["step"].dtype != torch.float32:   # a checkpoint written by torch.optim.AdamW (host counters)
                    st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
                ents.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], gi))
        uses = tuple(self._use_ptr(p) for p, *_ in ents)
        sig = tuple((p.data_ptr(), g.data_ptr()) for p, g, _, _, _ in ents) + uses
        if self._table is None or self._table.sig != sig:
            self._table = _Table(ents, [self.state[p]["step"] for p, *_ in ents], uses) if ents else None
        return ents

    @torch.no_grad()
    def step(self, closure=None, clip: Optional[Tuple[Iterable[torch.nn.Parameter], float]] = None):
        """One AdamW update.  ``clip=(parameters, max_norm)`` fuses clip_grad_norm_ over `parameters` into the update
        (gradients themselves are left unscaled)."""
        loss = closure() if closure is not None else None
        ents = self._ensure_state()
        if not ents:
            return loss
        ss, max_norm = None, 0.0
        if clip is not None:
            ss, _ = grad_norm_sq(clip[0], key=("step", id(self)))
            max_norm = float(clip[1])
        g0 = self.param_groups[0]
        from . import bank as _bank
        _bank.note_weights_changed()                          # (the kernel writes the parameters through raw pointers: Tensor._version stays)
        call("hdmoe_mt_adamw", self._table.descs, self._table.chunks, self._table.n, self._table.ntensors, ss, max_norm,
             [g["lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups], len(self.param_groups),
             g0["betas"][0], g0["betas"][1], g0["eps"])
        # the routed-row counters of the tracked expert lists accumulate over the forwards of a step (models/_assembly.py): start the next one at zero
        seen = set()
        for lst, _ in self._usage.values():
            if id(lst) not in seen:
                seen.add(id(lst))
                u = getattr(lst, "_hdmoe_usage", None)
                if u is not None:
                    u.zero_()
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = None
