"""Multi-tensor weight bank (host side of csrc/wbank.hip).

A model's MP_Conv weights are tiny (<= 0.8 MB each) but there are ~250 of them; preparing each one per layer costs two
kernel launches in the forward and one plus a gradient accumulation in the backward.  The bank registers every
(weights, dtype, gain, alpha) call site the first time it is used and from the next step on
  * prepares ALL forward / dgrad weight images with one launch at the start of the forward (`begin_step`),
  * lets every wgrad accumulate into one pre-zeroed slab, and
  * turns the slab into parameter gradients with one launch queued at the end of the backward pass (`_finish`),
    accumulating straight into ``p.grad`` (DDP-style bucket views or the bank's own flat buffer).
Layers whose gain is a learnable tensor (Unet_expert.out_gain) and one-off call sites keep the per-layer path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import call, dtype_code, lib

_DESC = np.dtype([("w_raw", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("G", "<u8"), ("dw", "<u8"),
                  ("O", "<i4"), ("I", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("Ipad", "<i4"), ("Opad", "<i4"),
                  ("dtype", "<i4"), ("normalize", "<i4"), ("mutate_ok", "<i4"), ("pad0", "<i4"),
                  ("gain", "<f4"), ("out_scale", "<f4"), ("wf_plane", "<i8"), ("wd_plane", "<i8")])

ACTIVE: Optional["WeightBank"] = None

# Streams that ran part of the current step beside the caller's stream (models/_assembly.py forks the ViT experts onto side
# streams).  Their weight-gradient kernels write the bank's slab without handing a gradient to autograd, so nothing makes
# the engine join them: `_finish` does it explicitly before the one launch that consumes the slab.
FORKED_STREAMS: list = []


def note_forked_streams(streams) -> None:
    for s in streams:
        if s not in FORKED_STREAMS:
            FORKED_STREAMS.append(s)


# Bumped by everything that rewrites parameters behind autograd's back (FusedAdamW's kernel writes through raw pointers and leaves
# Tensor._version alone): part of the "are the eval-mode weight images still current" signature of WeightBank.begin_step.
WEIGHTS_EPOCH = 0


def note_weights_changed() -> None:
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


def invalidate_weights() -> None:
    """Call after writing parameters behind autograd's back -- ``p.data.copy_()`` / ``lerp_`` / ``mul_`` (the usual EMA idiom) or a
    loader that assigns through ``.data`` do not bump ``Tensor._version`` -- so that the next eval-mode forward re-prepares the weight
    images.  (``load_state_dict``, in-place ops on the parameter itself, ``FusedAdamW.step`` and the train-mode forward are tracked
    automatically; ``EDM_Sampler.sample`` additionally compares a device-side checksum of the parameters before replaying its graph.)"""
    note_weights_changed()


DEFER_FINISH = False
STAGE = None                                              # name of the running backward section (graph.Stager)
PENDING: list = []


def finish_stage(stage: str) -> None:
    for b in PENDING:
        b.finish_stage(stage)


def finish_pending() -> None:
    while PENDING:
        PENDING.pop()._finish()


def join_forked_streams() -> None:
    if FORKED_STREAMS and torch.cuda.is_available():
        cur = torch.cuda.current_stream()
        for s in FORKED_STREAMS:
            cur.wait_stream(s)


class Entry:
    __slots__ = ("params", "dtype", "gain", "alpha", "normalize", "O", "I", "khs", "kws", "Ipad", "Opad", "wstride", "wdstride",
                 "gsizes", "wf", "wd", "G", "ready", "used_bwd", "bwd_stage", "rows", "done")

    def __init__(self, params, dtype, gain, alpha, normalize):
        self.params = tuple(params)
        self.dtype, self.gain, self.alpha, self.normalize = dtype, gain, alpha, normalize
        w0 = params[0]
        self.O, self.I = int(w0.shape[0]), int(w0.shape[1])
        self.khs = [int(w.shape[2]) if w.ndim == 4 else 1 for w in params]
        self.kws = [int(w.shape[3]) if w.ndim == 4 else 1 for w in params]
        self.Ipad = (self.I + 15) // 16 * 16
        self.Opad = (self.O + 15) // 16 * 16
        taps = max(a * b for a, b in zip(self.khs, self.kws))
        self.wstride, self.wdstride = taps * self.O * self.Ipad, taps * self.I * self.Opad
        self.gsizes = [a * b * self.O * self.I for a, b in zip(self.khs, self.kws)]
        self.wf = self.wd = None
        self.G: List[torch.Tensor] = []
        self.ready = False
        self.used_bwd = False


class WeightBank:
    def __init__(self, device: torch.device):
        self.device = device
        self.entries: Dict[tuple, Entry] = {}
        self._param_owner: Dict[int, tuple] = {}
        self._dirty = False
        self._built_ptrs: Optional[tuple] = None
        self._descs = self._rows = None
        self._nrows = 0
        self._gflat = self._gradflat = None
        self._cb_queued = False
        self._keep: list = []
        self._prep_sig = None                                # signature of the parameters the current EVAL-mode images were prepared from
        self._prep_sum = None                                # their content checksum (refresh_eval)
        if lib().hdmoe_wbank_desc_bytes() != _DESC.itemsize:
            raise RuntimeError("WBDesc layout mismatch between csrc/wbank.hip and hdmoe_hip/bank.py")

    # ------------------------------------------------------------------------------------------------- registration
    def lookup(self, weights: Sequence[torch.Tensor], dtype: torch.dtype, gain: float, alpha: float, normalize: bool) -> Optional[Entry]:
        key = (tuple(id(w) for w in weights), dtype, float(gain), float(alpha), bool(normalize))
        ent = self.entries.get(key)
        if ent is not None:
            return ent if ent.ready else None
        for w in weights:                                   # a parameter may live in one entry only (in-place train-mode
            if self._param_owner.get(id(w), key) != key:    # renormalisation must have a single writer)
                return None
        if not all(isinstance(w, torch.nn.Parameter) and w.dtype == torch.float32 for w in weights):
            return None
        for w in weights:
            self._param_owner[id(w)] = key
        self.entries[key] = Entry(weights, dtype, gain, alpha, normalize)
        self._dirty = True
        return None                                          # usable from the next begin_step on

    # ------------------------------------------------------------------------------------------------- build
    def _ptr_signature(self):
        sig = []
        for ent in self.entries.values():
            for p in ent.params:
                sig.append(p.data_ptr())
                sig.append(0 if p.grad is None else p.grad.data_ptr())
        return tuple(sig)

    def _alloc(self):
        """(Re)allocate the weight images, the wgrad slab and the bank-owned gradient buffer (entries changed)."""
        dev = self.device
        ents = list(self.entries.values())
        gtotal = sum(sum(e.gsizes) for e in ents)
        self._gflat = torch.zeros(max(gtotal, 1), dtype=torch.float32, device=dev)
        goff = 0
        for ent in ents:
            G = len(ent.params)
            # "split": fp32 layer computed as split bf16 -- bf16 images with a hi and a lo plane (csrc/conv6s.hip)
            wdt, planes = (torch.bfloat16, 2) if ent.dtype == "split" else (ent.dtype, 1)
            ent.wf = torch.zeros(planes * G * ent.wstride, dtype=wdt, device=dev)     # pads stay zero for ever
            ent.wd = torch.zeros(planes * G * ent.wdstride, dtype=wdt, device=dev)
            ent.G = []
            for g in range(G):
                ent.G.append(self._gflat[goff:goff + ent.gsizes[g]])
                goff += ent.gsizes[g]
        params = [p for e in ents for p in e.params]
        self._gradflat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        self._gradviews = {}
        off = 0
        for p in params:
            self._gradviews[id(p)] = self._gradflat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._dirty = False

    def _make_descs(self):
        """Pointer table for the two multi-tensor kernels (parameters' .grad may have been replaced since last step)."""
        ents = list(self.entries.values())
        cleared = False
        for e in ents:
            for p in e.params:
                if p.grad is None:                      # zero_grad(set_to_none=True): hand out our flat views again
                    p.grad = self._gradviews[id(p)]
                    cleared = True
        if cleared:
            self._gradflat.zero_()
        descs = np.zeros(sum(len(e.params) for e in ents), dtype=_DESC)
        rows = []
        di = 0
        for ent in ents:
            esz = 4 if ent.dtype == torch.float32 else 2
            G = len(ent.params)
            for g, p in enumerate(ent.params):
                d = descs[di]
                d["w_raw"], d["dw"], d["G"] = p.data_ptr(), p.grad.data_ptr(), ent.G[g].data_ptr()
                d["wf"] = ent.wf.data_ptr() + g * ent.wstride * esz
                d["wd"] = ent.wd.data_ptr() + g * ent.wdstride * esz
                d["O"], d["I"], d["kh"], d["kw"] = ent.O, ent.I, ent.khs[g], ent.kws[g]
                d["Ipad"], d["Opad"], d["normalize"] = ent.Ipad, ent.Opad, int(ent.normalize)
                d["dtype"] = 2 if ent.dtype == "split" else dtype_code(ent.dtype)
                d["wf_plane"], d["wd_plane"] = G * ent.wstride, G * ent.wdstride
                d["mutate_ok"], d["gain"], d["out_scale"] = 1, ent.gain, ent.alpha
                rows.extend((di, o) for o in range(ent.O))
                di += 1
            ent.rows = rows[-G * ent.O:]                     # this entry's (descriptor, output row) pairs
            ent.ready = True
        self._descs = torch.from_numpy(descs.view(np.uint8).copy()).to(self.device)
        self._rows = torch.tensor(rows, dtype=torch.int32).reshape(-1, 2).contiguous().to(self.device)
        self._nrows = len(rows)
        self._built_ptrs = self._ptr_signature()
        self._stage_rows = {}

    # ------------------------------------------------------------------------------------------------- per step
    def begin_step(self, training: bool):
        """Call at the start of a top-level forward: prepares every registered weight image in one launch."""
        global ACTIVE
        ACTIVE = self
        FORKED_STREAMS.clear()
        from . import ops as _ops
        _ops.zero_pool_reset(self.device)                 # one memset for all of this step's accumulate-into scratch
        _ops.w6_arena_reset(self.device)
        self._w6_pending = []
        if not self.entries:
            return
        if self._dirty:
            self._alloc()
            self._built_ptrs = None
        if self._built_ptrs is None or self._built_ptrs != self._ptr_signature():
            self._make_descs()
            self._prep_sig = None
        grads = torch.is_grad_enabled()
        if grads:
            self._gflat.zero_()
        for e in self.entries.values():
            e.used_bwd = False
            e.done = False
            e.bwd_stage = None
        # Eval-mode weights do not change between forwards (no in-forward re-normalisation): the images are prepared once and re-used
        # until a parameter changes (Tensor._version, or a fused optimizer step).  The sampler runs 2N - 1 evaluations per batch; the
        # prepare launch (~130 us, serial at the head of the step) then runs once instead of 79 times.
        sig = None
        if not training:
            sig = (WEIGHTS_EPOCH,) + tuple(p._version for e in self.entries.values() for p in e.params)
            if sig == self._prep_sig:
                return
        call("hdmoe_wbank_prep", self._descs, self._rows, self._nrows, 1 if training else 0)
        self._prep_sig = sig                                 # (None after a train-mode prepare: it mutates the stored weights)
        if training:
            self._prep_sum = None

    def invalidate(self):
        """Forget the eval-mode weight images: the next forward prepares them again (see `invalidate_weights`)."""
        self._prep_sig = None
        self._prep_sum = None

    def _content_sum(self) -> float:
        """Sum of squares of every registered parameter (one multi-tensor launch + one 4-byte read): catches writes that neither bump
        Tensor._version nor WEIGHTS_EPOCH (`p.data.copy_`, a loader assigning through .data).  Host-synchronous: used where a sync
        is cheap -- once per EDM_Sampler.sample(), not per forward."""
        from . import optim as _optim
        params = [p for e in self.entries.values() for p in e.params]
        ents = [(p, p, None, None, 0) for p in params]
        tab = getattr(self, "_sum_tab", None)
        sig = tuple(p.data_ptr() for p in params)
        if tab is None or tab[0] != sig:
            tab = self._sum_tab = (sig, _optim._Table(ents))
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        call("hdmoe_mt_sumsq", out, tab[1].descs, tab[1].chunks, tab[1].n, tab[1].ws)
        return float(out.item())

    def refresh_eval(self):
        """Bring the eval-mode weight images up to date outside any captured graph (a hipGraph captured while the images were current
        holds no prepare launch: EDM_Sampler calls this once per sample() before replaying).  Compares the parameters' CONTENT too."""
        if self.entries:
            cs = self._content_sum()
            if cs != self._prep_sum:
                self._prep_sig = None
            self.begin_step(False)
            self._prep_sum = cs
            deactivate()

    def note_backward(self, ent: Entry):
        ent.used_bwd = True
        if DEFER_FINISH:                                  # staged backward (graph.Stager): finished section by section
            ent.bwd_stage = STAGE
            if self not in PENDING:
                PENDING.append(self)
            return
        if not self._cb_queued:
            self._cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def defer_w6(self, Gs, seg, ws, dims):
        """A k x k layer left its partial weight-gradient slabs in `ws` (ops._wgrad): summed in _finish / finish_stage."""
        self._w6_pending.append((Gs, seg, ws, dims, STAGE if DEFER_FINISH else None))

    def _reduce_w6(self, stage):
        pend = [it for it in getattr(self, "_w6_pending", []) if stage is None or it[4] == stage]
        if not pend:
            return
        self._w6_pending = [it for it in self._w6_pending if not (stage is None or it[4] == stage)]
        Gflat, segs, wss, dims = [], [], [], []                # one launch per 16 (layer, kernel-size class) items
        for Gs, seg, ws, d, _ in pend:
            Gflat += list(Gs) + [None] * (8 - len(Gs))
            segs.append(seg); wss.append(ws); dims += d
        call("hdmoe_conv_wgrad6_reduce_batch", Gflat, segs, wss, dims, len(pend))

    def finish_stage(self, stage: str):
        """Staged backward: the weight gradients of the entries whose backward ran in section ``stage`` -- their parameters' .grad are
        final afterwards, so that section's gradient bucket can go to RCCL while later sections still run."""
        ents = [e for e in self.entries.values() if e.ready and e.used_bwd and not e.done and e.bwd_stage == stage]
        self._reduce_w6(stage)
        if not ents:
            return
        key = (stage, tuple(id(e) for e in ents))
        rows = self._stage_rows.get(key)
        if rows is None:                                     # built in the eager warm-up steps; static afterwards
            flat = [r for e in ents for r in e.rows]
            rows = self._stage_rows[key] = torch.tensor(flat, dtype=torch.int32).reshape(-1, 2).contiguous().to(self.device)
        call("hdmoe_wbank_bwd", self._descs, rows, rows.shape[0])
        for e in ents:
            e.done = True

    def _finish(self):
        """Runs once at the end of the backward pass: all (remaining) weight gradients in one launch."""
        self._cb_queued = False
        join_forked_streams()
        self._reduce_w6(None)
        ents = [e for e in self.entries.values() if e.ready]
        if any(e.done for e in ents):                        # some sections were finished on their own: only the rest
            # (an entry whose backward did not run has nothing to add, and its parameters may sit in a bucket that is already being
            #  all-reduced beside this launch: the finish kernel's read-modify-write must not touch it)
            rest = [e for e in ents if not e.done and e.used_bwd]
            if rest:
                key = ("rest", tuple(id(e) for e in rest))
                rows = self._stage_rows.get(key)
                if rows is None:
                    flat = [r for e in rest for r in e.rows]
                    rows = self._stage_rows[key] = torch.tensor(flat, dtype=torch.int32).reshape(-1, 2).contiguous().to(self.device)
                call("hdmoe_wbank_bwd", self._descs, rows, rows.shape[0])
            return
        call("hdmoe_wbank_bwd", self._descs, self._rows, self._nrows)


def bank_for(module: torch.nn.Module) -> WeightBank:
    """The bank attached to a top-level module (created on first use)."""
    b = getattr(module, "_hdmoe_bank", None)
    dev = next(module.parameters()).device
    if b is None or b.device != dev:
        b = WeightBank(dev)
        object.__setattr__(module, "_hdmoe_bank", b)
    return b


def deactivate():
    global ACTIVE
    ACTIVE = None
