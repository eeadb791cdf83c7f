"""Autograd-aware wrappers over the HIP kernels (one torch.autograd.Function per fused op).

Layout contract of this module: activation tensors are contiguous channel-last -- spatial ``(N, H, W, C)``,
token ``(N, S, C)`` or matrix ``(M, C)`` -- in float32 or bfloat16.  "Vector path" tensors (per-sample scalars,
``(B, F)`` embeddings, parameters and their gradients) are float32.  PyTorch is used for memory (``torch.empty`` /
``zeros``), views and autograd bookkeeping only; every arithmetic op below is a call into libhdmoe_hip.so.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

from . import bank as _bank
from ._lib import dtype_code_cast  # noqa: F401
from ._lib import call, dtype_code

Tensor = torch.Tensor


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    return t if t.is_contiguous() else t.contiguous()


def _dt(t: Tensor) -> int:
    return dtype_code(t.dtype)


def _f32(t: Tensor) -> Tensor:
    if t.dtype != torch.float32:
        raise TypeError("expected a float32 vector-path tensor")
    return _c(t)


_seed_state = {"seed": 0x5DEECE66D, "ctr": 0}

# Host-side counters of which fused paths a step took ("trunk": fused router trunk forward, "bwd6" / "bwd6s": dgrad + wgrad in one
# launch, "w6_defer": deferred wgrad6 reduction, "blk": fused Unet_block main branch): the parity tests assert that the path they
# pin to the reference is the one the benchmark times.
import collections as _collections
STATS = _collections.Counter()


def manual_seed(seed: int) -> None:
    """Seed of the device counter RNG used for dropout masks and router logit noise."""
    _seed_state["seed"] = int(seed) & 0xFFFFFFFFFFFF
    _seed_state["ctr"] = 0


def _next_seed() -> int:
    _seed_state["ctr"] += 1
    return ((_seed_state["seed"] * 0x9E3779B97F4A7C15) ^ (_seed_state["ctr"] * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF


_step_counters = {}

# Independent, launch-latency-bound branches of the model (the ViT experts) run on side streams beside the main stream's
# large launches; HDMOE_SIDE_STREAMS=0 serialises everything on the caller's stream.
import os as _os
SIDE_STREAMS = _os.environ.get("HDMOE_SIDE_STREAMS", "1") != "0"
_side_pool = {}


def side_streams(device, n: int):
    key = torch.device(device)
    pool = _side_pool.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=key))
    return pool[:n]


def step_counter(device) -> Tensor:
    """Device-resident step counter mixed into every dropout / noise Philox key (see csrc/elementwise.hip: mix_seed)."""
    key = torch.device(device)
    t = _step_counters.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=key)
        _step_counters[key] = t
    return t


def advance_seed(device) -> None:
    """Call once per training step (inside the captured region when the step is a hipGraph)."""
    call("hdmoe_seed_advance", step_counter(device))


# =====================================================================================================
# MP_Conv: weight prep + implicit GEMM conv (+dgrad, wgrad)
# =====================================================================================================
# Optional per-launch timing of the GEMM-shaped kernels (bench.py's roofline leg): when PROFILE is a list, every
# hdmoe_conv_fwd / hdmoe_conv_wgrad launch is bracketed by events on the launch stream and logged with its shape.
PROFILE = None
# PROFILE_FUSED: the timed leg keeps the weight-bank path of the replayed step -- dgrad + wgrad in one launch (bwd6 / bwd6s), the fused
# Unet_block launch (blk6), the fused router trunk, deferred wgrad6 reductions -- and brackets THOSE launches with events
# (bench.py: roofline = the in-step dominant kernel).  False: every layer on its own unfused kernels (the round-1/2 leg).
PROFILE_FUSED = False
_spacer = None


def _prof_ok() -> bool:
    return PROFILE is None or PROFILE_FUSED


def _timed(kind: str, info: dict, name: str, *args):
    if PROFILE is None:
        return call(name, *args)
    # An event pair also spans the time the stream sits idle waiting for the host to enqueue the kernel.  A spacer launch in
    # front keeps the GPU busy while start event, kernel and end event are all enqueued, so the pair brackets the kernel alone.
    global _spacer
    if _spacer is None:
        _spacer = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    _spacer.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = call(name, *args)
    e.record()
    if rc == 0:
        if name == "hdmoe_conv_wgrad6" and "deferred" not in info.get("wgrad_name", ""):
            info = dict(info, wgrad_name="wgrad6_kernel<split> (+ reduce)" if info.get("dtype") == "split_bf16" else "wgrad6_kernel (+ reduce)")
        PROFILE.append((kind, info, s, e))
    return rc


def _conv6_domain(x, Ho, Wo, O, I, Cstore, khs, kws, cphys, split):
    """(kernel name, dtype label) when the launch goes to the conv6 kernels (csrc/conv6.hip / conv6s.hip), else None."""
    if Cstore != O or cphys != I or I % 32 or O % 32 or not (Wo == 16 or Wo % 32 == 0) or Ho < 8:
        return None
    if split:
        return (f"conv6_split_kernel<{2 if O % 64 == 0 else 1}>", "split_bf16") if all(k == 3 for k in khs) else None
    if x.dtype != torch.bfloat16 or any(a != b or a not in (3, 5, 7) for a, b in zip(khs, kws)):
        return None
    return "conv6_bf16_kernel", "bfloat16"


def _conv_info(x, seg, N, Ho, Wo, O, I, Cstore, khs, kws, cphys, split=False):
    """Shape record of one launch + the kernel instantiation hdmoe_conv_fwd will pick (mirrors csrc/conv.hip)."""
    c6 = _conv6_domain(x, Ho, Wo, O, I, Cstore, khs, kws, cphys, split)
    if c6 is not None:
        tname = "float" if x.dtype == torch.float32 else "__bf16"
        return dict(dtype=c6[1], seg=seg, N=N, HW=Ho * Wo, O=O, I=I, taps=[a * b for a, b in zip(khs, kws)], fwd_name=c6[0],
                    wgrad_name=_wgrad_name(tname, O, sorted(set(a * b for a, b in zip(khs, kws))), I, cphys == I))
    esz = x.element_size()
    vec = cphys % (16 // esz) == 0
    tname = "float" if x.dtype == torch.float32 else "__bf16"
    nt = 1 if Cstore <= 32 else 2
    v5ok = I == cphys and Cstore % 4 == 0
    mk, mw = max(khs), max(kws)
    tile2d = (v5ok and vec and Ho >= 8 and Wo > 32 and Wo % 32 == 0 and mk * mw > 1 and (8 + mk - 1) * (32 + mw - 1) * 4 <= 9 * 256
              and mw * 32 * nt <= 576 and 80 * ((8 + mk - 1) * (32 + mw - 1) + mw * 32 * nt) <= 80 * 1024)
    tw = 32 if tile2d else min(Wo, 256)
    th = max(1, min(256 // tw, Ho))
    lds = 80 * ((th + max(khs) - 1) * (tw + max(kws) - 1) + max(kws) * 32 * nt)
    halo = (th + max(khs) - 1) * (tw + max(kws) - 1)
    if Ho * Wo >= 64 and vec and halo * 4 <= (9 if v5ok else 7) * 256 and max(kws) * 32 * nt <= 576:
        tg = min(576 // (max(kws) * 32 * nt), max(khs))
        while tg > 1 and 80 * (halo + tg * max(kws) * 32 * nt) > 64 * 1024:
            tg -= 1
        lds3 = 80 * (halo + tg * max(kws) * 32 * nt)
        if v5ok and lds3 <= 80 * 1024:
            lepi = Cstore % (16 // esz) == 0 and 4 * 32 * (32 * nt + 16 // esz) * esz <= lds3
            fwd_name = f"conv_fwd5_kernel<{tname}, {nt}, {'true' if lepi else 'false'}, {9 if halo * 4 > 7 * 256 else 7}>"
        elif lds3 <= 64 * 1024 and halo * 4 <= 7 * 256:
            fwd_name = f"conv_fwd3_kernel<{tname}, {nt}>"
        else:
            fwd_name = f"conv_fwd2_kernel<{tname}, {nt}, {'true' if vec else 'false'}>"
    elif Ho * Wo >= 64 and lds <= 64 * 1024:
        fwd_name = f"conv_fwd2_kernel<{tname}, {nt}, {'true' if vec else 'false'}>"
    else:
        nb = 1 if Cstore <= 32 else (2 if Cstore <= 64 else 4)
        fwd_name = f"conv_fwd_kernel<{tname}, {nb}, {'true' if vec else 'false'}>"
    return dict(dtype=str(x.dtype).replace("torch.", ""), seg=seg, N=N, HW=Ho * Wo, O=O, I=I, taps=[a * b for a, b in zip(khs, kws)],
                fwd_name=fwd_name, wgrad_name=_wgrad_name(tname, O, sorted(set(a * b for a, b in zip(khs, kws))), I, cphys == I))


def _wgrad_name(tname, O, tap_classes, I=0, plain=False):
    """Kernel instantiation(s) hdmoe_conv_wgrad picks (mirrors csrc/conv.hip: one launch per kernel-size class)."""
    if plain and tap_classes == [1] and O % 32 == 0 and I % 32 == 0:          # pointwise layers: csrc/lwgrad.hip
        return f"lwg_{'f32' if tname == 'float' else 'bf16'}_kernel<{2 if O % 64 == 0 else 1}, {2 if I % 64 == 0 else 1}>"
    names = []
    for taps in tap_classes:
        passes = (taps + 27) // 28 if taps > 28 else 1
        mt = 7 if passes > 1 else (taps if taps < 4 else (taps + 3) // 4)
        ot = 1 if ((passes == 1 and mt > 7) or O <= 32) else 2
        names.append(f"{ot}, {3 if mt <= 3 else (7 if mt <= 7 else 13)}" + (f" x{passes} passes" if passes > 1 else ""))
    if len(names) == 1:
        return f"conv_wgrad2_kernel<{tname}, {names[0]}, true>"
    return f"conv_wgrad2_kernel<{tname}, {{{' | '.join(names)}}}, true> ({len(names)} launches per call)"


def _kernel_hw(w: Tensor):
    if w.ndim == 4:
        return int(w.shape[2]), int(w.shape[3])
    if w.ndim == 2:
        return 1, 1
    raise ValueError(f"weight must be 2-D or 4-D, got {tuple(w.shape)}")


_w6_ws = {}

# Small learnable tensors (norm affines, biases, rel_pos_bias, alpha_txt): when the parameter already owns a gradient buffer (the
# flat DP buckets, or a .grad kept from the previous step) the backward kernels -- which accumulate anyway -- add straight into it
# and hand autograd None.  That is what the weight bank does for the conv weights; per step it removes a zero-fill and an
# AccumulateGrad add per parameter (~300 launches of ~5 us that the host enqueues one by one).  HDMOE_DIRECT_PARAM_GRADS=0: off.
DIRECT_PARAM_GRADS = _os.environ.get("HDMOE_DIRECT_PARAM_GRADS", "1") != "0"


def _direct(p) -> bool:
    return (DIRECT_PARAM_GRADS and p is not None and p.is_leaf and p.requires_grad and p.grad is not None
            and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.shape == p.shape)



class _ZeroPool:
    """Zero-initialised scratch for the backward kernels that ACCUMULATE (atomics) into a fresh buffer: ~50 such buffers per step
    were 50 memset launches.  One persistent buffer per device, cleared by ONE memset at the start of a step (WeightBank.begin_step),
    handed out in 256-byte-aligned slices.  Only for tensors that never become a leaf's .grad (autograd may adopt those)."""

    def __init__(self, device, nbytes=32 << 20):
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.off = 0
        self.high = 0

    def reset(self):
        if self.high:
            self.buf[:(self.high + 255) // 256 * 256].view(torch.float32).zero_()      # (a byte-typed fill runs one byte per thread)
        self.off = self.high = 0

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nb = n * torch.empty((), dtype=dtype).element_size()
        start = (self.off + 255) // 256 * 256
        if start + nb > self.buf.numel():
            return None
        self.off = start + nb
        self.high = max(self.high, self.off)
        return self.buf[start:start + nb].view(dtype).view(*shape)


_zero_pools = {}
ZERO_POOL = _os.environ.get("HDMOE_ZERO_POOL", "1") != "0"


def zero_pool_reset(device) -> None:
    pool = _zero_pools.get(torch.device(device))
    if pool is not None:
        pool.reset()


def _zeros(shape, dtype, device, pool_ok: bool = True) -> Tensor:
    """torch.zeros for accumulate-into scratch / non-leaf gradients, served from the per-step zero pool when possible
    (``pool_ok=False``: the buffer may become a leaf's .grad -- autograd adopts incoming gradients -- and must own its memory)."""
    device = torch.device(device)
    if pool_ok and ZERO_POOL and device.type == "cuda":
        pool = _zero_pools.get(device)
        if pool is None:
            if torch.cuda.is_current_stream_capturing():
                return torch.zeros(shape, dtype=dtype, device=device)
            pool = _zero_pools[device] = _ZeroPool(device)
        t = pool.take(tuple(shape), dtype)
        if t is not None:
            return t
    return torch.zeros(shape, dtype=dtype, device=device)


class _FanoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n, scales):
        ctx.set_materialize_grads(False)
        ctx.scales = scales
        return tuple(x.view(x.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        sc = ctx.scales or (1.0,) * len(gs)
        pairs = [(_c(g), float(s)) for g, s in zip(gs, sc) if g is not None]
        if not pairs:
            return None, None, None
        if len(pairs) == 1 and pairs[0][1] == 1.0:
            return pairs[0][0], None, None
        out = torch.empty_like(pairs[0][0])
        for i in range(0, len(pairs), 15):                    # 16 sources per launch (the running sum takes one slot)
            part = pairs[i:i + 15] if i == 0 else [(out, 1.0)] + pairs[i:i + 15]
            call("hdmoe_sum_n", out, [g for g, _ in part], [s for _, s in part], len(part), out.numel(), _dt(out))
        return out, None, None


def fanout(x: Tensor, n: int, scales=None):
    """n aliases of ``x`` for n consumers: their gradients are summed by ONE kernel in the backward (autograd's own accumulation is
    n - 1 separate add launches).  ``scales``: per-alias factors applied to the incoming gradients inside that sum (a consumer that
    hands back an unscaled gradient, mp_conv(res_grad_raw=True)).  The aliases must not be modified in place."""
    if n <= 1 or not (torch.is_tensor(x) and x.requires_grad):
        return tuple(x for _ in range(max(n, 1)))
    return _FanoutFn.apply(x, int(n), None if scales is None else tuple(float(v) for v in scales))


class _W6Arena:
    """Workspace of the DEFERRED wgrad6 reductions: every k x k layer of the weight bank keeps its partial slabs until the end of the
    backward pass, where one batched launch per 16 layers sums them (WeightBank._finish).  One bump-allocated buffer per device,
    rewound at the start of a step.  It starts at HDMOE_W6_ARENA_MB (default 1024) and is re-sized to the step's high-water mark at the
    next rewind outside a graph capture: a layer that does not fit meanwhile takes the non-deferred path (its own cached workspace)."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(int(_os.environ.get("HDMOE_W6_ARENA_MB", "1024")) << 18, dtype=torch.float32, device=device)
        self.off = 0
        self.want = 0                                        # floats the current step would have needed
        self.captured = False                                # a hipGraph capture handed out slices of the CURRENT buffer
        self._old = []                                       # outgrown buffers that a captured graph may still point into

    def take(self, nfloats):
        self.want = (self.want + 63) // 64 * 64 + nfloats    # advances whether or not the slice fits: the next rewind grows to it
        start = (self.off + 63) // 64 * 64
        if start + nfloats > self.buf.numel():
            if torch.cuda.is_current_stream_capturing():
                # the captured step would bake the slower non-deferred path in: say how to avoid it instead of degrading silently
                import warnings
                warnings.warn(f"hdmoe_hip: the deferred weight-gradient arena ({self.buf.numel() * 4 >> 20} MB) is too small for this step and cannot "
                              f"grow inside a hipGraph capture; run one eager step before capturing or set HDMOE_W6_ARENA_MB >= "
                              f"{(self.want * 5 >> 20) + 1}", RuntimeWarning, stacklevel=3)
            return None
        self.off = start + nfloats
        if torch.cuda.is_current_stream_capturing():
            self.captured = True
        return self.buf[start:start + nfloats]

    def rewind(self):
        if self.want > self.buf.numel() and not torch.cuda.is_current_stream_capturing():
            if self.captured:                                # only a buffer some captured graph points into has to stay alive
                self._old.append(self.buf)
            self.buf = torch.empty(int(self.want * 1.25) // 64 * 64 + 64, dtype=torch.float32, device=self.device)
            self.captured = False
        self.off = self.want = 0

    def release_old(self):
        """Drop the outgrown buffers (call once every graph captured before the last growth has been destroyed)."""
        self._old.clear()


_w6_arenas = {}
W6_DEFER = _os.environ.get("HDMOE_W6_DEFER", "1") != "0"
BWD6 = _os.environ.get("HDMOE_BWD6", "1") != "0"          # dgrad + wgrad of a k x k expert layer in one launch (csrc/bwd6.hip)


def w6_arena_reset(device) -> None:
    a = _w6_arenas.get(torch.device(device))
    if a is not None:
        a.rewind()


def _w6_arena_take(device, nfloats):
    device = torch.device(device)
    a = _w6_arenas.get(device)
    if a is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        a = _w6_arenas[device] = _W6Arena(device)
    return a.take(nfloats)


def _wgrad(info, x, dy, Gs, seg, G, N, H, W, Ho, Wo, I, Cphys, O, ones, khs, kws, pts, split=False, bank=None):
    """Weight gradient of a (grouped) conv into the [tap][O][I] fp32 slabs ``Gs`` (+=).  k x k bf16 layers -- and fp32 layers in
    split-bf16 mode (the router trunks) -- take the atomic-free kernel (csrc/wgrad6.hip) with a cached workspace; everything else
    the general kernel."""
    if (x.dtype == torch.bfloat16 or split) and not ones and Ho == H and Wo == W and Cphys == I:
        from ._lib import lib, _int_array
        import ctypes
        dtc = F32S if split else _dt(x)
        kib = lib().hdmoe_conv_wgrad6_ws_kib(G, N, H, W, I, O, ctypes.cast(_int_array(khs), ctypes.c_void_p),
                                             ctypes.cast(_int_array(kws), ctypes.c_void_p), dtc)
        if kib > 0 and bank is not None and W6_DEFER and _prof_ok():
            # weight-bank layer: the partial slabs stay in the arena, the bank sums all layers' partials in one batched launch
            ws = _w6_arena_take(x.device, 2 * kib * 256)
            if ws is not None:
                if _timed("conv_wgrad", dict(info, dtype="split_bf16" if split else info.get("dtype"), wgrad_name="wgrad6_kernel<split> (deferred reduce)" if split else "wgrad6_kernel (deferred reduce)"),
                          "hdmoe_conv_wgrad6", x, dy, Gs, seg, G, N, H, W, I, O, khs, kws, pts, pts, ws, ws.numel() * 4, dtc, 1) == 0:
                    bank.defer_w6(list(Gs), seg, ws, [G, N, H, W, I, O, dtc, 0] + [int(k) for k in khs] + [0] * (8 - len(khs)))
                    STATS["w6_defer"] += 1
                    return
        if kib > 0:
            key = (x.device, torch.cuda.current_stream().stream_id)       # branches on different streams run concurrently
            ws = _w6_ws.get(key)
            if ws is None or ws.numel() < kib * 256:
                ws = torch.empty(kib * 256, dtype=torch.float32, device=x.device)      # fp32 words; grown outside graph capture (warm-up steps)
                _w6_ws[key] = ws
            if split:
                info = dict(info, dtype="split_bf16")
            if _timed("conv_wgrad", info, "hdmoe_conv_wgrad6", x, dy, Gs, seg, G, N, H, W, I, O, khs, kws, pts, pts, ws, ws.numel() * 4, dtc, 0) == 0:
                return
    _timed("conv_wgrad", info, "hdmoe_conv_wgrad", x, dy, Gs, seg, G, N, H, W, Ho, Wo, I, Cphys, O, 1, 1 if ones else 0, khs, kws, pts, pts, _dt(x))


class _MPConvFn(torch.autograd.Function):
    """y = alpha * conv(x, prep(w)) + beta * res.   x: (N, H, W, Cphys).
    ``tensors`` = G weights followed by 0 or G learnable gain scalars (one per group)."""

    @staticmethod
    def forward(ctx, x, res, seg, meta, *tensors):
        (G, gain_val, alpha, beta, ones, training, normalize, split, res_raw) = meta
        weights, gains = tensors[:G], list(tensors[G:]) or None
        x = _c(x)
        N, H, W, Cphys = x.shape
        O, I = int(weights[0].shape[0]), int(weights[0].shape[1])
        khs = [_kernel_hw(w)[0] for w in weights]
        kws = [_kernel_hw(w)[1] for w in weights]
        # reference pads by the LAST kernel dim on both axes (model_internals.py:264-270)
        pts = [(kw - 1) // 2 for kw in kws]
        Hos = {H + (kw - 1) - kh + 1 for kh, kw in zip(khs, kws)}
        if len(Hos) != 1:
            raise ValueError("grouped conv: all groups must produce the same output size")
        if any(w.shape[0] != O or w.shape[1] != I for w in weights):
            raise ValueError("grouped conv: all groups must share (out_channels, in_channels)")
        if I != Cphys + (1 if ones else 0):
            raise RuntimeError(f"MP_Conv: input has {Cphys} channels, weight expects {I}")
        Ho, Wo = Hos.pop(), W
        Ipad = (I + 15) // 16 * 16
        Opad = (O + 15) // 16 * 16
        taps = max(a * b for a, b in zip(khs, kws))
        wstride, wdstride = taps * O * Ipad, taps * I * Opad
        # split: fp32 tensors, split-bf16 arithmetic (csrc/conv6s.hip) -- weight images are bf16 [hi | lo] planes, dtype code 2
        wdt, planes, dtc = (torch.bfloat16, 2, F32S) if split else (x.dtype, 1, _dt(x))
        ent = None
        if _bank.ACTIVE is not None and gains is None:
            ent = _bank.ACTIVE.lookup(weights, "split" if split else x.dtype, gain_val, alpha, normalize)
        if ent is not None:                                   # images already prepared by the bank's single launch
            wf, wd = ent.wf, ent.wd
        else:
            wf = torch.empty(planes * G * wstride, dtype=wdt, device=x.device)
            # the flipped dgrad image is produced by the same prep launch when the input needs a gradient
            wd = torch.empty(planes * G * wdstride, dtype=wdt, device=x.device) if ctx.needs_input_grad[0] else None
            call("hdmoe_wprep_fwd", list(weights), gains, gain_val, khs, kws, G, O, I, Ipad, Opad, wf, wstride, wd, wdstride,
                 1 if normalize else 0, 1 if training else 0, 1, dtc)
            if training and normalize:                         # the train-mode forward re-normalises the stored weights in place (raw pointers:
                _bank.note_weights_changed()                   #  Tensor._version stays) -- eval-mode images prepared earlier are stale
        ctx.wd = wd
        ctx.ent = ent
        ctx.bank = _bank.ACTIVE if ent is not None else None
        global _PRECOMP
        pre, _PRECOMP = _PRECOMP, None
        y = pre if pre is not None else torch.empty((N, Ho, Wo, O), dtype=x.dtype, device=x.device)
        req = _FILM_REQ
        fused_film = pre is not None                          # (output already computed by the fused block kernel, ops.unet_block_fused)
        ctx.ones6 = False
        if (ones and ONES6 and pre is None and ent is not None and not split and res is None and x.dtype == torch.bfloat16 and _prof_ok()
                and Ho == H and Wo == W and khs == kws):
            # torch.cat([x, ones]) (reference model_components.py:416): the 32 real channels on conv6, the ones channel as a bias map (csrc/ones6.hip)
            gb = torch.empty((G, H, W, O), dtype=torch.float32, device=x.device)
            if _timed("fused", dict(name="conv6_bf16_kernel (ones-channel layer)", dtype="bfloat16", seg=seg, N=N, HW=H * W, O=O, I=Cphys, taps=[a * b for a, b in zip(khs, kws)], mult=1.0),
                      "hdmoe_conv6_ones_fwd", x, wf, y, gb, alpha, seg, G, wstride, N, H, W, Cphys, O, Ipad, khs, dtc) == 0:
                fused_film = True
                ctx.ones6 = True
                STATS["ones6"] += 1
        if (pre is None and req is not None and ent is not None and not split and res is None and not ones and x.dtype == torch.bfloat16 and PROFILE is None
                and Ho == H and Wo == W and Cphys == I):
            # FiLM + mp_silu + dropout as a second output of this conv's epilogue (ops.mp_conv_film): one launch less on the branch's chain
            hbuf = torch.empty_like(y)
            if call("hdmoe_conv_fwd_film", x, wf, y, hbuf, req.emb, req.seed, step_counter(x.device), req.p, alpha, seg, G, wstride, N, H, W, I, O,
                    khs, kws, pts, pts, dtc) == 0:
                req.h = hbuf
                fused_film = True
        if not fused_film:
            _timed("conv_fwd", _conv_info(x, seg, N, Ho, Wo, O, I, O, khs, kws, Cphys, split), "hdmoe_conv_fwd", x, wf, y, _c(res), alpha, beta,
                   seg, G, wstride, N, H, W, Ho, Wo, I, Cphys, Ipad, O, O, 1, 1 if ones else 0, khs, kws, pts, pts, dtc)
        ctx.save_for_backward(x, seg, *tensors)
        ctx.meta = (G, gain_val, alpha, beta, ones, normalize, khs, kws, pts, Ho, Wo, res is not None, split, res_raw)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, seg, *tensors = ctx.saved_tensors
        G, gain_val, alpha, beta, ones, normalize, khs, kws, pts, Ho, Wo, has_res, split, res_raw = ctx.meta
        weights, gains = tensors[:G], list(tensors[G:]) or None
        dy = _c(dy)
        N, H, W, Cphys = x.shape
        O, I = int(weights[0].shape[0]), int(weights[0].shape[1])
        dt = x.dtype
        dx = dres = None
        nig = ctx.needs_input_grad
        need_gain = gains is not None and any(nig[4 + G + g] for g in range(G))
        need_w = any(nig[4 + g] for g in range(G)) or need_gain
        fused = False
        if (nig[0] and need_w and ctx.ent is not None and BWD6 and W6_DEFER and _prof_ok() and x.dtype == torch.bfloat16 and not split
                and not ones and Ho == H and Wo == W and Cphys == I and set(khs) == {3, 5} and khs == kws):
            # input gradient + (deferred) weight gradient of a 3x3 / 5x5 expert layer in one launch (csrc/bwd6.hip)
            from ._lib import lib, _int_array
            import ctypes
            kib = lib().hdmoe_conv_wgrad6_ws_kib(G, N, H, W, I, O, ctypes.cast(_int_array(khs), ctypes.c_void_p),
                                                 ctypes.cast(_int_array(kws), ctypes.c_void_p), _dt(x))
            ws = _w6_arena_take(x.device, 2 * kib * 256) if kib > 0 else None
            if ws is not None:
                dx = torch.empty_like(x)
                Opad = (O + 15) // 16 * 16
                wdstride = max(a * b for a, b in zip(khs, kws)) * I * Opad
                if _timed("fused", dict(name="bwd6_kernel", dtype="bfloat16", seg=seg, N=N, HW=H * W, O=O, I=I, taps=[a * b for a, b in zip(khs, kws)], mult=2.0),
                          "hdmoe_conv_bwd6", x, dy, ctx.wd, dx, list(ctx.ent.G), seg, G, wdstride, N, H, W, I, O, khs, kws, pts, pts, alpha,
                          ws, ws.numel() * 4, _dt(x)) == 0:
                    ctx.bank.defer_w6(list(ctx.ent.G), seg, ws, [G, N, H, W, I, O, _dt(x), 0] + [int(k) for k in khs] + [0] * (8 - len(khs)))
                    ctx.bank.note_backward(ctx.ent)
                    fused = True
                    STATS["bwd6"] += 1
                else:
                    dx = None
        if (nig[0] and need_w and ctx.ent is not None and BWD6 and W6_DEFER and _prof_ok() and split and not ones and Ho == H and Wo == W
                and Cphys == I and set(khs) == {3} and khs == kws):
            # the same for a router-trunk layer (fp32 tensors, split-bf16 arithmetic)
            from ._lib import lib, _int_array
            import ctypes
            kib = lib().hdmoe_conv_wgrad6_ws_kib(G, N, H, W, I, O, ctypes.cast(_int_array(khs), ctypes.c_void_p),
                                                 ctypes.cast(_int_array(kws), ctypes.c_void_p), F32S)
            ws = _w6_arena_take(x.device, 2 * kib * 256) if kib > 0 else None
            if ws is not None:
                dx = torch.empty_like(x)
                Opad = (O + 15) // 16 * 16
                wdstride = 9 * I * Opad
                if _timed("fused", dict(name="bwd6s_kernel", dtype="bfloat16" if TRUNK_BWD_BF16 else "split_bf16", seg=seg, N=N, HW=H * W, O=O, I=I, taps=[9] * G, mult=2.0),
                          "hdmoe_conv_bwd6s", x, dy, ctx.wd, dx, list(ctx.ent.G), seg, G, wdstride, G * wdstride, N, H, W, I, O, khs, kws, pts, pts,
                          alpha, ws, ws.numel() * 4, None, None, 0, 1 if TRUNK_BWD_BF16 else 0) == 0:
                    ctx.bank.defer_w6(list(ctx.ent.G), seg, ws, [G, N, H, W, I, O, F32S, 0] + [int(k) for k in khs] + [0] * (8 - len(khs)))
                    ctx.bank.note_backward(ctx.ent)
                    fused = True
                    STATS["bwd6s"] += 1
                else:
                    dx = None
        if ctx.ones6 and need_w and ctx.ent is not None:
            # the ones-channel layer: dgrad on conv6, weight gradient on wgrad6 + pixel sums of dy (csrc/ones6.hip)
            from ._lib import lib, _int_array
            import ctypes
            kib = lib().hdmoe_conv_wgrad6_ws_kib(G, N, H, W, Cphys, O, ctypes.cast(_int_array(khs), ctypes.c_void_p),
                                                 ctypes.cast(_int_array(kws), ctypes.c_void_p), _dt(x))
            if kib > 0:
                key = (x.device, torch.cuda.current_stream().stream_id)
                ws = _w6_ws.get(key)
                if ws is None or ws.numel() < kib * 256:
                    ws = torch.empty(kib * 256, dtype=torch.float32, device=x.device)
                    _w6_ws[key] = ws
                S = torch.empty((G, H, W, O), dtype=torch.float32, device=x.device)
                g32 = [_zeros((kh * kw, O, Cphys), torch.float32, x.device) for kh, kw in zip(khs, kws)]
                dxo = torch.empty_like(x) if nig[0] else None
                Opad = (O + 15) // 16 * 16
                wdstride = max(a * b for a, b in zip(khs, kws)) * I * Opad
                if _timed("fused", dict(name="conv6 dgrad + wgrad6 (ones-channel layer)", dtype="bfloat16", seg=seg, N=N, HW=H * W, O=O, I=Cphys, taps=[a * b for a, b in zip(khs, kws)], mult=2.0 if nig[0] else 1.0),
                          "hdmoe_conv6_ones_bwd", x, dy, ctx.wd, dxo, list(ctx.ent.G), S, g32, seg, G, wdstride, N, H, W, Cphys, O, Opad, khs, alpha,
                          ws, ws.numel() * 4, _dt(x)) == 0:
                    dx = dxo
                    ctx.bank.note_backward(ctx.ent)
                    fused = True
        if nig[0] and not fused:
            Opad = (O + 15) // 16 * 16
            wdstride = max(a * b for a, b in zip(khs, kws)) * I * Opad
            wd = ctx.wd
            dx = torch.empty_like(x)
            pt_d = [kh - 1 - p for kh, p in zip(khs, pts)]
            pl_d = [kw - 1 - p for kw, p in zip(kws, pts)]
            # dgrad: conv over dy (O channels) with the flipped kernel; logical out channels I, stored Cphys
            _timed("conv_fwd", _conv_info(dy, seg, N, H, W, I, O, Cphys, khs, kws, O, split), "hdmoe_conv_fwd", dy, wd, dx, None, alpha, 0.0,
                   seg, G, wdstride, N, Ho, Wo, H, W, O, O, Opad, I, Cphys, 1, 0, khs, kws, pt_d, pl_d, F32S if split else _dt(x))
        if has_res and nig[1]:
            if res_raw or beta == 1.0:
                dres = dy                                     # beta == 1, or the producer of `res` applies beta in its own backward pass
            else:
                dres = torch.empty_like(dy)
                call("hdmoe_axpby", dres, dy, None, beta, 0.0, dy.numel(), _dt(dy))
        dws: List[Optional[Tensor]] = [None] * G
        dgs: List[Optional[Tensor]] = [None] * (len(tensors) - G)
        if fused:
            pass
        elif need_w and ctx.ent is not None:
            # bank path: accumulate into the bank's slab; one multi-tensor launch at the end of backward finishes every gradient
            _wgrad(_conv_info(x, seg, N, Ho, Wo, O, I, O, khs, kws, Cphys), x, dy, ctx.ent.G, seg, G, N, H, W, Ho, Wo, I, Cphys, O, ones, khs, kws, pts, split,
                   bank=ctx.bank)
            ctx.bank.note_backward(ctx.ent)
        elif need_w:
            sizes = [khs[g] * kws[g] * O * I for g in range(G)]
            Gflat = torch.zeros(sum(sizes), dtype=torch.float32, device=x.device)           # one memset for all groups
            Gs, off = [], 0
            for g in range(G):
                Gs.append(Gflat[off:off + sizes[g]])
                off += sizes[g]
            _wgrad(_conv_info(x, seg, N, Ho, Wo, O, I, O, khs, kws, Cphys), x, dy, Gs, seg, G, N, H, W, Ho, Wo, I, Cphys, O, ones, khs, kws, pts, split)
            dws = [torch.empty_like(w) for w in weights]
            if need_gain:
                dgs = [torch.zeros((), dtype=torch.float32, device=x.device) for _ in range(G)]
            call("hdmoe_wprep_bwd", list(weights), gains, gain_val, Gs, dws, dgs if need_gain else None, khs, kws, G, O, I,
                 1 if normalize else 0)
            if alpha != 1.0:
                for d in list(dws) + [d for d in dgs if d is not None]:
                    call("hdmoe_axpby", d, d, None, alpha, 0.0, d.numel(), 0)
        return (dx, dres, None, None, *dws, *dgs)


class _StridedConvFn(torch.autograd.Function):
    """MP_Conv with stride > 1 (reference model_internals.py:272-275: F.conv2d(x, w, padding=k // 2, stride)).  No reference model uses
    it; forward and weight gradient run on the general strided kernels (the patch embedding's), the input gradient is not implemented."""

    @staticmethod
    def forward(ctx, x, w, gain, stride, training):
        x = _c(x)
        N, H, W, C = x.shape
        O, I, kh, kw = (int(v) for v in w.shape)
        if I != C:
            raise RuntimeError(f"MP_Conv: input has {C} channels, weight expects {I}")
        pad = kw // 2                                          # the reference pads both axes by the last kernel dim
        Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        Ipad = (I + 15) // 16 * 16
        wf = torch.empty(kh * kw * O * Ipad, dtype=x.dtype, device=x.device)
        call("hdmoe_wprep_fwd", [w], None, float(gain), [kh], [kw], 1, O, I, Ipad, 16, wf, wf.numel(), None, 0, 1, 1 if training else 0, 1, _dt(x))
        if training:
            _bank.note_weights_changed()
        y = torch.empty((N, Ho, Wo, O), dtype=x.dtype, device=x.device)
        call("hdmoe_conv_fwd", x, wf, y, None, 1.0, 0.0, None, 1, wf.numel(), N, H, W, Ho, Wo, I, I, Ipad, O, O, stride, 0, [kh], [kw], [pad], [pad], _dt(x))
        ctx.save_for_backward(x, w)
        ctx.meta = (float(gain), stride, pad, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        gain, stride, pad, Ho, Wo = ctx.meta
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("MP_Conv(stride > 1): the input gradient is not implemented (no reference model uses a strided MP_Conv)")
        N, H, W, C = x.shape
        O, I, kh, kw = (int(v) for v in w.shape)
        dw = None
        if ctx.needs_input_grad[1]:
            Gs = [_zeros((kh * kw, O, I), torch.float32, x.device)]
            call("hdmoe_conv_wgrad", x, _c(dy), Gs, None, 1, N, H, W, Ho, Wo, I, I, O, stride, 0, [kh], [kw], [pad], [pad], _dt(x))
            dw = torch.empty_like(w)
            call("hdmoe_wprep_bwd", [w], None, gain, Gs, [dw], None, [kh], [kw], 1, O, I, 1)
        return None, dw, None, None, None


def mp_conv_strided(x: Tensor, w: Tensor, gain: float, stride: int, training: bool = False) -> Tensor:
    (w,) = f32_params([w])
    return _StridedConvFn.apply(x, w, float(gain), int(stride), bool(training))


_PRECOMP = None                           # output tensor a fused launch has already produced for the NEXT _MPConvFn.forward (ops.unet_block_fused)
F32S = 2                                  # C-ABI dtype code: fp32 tensors, split-bf16 arithmetic (include/hdmoe.h HDMOE_F32S)
# The fp32 router trunks run on the bf16 matrix pipe as split-bf16 (3 MFMAs per product, ~1e-5 relative): HDMOE_ROUTER_SPLIT=0
# keeps them on the fp32-input MFMA kernels.
ROUTER_SPLIT = _os.environ.get("HDMOE_ROUTER_SPLIT", "1") != "0" and _os.environ.get("HDMOE_CONV6", "1") != "0"   # (the split kernels are conv6's)


def _split_ok(x4: Tensor, ws, ones: bool) -> bool:
    """Domain of the split-bf16 conv kernels (csrc/conv6s.hip); mirrored here because the weight image format depends on it."""
    if x4.dtype != torch.float32 or ones or x4.ndim != 4:
        return False
    _, H, W, C = x4.shape
    O = ws[0].shape[0]
    return (all(w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3 for w in ws) and C % 32 == 0 and O % 32 == 0
            and (W == 16 or W % 32 == 0) and H >= 8)


def mp_conv(x: Tensor, weights, gain=1.0, *, seg: Optional[Tensor] = None, res: Optional[Tensor] = None, alpha: float = 1.0,
            beta: float = 0.0, ones: bool = False, training: bool = False, normalize: bool = True, split: bool = False,
            res_grad_raw: bool = False) -> Tensor:
    """Magnitude-preserving conv / linear (reference MP_Conv.forward, model_internals.py:253-275).

    ``x``: (N,H,W,C) -> (N,Ho,Wo,O);  (N,S,C) -> (N,S,O);  (M,C) -> (M,O).  ``weights``: a tensor, or a list of
    per-expert tensors together with ``seg`` (device int32 row offsets) for a grouped launch.
    ``gain``: python float, a 0-dim float32 tensor (learnable out_gain), or a list of such tensors (one per group)."""
    ws = f32_params(list(weights) if isinstance(weights, (list, tuple)) else [weights])
    G = len(ws)
    if isinstance(gain, (list, tuple)):
        gts, gain_val = f32_params(list(gain)), 1.0
    elif torch.is_tensor(gain):
        gts, gain_val = f32_params([gain]) * G, 1.0
    else:
        gts, gain_val = [], float(gain)
    shape = x.shape
    grouped = seg is not None
    if x.ndim not in (2, 3, 4):
        raise ValueError("mp_conv: x must be 2-, 3- or 4-D")
    pointwise = all(w.ndim == 2 or (w.shape[2] == 1 and w.shape[3] == 1) for w in ws)
    if pointwise and not grouped and not ones:
        # 1x1 / linear layer without per-row experts: every position is independent, so present the tensor as ONE long row
        # of positions -- tiles are then always full (a (B,16,C) token tensor would otherwise fill 16 of 128 tile slots)
        x4 = x.reshape(1, 1, -1, shape[-1])
    elif x.ndim == 2:
        x4 = x.reshape(shape[0], 1, 1, shape[1])
    elif x.ndim == 3:
        x4 = x.reshape(shape[0], 1, shape[1], shape[2])
    else:
        x4 = x
    if res is not None:
        res = res.reshape(x4.shape[0], x4.shape[1], x4.shape[2], -1)
    split = bool(split) and _split_ok(x4, ws, ones)
    # res_grad_raw: the gradient handed back for `res` is dy itself, NOT beta * dy -- only for a `res` whose producer was created with
    # gx_scale = beta (ops.silu_branch / ops.pixel_norm_silu) and has no other consumer: saves a scaling pass per residual block
    meta = (G, gain_val, float(alpha), float(beta), bool(ones), bool(training), bool(normalize), split, bool(res_grad_raw))
    y = _MPConvFn.apply(x4, res, seg, meta, *ws, *gts)
    return y.reshape(*shape[:-1], y.shape[-1])


class _MultiLinearFn(torch.autograd.Function):
    """L linear layers over the same fp32 input in one launch each for forward / input gradient / weight gradient
    (csrc/mlinear.hip).  ``tensors`` = L * G weights (layer-major); every layer must be a ready weight-bank entry."""

    @staticmethod
    def forward(ctx, x, seg, ents, bank, c, *tensors):
        x = _f32(x)
        R, I = x.shape
        L = len(ents)
        G = len(ents[0].params)
        Os = [e.O for e in ents]
        ys = [torch.empty((R, o), dtype=torch.float32, device=x.device) for o in Os]
        call("hdmoe_mlinear_fwd", ys, x, [e.wf for e in ents], seg, Os, L, R, I, ents[0].Ipad, G, c)
        ctx.save_for_backward(x, seg)
        ctx.meta = (ents, bank, Os, G)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        x, seg = ctx.saved_tensors
        ents, bank, Os, G = ctx.meta
        R, I = x.shape
        L = len(ents)
        gs = [_f32(g) if g is not None else _zeros((R, o), torch.float32, x.device) for g, o in zip(gs, Os)]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("hdmoe_mlinear_dgrad", dx, gs, [e.wf for e in ents], seg, Os, L, R, I, ents[0].Ipad, G)
        if any(ctx.needs_input_grad[5:]):
            Gs = []
            for e in ents:
                Gs += list(e.G) + [None] * (8 - G)
            call("hdmoe_mlinear_wgrad", Gs, gs, x, seg, Os, L, R, I, G)
            for e in ents:
                bank.note_backward(e)
        return (dx, None, None, None, None) + (None,) * (L * G)


def multi_linear(x: Tensor, layers, gain: float, *, seg: Optional[Tensor] = None, c: float = 0.0, training: bool = False):
    """[c + mp_linear_l(x) for l in layers] for MP_Conv layers (kernel ()) that all read ``x`` (R, I) fp32; ``layers`` = list of per-layer
    weight lists (one weight per expert with ``seg``).  One launch for all layers once the weight bank has their images; the plain
    per-layer path otherwise (first steps, no bank, more than 16 layers)."""
    L = len(layers)
    bank = _bank.ACTIVE
    ents = None
    if bank is not None and 1 <= L <= 16 and x.ndim == 2 and x.shape[1] <= 256 and x.dtype == torch.float32 and MULTI_LINEAR:
        ents = [bank.lookup(ws, torch.float32, float(gain), 1.0, True) for ws in layers]
        ok = all(e is not None for e in ents) and len({(e.I, e.Ipad, len(e.params)) for e in ents if e is not None}) == 1
        if ok and all(e.khs == [1] * len(e.params) for e in ents) and (seg is not None or len(ents[0].params) == 1):
            flat = [w for ws in layers for w in ws]
            return list(_MultiLinearFn.apply(x, seg, ents, bank, float(c), *flat))
    outs = [mp_conv(x, ws if seg is not None else ws[0], gain, seg=seg, training=training) for ws in layers]
    return [affine(o, 1.0, c) for o in outs] if c != 0.0 else outs


MULTI_LINEAR = _os.environ.get("HDMOE_MULTI_LINEAR", "1") != "0"


class _PatchLinearFn(torch.autograd.Function):
    """Vit_expert.patch when the image divides into patches (model_components.py:670-679): a stride-p p x p conv is a linear layer on
    the patch vectors, so: one relayout pass (image -> tokens of C*p*p features in the weight's own (c, i, j) order, the parameter
    is used as the [E][C*p*p] matrix it already is in memory), then the long-contraction pointwise kernel (csrc/kgemm.hip), and in
    the backward the streamed pointwise weight gradient (csrc/lwgrad.hip) -- instead of the generic strided conv kernels, whose
    weight gradient alone cost 190 us per expert."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        N, H, W, C = x.shape
        E, p = int(w.shape[0]), int(w.shape[2])
        hp, wp, K = H // p, W // p, C * p * p
        Epad = (E + 15) // 16 * 16
        tok = torch.empty((N, hp, wp, K), dtype=x.dtype, device=x.device)
        call("hdmoe_patch_relayout", tok, x, N, H, W, C, p, hp, wp, 1, 0, _dt(x))
        wf = torch.empty(E * K, dtype=x.dtype, device=x.device)
        wd = torch.empty(K * Epad, dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        call("hdmoe_wprep_fwd", [w], None, 1.0, [1], [1], 1, E, K, K, Epad, wf, wf.numel(), wd, 0 if wd is None else wd.numel(), 0, 0, 0, _dt(x))
        y0 = torch.empty((N, hp, wp, E), dtype=x.dtype, device=x.device)
        call("hdmoe_conv_fwd", tok, wf, y0, None, 1.0, 0.0, None, 1, wf.numel(), N, hp, wp, hp, wp, K, K, K, E, E, 1, 0, [1], [1], [0], [0], _dt(x))
        y = torch.empty_like(y0)
        call("hdmoe_bias_add", y, y0, b, N * hp * wp, E, _dt(x))
        ctx.save_for_backward(tok, w)
        ctx.wd = wd
        ctx.dims = (N, H, W, C, E, p, hp, wp, K, Epad)
        return y.reshape(N, hp * wp, E)

    @staticmethod
    def backward(ctx, dy):
        tok, w = ctx.saved_tensors
        N, H, W, C, E, p, hp, wp, K, Epad = ctx.dims
        dy = _c(dy).reshape(N, hp, wp, E)
        dt = _dt(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dtok = torch.empty((N, hp, wp, K), dtype=dy.dtype, device=dy.device)
            call("hdmoe_conv_fwd", dy, ctx.wd, dtok, None, 1.0, 0.0, None, 1, ctx.wd.numel(), N, hp, wp, hp, wp, E, E, Epad, K, K, 1, 0, [1], [1], [0], [0], dt)
            dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
            call("hdmoe_patch_relayout", dx, dtok, N, H, W, C, p, hp, wp, 1, 1, dt)
        if ctx.needs_input_grad[1]:
            Gs = [_zeros((1, E, K), torch.float32, dy.device)]
            call("hdmoe_conv_wgrad", tok, dy, Gs, None, 1, N, hp, wp, hp, wp, K, K, E, 1, 0, [1], [1], [0], [0], dt)
            dw = torch.empty_like(w)
            call("hdmoe_wprep_bwd", [w], None, 1.0, Gs, [dw], None, [1], [1], 1, E, K, 0)
        if ctx.needs_input_grad[2]:
            db = torch.zeros(E, dtype=torch.float32, device=dy.device)
            call("hdmoe_colsum", db, dy, N * hp * wp, E, dt)
        return dx, dw, db


class _PatchEmbedFn(torch.autograd.Function):
    """Vit_expert.patch: nn.Conv2d(C, E, p, stride=p) with bias on the zero-padded image (model_components.py:670-679)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        N, H, W, C = x.shape
        E, p = int(w.shape[0]), int(w.shape[2])
        hp, wp = -(-H // p), -(-W // p)
        Ipad = (C + 15) // 16 * 16
        wf = torch.empty(p * p * E * Ipad, dtype=x.dtype, device=x.device)
        call("hdmoe_wprep_fwd", [w], None, 1.0, [p], [p], 1, E, C, Ipad, 16, wf, wf.numel(), None, 0, 0, 0, 0, _dt(x))
        y0 = torch.empty((N, hp, wp, E), dtype=x.dtype, device=x.device)
        call("hdmoe_conv_fwd", x, wf, y0, None, 1.0, 0.0, None, 1, wf.numel(), N, H, W, hp, wp, C, C, Ipad, E, E, p, 0, [p], [p],
             [0], [0], _dt(x))
        y = torch.empty_like(y0)
        call("hdmoe_bias_add", y, y0, b, N * hp * wp, E, _dt(x))
        ctx.save_for_backward(x, w)
        return y.reshape(N, hp * wp, E)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, H, W, C = x.shape
        E, p = int(w.shape[0]), int(w.shape[2])
        hp, wp = -(-H // p), -(-W // p)
        dy = _c(dy).reshape(N, hp, wp, E)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            Ipad = (C + 15) // 16 * 16
            Epad = (E + 15) // 16 * 16
            wf = torch.empty(p * p * E * Ipad, dtype=x.dtype, device=x.device)
            wd = torch.empty(p * p * C * Epad, dtype=x.dtype, device=x.device)     # [(tap, c)][Epad] == 1x1 weight, p*p*C outputs
            call("hdmoe_wprep_fwd", [w], None, 1.0, [p], [p], 1, E, C, Ipad, Epad, wf, wf.numel(), wd, wd.numel(), 0, 0, 0, _dt(x))
            tok = torch.empty((N, hp, wp, p * p * C), dtype=x.dtype, device=x.device)
            call("hdmoe_conv_fwd", dy, wd, tok, None, 1.0, 0.0, None, 1, wd.numel(), N, hp, wp, hp, wp, E, E, Epad, p * p * C,
                 p * p * C, 1, 0, [1], [1], [0], [0], _dt(x))
            dx = torch.empty_like(x)
            call("hdmoe_patch_relayout", dx, tok, N, H, W, C, p, hp, wp, 0, 1, _dt(x))
        if ctx.needs_input_grad[1]:
            Gs = [_zeros((p * p, E, C), torch.float32, x.device)]
            call("hdmoe_conv_wgrad", x, dy, Gs, None, 1, N, H, W, hp, wp, C, C, E, p, 0, [p], [p], [0], [0], _dt(x))
            dw = torch.empty_like(w)
            call("hdmoe_wprep_bwd", [w], None, 1.0, Gs, [dw], None, [p], [p], 1, E, C, 0)
        if ctx.needs_input_grad[2]:
            db = torch.zeros(E, dtype=torch.float32, device=x.device)
            call("hdmoe_colsum", db, dy, N * hp * wp, E, _dt(x))
        return dx, dw, db


def patch_embed(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """(N,H,W,C) -> tokens (N, ceil(H/p)*ceil(W/p), E)."""
    p = int(w.shape[2])
    if (x.shape[1] % p == 0 and x.shape[2] % p == 0 and w.shape[2] == w.shape[3] and w.is_contiguous()
            and x.shape[3] % (8 if x.dtype == torch.bfloat16 else 4) == 0 and (x.shape[3] * p * p) % 16 == 0):
        return _PatchLinearFn.apply(x, w, b)
    return _PatchEmbedFn.apply(x, w, b)


# =====================================================================================================
# pointwise
# =====================================================================================================
class _AxpbyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, a, b):
        x = _c(x); y = _c(y)
        out = torch.empty_like(x)
        call("hdmoe_axpby", out, x, y, a, b, x.numel(), _dt(x))
        ctx.ab = (a, b, y is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, has_y = ctx.ab
        g = _c(g)
        dx = dy = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(g); call("hdmoe_axpby", dx, g, None, a, 0.0, g.numel(), _dt(g))
        if has_y and ctx.needs_input_grad[1]:
            if dx is not None and a == b:
                dy = dx                                       # equal weights (mp_sum with t = 0.5): one scaled copy serves both inputs
            else:
                dy = torch.empty_like(g); call("hdmoe_axpby", dy, g, None, b, 0.0, g.numel(), _dt(g))
        return dx, dy, None, None


def axpby(x: Tensor, y: Optional[Tensor], a: float, b: float = 0.0) -> Tensor:
    return _AxpbyFn.apply(x, y, float(a), float(b))


def mp_sum(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """lerp(a,b,t)/sqrt((1-t)^2+t^2) (model_internals.py:50-66)."""
    n = math.sqrt((1.0 - t) ** 2 + t ** 2)
    return axpby(a, b, (1.0 - t) / n, t / n)


class _AffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, c):
        x = _c(x)
        out = torch.empty_like(x)
        call("hdmoe_affine", out, x, a, c, x.numel(), _dt(x))
        ctx.a = a
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        dx = torch.empty_like(g)
        call("hdmoe_axpby", dx, g, None, ctx.a, 0.0, g.numel(), _dt(g))
        return dx, None, None


def affine(x: Tensor, a: float, c: float) -> Tensor:
    return _AffineFn.apply(x, float(a), float(c))


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x = _c(x); y = _c(y)
        out = torch.empty_like(x)
        call("hdmoe_mul", out, x, y, x.numel(), _dt(x))
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _c(g)
        dx = dy = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(g); call("hdmoe_mul", dx, g, y, g.numel(), _dt(g))
        if ctx.needs_input_grad[1]:
            dy = torch.empty_like(g); call("hdmoe_mul", dy, g, x, g.numel(), _dt(g))
        return dx, dy


def mul(x: Tensor, y: Tensor) -> Tensor:
    return _MulFn.apply(x, y)


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        x = _c(x)
        ctx.src = x.dtype
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
        call("hdmoe_cast", out, x, x.numel(), dtype_code_cast(x.dtype), dtype_code_cast(dtype))
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        dx = torch.empty(g.shape, dtype=ctx.src, device=g.device)
        call("hdmoe_cast", dx, g, g.numel(), dtype_code_cast(g.dtype), dtype_code_cast(ctx.src))
        return dx, None


def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    """Element type conversion.  float16 is accepted at the module boundary only (fp16 <-> fp32; the kernels compute in fp32 / bf16)."""
    if x.dtype == dtype:
        return x
    if torch.float16 in (x.dtype, dtype) and torch.bfloat16 in (x.dtype, dtype):
        return _CastFn.apply(_CastFn.apply(x, torch.float32), dtype)
    return _CastFn.apply(x, dtype)


def f32_params(ts):
    """fp16 parameters (a module after `.half()`) as fp32 tensors for the kernels; gradients flow back in fp16."""
    return [cast(t, torch.float32) if torch.is_tensor(t) and t.dtype == torch.float16 else t for t in ts]


class _MPSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        out = torch.empty_like(x)
        call("hdmoe_mp_silu_fwd", out, x, x.numel(), _dt(x))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty_like(x)
        call("hdmoe_mp_silu_bwd", dx, g, x, x.numel(), _dt(x))
        return dx


def mp_silu(x: Tensor) -> Tensor:
    return _MPSiluFn.apply(x)


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a):
        x = _c(x)
        out = torch.empty_like(x)
        call("hdmoe_sigmoid_fwd", out, x, a, x.numel(), _dt(x))
        ctx.save_for_backward(out)
        ctx.a = a
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty_like(y)
        call("hdmoe_sigmoid_bwd", dx, g, y, ctx.a, y.numel(), _dt(y))
        return dx, None


def sigmoid(x: Tensor, a: float = 1.0) -> Tensor:
    """sigmoid(a * x)."""
    return _SigmoidFn.apply(x, float(a))


class _FilmSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, e, p, seed):
        ctx.pool_ok = not e.is_leaf
        u = _c(u); e = _f32(e)
        N, C = u.shape[0], u.shape[-1]
        HW = u.numel() // (N * C)
        out = torch.empty_like(u)
        if p > 0.0:
            call("hdmoe_film_silu_drop_fwd", out, u, e, N, HW, C, seed, step_counter(u.device), p, _dt(u))
        else:
            call("hdmoe_film_silu_fwd", out, u, e, N, HW, C, _dt(u))
        ctx.save_for_backward(u, e)
        ctx.meta = (p, seed)
        return out

    @staticmethod
    def backward(ctx, g):
        u, e = ctx.saved_tensors
        p, seed = ctx.meta
        g = _c(g)
        N, C = u.shape[0], u.shape[-1]
        HW = u.numel() // (N * C)
        du = torch.empty_like(u)
        de = _zeros(e.shape, e.dtype, e.device, ctx.pool_ok)
        if p > 0.0:
            call("hdmoe_film_silu_drop_bwd", du, de, g, u, e, N, HW, C, seed, step_counter(u.device), p, _dt(u))
        else:
            call("hdmoe_film_silu_bwd", du, de, g, u, e, N, HW, C, _dt(u))
        return du, de, None, None


class _FilmReq:
    __slots__ = ("emb", "p", "seed", "h")

    def __init__(self, emb, p, seed):
        self.emb, self.p, self.seed, self.h = emb, p, seed, None


_FILM_REQ = None                                        # set around the mp_conv call of ops.mp_conv_film
CONV_FILM = _os.environ.get("HDMOE_CONV_FILM", "1") != "0"
CONV_FILM_TRAIN = _os.environ.get("HDMOE_CONV_FILM", "1") == "2"


class _FilmDoneFn(torch.autograd.Function):
    """film_silu whose forward was already computed by the producing conv's epilogue (h): only the backward is left."""

    @staticmethod
    def forward(ctx, u, e, p, seed, h):
        ctx.pool_ok = not e.is_leaf
        ctx.save_for_backward(u, e)
        ctx.meta = (p, seed)
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        return _FilmSiluFn.backward(ctx, g) + (None,)


def mp_conv_film(x: Tensor, weights, gain, emb: Tensor, p: float, training: bool, seg: Optional[Tensor] = None) -> Tensor:
    """film_silu(mp_conv(x, weights, gain), emb, p) -- conv_res1 of Unet_block followed by FiLM, mp_silu and dropout (reference
    model_components.py:240-246).  In the bf16 bank path the second step is an extra output of the conv kernel's epilogue."""
    global _FILM_REQ
    p = float(p) if training else 0.0
    C = int((weights[0] if isinstance(weights, (list, tuple)) else weights).shape[0])
    # Only without dropout (eval / sampling), unless forced (HDMOE_CONV_FILM=2): drawing the Philox bits in the conv epilogue -- 8 waves per CU,
    # on every unit's critical path -- costs more than the whole separate pass (measured: conv6<2,1> 44 -> 85 us against a 22-us film kernel).
    ok = (CONV_FILM and (p == 0.0 or CONV_FILM_TRAIN) and x.dtype == torch.bfloat16 and x.ndim == 4 and C % 8 == 0 and 256 % (C // 8) == 0
          and _bank.ACTIVE is not None)
    if not ok:
        return film_silu(mp_conv(x, weights, gain, seg=seg, training=training), emb, p, training)
    req = _FilmReq(_f32(emb), p, _next_seed() if p > 0.0 else 0)
    _FILM_REQ = req
    try:
        y = mp_conv(x, weights, gain, seg=seg, training=training)
    finally:
        _FILM_REQ = None
    if req.h is not None:
        return _FilmDoneFn.apply(y, req.emb, p, req.seed, req.h)
    return _FilmSiluFn.apply(y, req.emb, p, req.seed)


ONES6 = _os.environ.get("HDMOE_ONES6", "1") != "0"          # the 33-channel first conv of Unet_expert on conv6 / wgrad6 (csrc/ones6.hip)
BLK6 = _os.environ.get("HDMOE_BLK6", "1") != "0"
# Which blocks take the fused launch.  Measured per layer shape (tools/blk6_bench.py, graph replay, N = 512 rows, experts [3,3,5,5],
# dropout 0.2; fused vs conv6 + film_silu + conv6): 32 -> 32 at 32x32 108 vs 117 us, 64 -> 32 140 vs 142, 32 -> 32 at 16x16 33 vs 42; but
# 64 -> 64 at 16x16 76 vs 73, 128 -> 64 98 vs 93, 64 -> 64 at 32x32 306 vs 280: with 64 output channels the unit's phases (conv A, middle op,
# conv B) run one after the other in a single 8-wave workgroup and the halo recompute is not paid back.  "c32": 32-channel blocks only.
# Round 4: on 32 x 32 maps with enough routed rows the whole-image streaming kernel (csrc/conv7.hip) runs the two convs + the FiLM pass in
# 2 x 27 + 19 us against the fused launch's 97-108 us, so "c32" leaves those blocks to it (same box: 14.00 -> 13.72 ms / step).
BLK6_SCOPE = _os.environ.get("HDMOE_BLK6_SCOPE", "c32")
CONV7 = _os.environ.get("HDMOE_CONV7", "1") != "0"
C7_MINN = int(_os.environ.get("HDMOE_C7_MINN", "192"))


def unet_block_fused(h: Tensor, res: Optional[Tensor], w1s, w2s, gain1: float, gain2: float, emb: Tensor, p: float, training: bool,
                     seg: Optional[Tensor], alpha: float, beta: float, res_grad_raw: bool = False) -> Optional[Tensor]:
    """Main branch of Unet_block (reference model_components.py:240-253) for a bank of experts as ONE launch (csrc/blk6.hip):
    alpha * conv_res2(dropout(mp_silu(conv_res1(h) * emb))) + beta * res, the activation tile kept in LDS between the two convs.
    Returns None when the fused kernel does not apply (no ready weight-bank entries, fp32 mode, shapes outside its domain): the caller
    then runs the layers one by one.  The autograd graph is the unfused one (conv -> FiLM -> conv nodes over the tensors the fused
    launch wrote), so the backward is unchanged."""
    global _PRECOMP
    w1s, w2s = list(w1s), list(w2s)
    if not (BLK6 and _bank.ACTIVE is not None and _prof_ok() and h.dtype == torch.bfloat16 and h.ndim == 4 and h.is_cuda):
        return None
    if res is not None and (res.dtype != h.dtype or not res.is_contiguous()):
        return None
    ent1 = _bank.ACTIVE.lookup(w1s, h.dtype, float(gain1), 1.0, True)
    ent2 = _bank.ACTIVE.lookup(w2s, h.dtype, float(gain2), float(alpha), True)
    if ent1 is None or ent2 is None or ent1.khs != ent1.kws or ent1.khs != ent2.khs or ent2.khs != ent2.kws:
        return None
    h = _c(h)
    N, H, W, Cin = h.shape
    C = ent1.O
    if ent1.I != Cin or ent2.I != C or ent2.O != C or (seg is None and len(w1s) != 1):
        return None
    if BLK6_SCOPE == "c32" and (C != 32 or (CONV7 and H == 32 and W == 32 and N >= C7_MINN)):
        return None
    p = float(p) if training else 0.0
    e32 = _f32(emb)
    seed = _next_seed() if p > 0.0 else 0
    y = torch.empty((N, H, W, C), dtype=h.dtype, device=h.device)
    need_bwd = torch.is_grad_enabled() and (h.requires_grad or emb.requires_grad or any(w.requires_grad for w in w1s + w2s))
    # (inference: the pre-activation and the activation are not written at all -- two of the launch's three output streams)
    u = torch.empty_like(y) if need_bwd else None
    hb = torch.empty_like(y) if need_bwd else None
    if _timed("fused", dict(name="blk6_kernel (fwd)", dtype="bfloat16", seg=seg, N=N, HW=H * W, O=C, I=Cin + C, taps=[k * k for k in ent1.khs], mult=1.0),
              "hdmoe_unet_block_fwd", h, ent1.wf, ent2.wf, u, hb, y, res, e32, seed, step_counter(h.device), p, float(alpha), float(beta), seg,
              len(w1s), ent1.wstride, ent2.wstride, N, H, W, Cin, C, ent1.khs, _dt(h)) != 0:
        if p > 0.0:
            _seed_state["ctr"] -= 1                            # nothing was launched: the unfused path draws this salt itself
        return None
    STATS["blk"] += 1
    if not need_bwd:
        return y
    ws1 = w1s if seg is not None else w1s[0]
    ws2 = w2s if seg is not None else w2s[0]
    try:
        _PRECOMP = u
        uu = mp_conv(h, ws1, gain1, seg=seg, training=training)
        hh = _FilmDoneFn.apply(uu, e32, p, seed, hb)
        _PRECOMP = y
        return mp_conv(hh, ws2, gain2, seg=seg, res=res, alpha=alpha, beta=beta, training=training, res_grad_raw=res_grad_raw)
    finally:
        _PRECOMP = None


def film_silu(u: Tensor, e: Tensor, p: float = 0.0, training: bool = False) -> Tensor:
    """dropout_p(mp_silu(u * e[n, c])) with e a float32 (N, C) embedding (model_components.py:242-246); the dropout is fused
    into the same pass when the layout is 16-byte vectorisable, otherwise it runs as a separate kernel."""
    p = float(p) if training else 0.0
    C = u.shape[-1]
    vw = 16 // u.element_size()
    if p > 0.0 and not (C % vw == 0 and 256 % (C // vw) == 0):
        return dropout(_FilmSiluFn.apply(u, e, 0.0, 0), p, True)
    return _FilmSiluFn.apply(u, e, p, _next_seed() if p > 0.0 else 0)


class _ScaleRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.pool_ok = not s.is_leaf
        x = _c(x); s = _f32(s).reshape(-1)
        rows = x.shape[0]
        L = x.numel() // rows
        out = torch.empty_like(x)
        call("hdmoe_scale_rows_fwd", out, x, s, rows, L, _dt(x))
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = _c(g)
        rows = x.shape[0]
        L = x.numel() // rows
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ds = _zeros(s.shape, s.dtype, s.device, ctx.pool_ok) if ctx.needs_input_grad[1] else None
        call("hdmoe_scale_rows_bwd", dx, ds, g, x, s, rows, L, _dt(x))
        return dx, ds


def scale_rows(x: Tensor, s: Tensor) -> Tensor:
    """out[n] = s[n] * x[n] with s a float32 (N,) vector."""
    return _ScaleRowsFn.apply(x, s)


class _Cat2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, wa, wb):
        a = _c(a); b = _c(b)
        Ca, Cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // Ca
        out = torch.empty((*a.shape[:-1], Ca + Cb), dtype=a.dtype, device=a.device)
        call("hdmoe_cat2_fwd", out, a, b, wa, wb, Ca, Cb, rows, _dt(a))
        ctx.meta = (wa, wb, a.shape, b.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        wa, wb, sa, sb = ctx.meta
        g = _c(g)
        da = torch.empty(sa, dtype=g.dtype, device=g.device)
        db = torch.empty(sb, dtype=g.dtype, device=g.device)
        call("hdmoe_cat2_bwd", da, db, g, wa, wb, sa[-1], sb[-1], da.numel() // sa[-1], _dt(g))
        return da, db, None, None


def mp_cat(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """Channel-last mp_cat (model_internals.py:69-92)."""
    na, nb = a.shape[-1], b.shape[-1]
    c = math.sqrt((na + nb) / ((1.0 - t) ** 2 + t ** 2))
    return _Cat2Fn.apply(a, b, c * (1.0 - t) / math.sqrt(na), c * t / math.sqrt(nb))


class _SiluBranchFn(torch.autograd.Function):
    """x -> (x, mp_silu(x)) for a tensor that feeds mp_silu and a second consumer (decoder-block input: skip / residual path).
    Backward: ONE pass, dx = gx + gh * mp_silu'(x) (instead of mp_silu_bwd + a gradient-sum launch)."""

    @staticmethod
    def forward(ctx, x, gx_scale):
        ctx.set_materialize_grads(False)
        x = _c(x)
        h = torch.empty_like(x)
        call("hdmoe_mp_silu_fwd", h, x, x.numel(), _dt(x))
        ctx.save_for_backward(x)
        ctx.sx = float(gx_scale)
        return x.view(x.shape), h

    @staticmethod
    def backward(ctx, gx, gh):
        (x,) = ctx.saved_tensors
        if gh is None:
            if gx is None or ctx.sx == 1.0:
                return gx, None
            dx = torch.empty_like(x)
            call("hdmoe_axpby", dx, _c(gx), None, ctx.sx, 0.0, dx.numel(), _dt(dx))
            return dx, None
        gh = _c(gh)
        dx = torch.empty_like(x)
        if gx is None:
            call("hdmoe_mp_silu_bwd", dx, gh, x, x.numel(), _dt(x))
        else:
            call("hdmoe_mp_silu_bwd_add", dx, gh, x, _c(gx), ctx.sx, x.numel(), _dt(x))
        return dx, None


class _CatSiluFn(torch.autograd.Function):
    """(mp_cat(a, b), mp_silu(mp_cat(a, b))) in one pass; backward in one pass as well."""

    @staticmethod
    def forward(ctx, a, b, wa, wb):
        ctx.set_materialize_grads(False)
        a = _c(a); b = _c(b)
        Ca, Cb = a.shape[-1], b.shape[-1]
        out = torch.empty((*a.shape[:-1], Ca + Cb), dtype=a.dtype, device=a.device)
        h = torch.empty_like(out)
        call("hdmoe_cat2_silu_fwd", out, h, a, b, wa, wb, Ca, Cb, a.numel() // Ca, _dt(a))
        ctx.save_for_backward(out)
        ctx.meta = (wa, wb, a.shape, b.shape)
        return out, h

    @staticmethod
    def backward(ctx, gcat, gh):
        (out,) = ctx.saved_tensors
        wa, wb, sa, sb = ctx.meta
        da = torch.empty(sa, dtype=out.dtype, device=out.device)
        db = torch.empty(sb, dtype=out.dtype, device=out.device)
        if gh is None:
            if gcat is None:
                return None, None, None, None
            call("hdmoe_cat2_bwd", da, db, _c(gcat), wa, wb, sa[-1], sb[-1], da.numel() // sa[-1], _dt(out))
        else:
            call("hdmoe_cat2_silu_bwd", da, db, None if gcat is None else _c(gcat), _c(gh), out, wa, wb, sa[-1], sb[-1], da.numel() // sa[-1], _dt(out))
        return da, db, None, None


def _vec_ok(*ts) -> bool:
    return all(t.shape[-1] % (8 if t.dtype == torch.bfloat16 else 4) == 0 and t.dtype in (torch.bfloat16, torch.float32) for t in ts)


def silu_branch(x: Tensor, gx_scale: float = 1.0):
    """(x, mp_silu(x)): the pair a decoder block needs (x continues on the skip / residual path).  ``gx_scale``: factor applied to
    the gradient arriving for the returned x (see mp_conv(res_grad_raw=True))."""
    if _vec_ok(x) and x.numel() % 8 == 0:
        return _SiluBranchFn.apply(x, float(gx_scale))
    x, xh = fanout(x, 2)
    if gx_scale != 1.0:
        x = affine_grad(x, gx_scale)
    return x, mp_silu(xh)


class _GradScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        dx = torch.empty_like(g)
        call("hdmoe_axpby", dx, g, None, ctx.s, 0.0, g.numel(), _dt(g))
        return dx, None


def affine_grad(x: Tensor, s: float) -> Tensor:
    """Identity in the forward, gradient times ``s`` in the backward (the unfused form of a deferred residual-gradient scale)."""
    return _GradScaleFn.apply(x, float(s))


def mp_cat_silu(a: Tensor, b: Tensor, t: float = 0.5):
    """(mp_cat(a, b, t), mp_silu of it) in one pass (model_internals.py:69-92 followed by :33-36)."""
    if not (_vec_ok(a, b) and a.dtype == b.dtype):
        return silu_branch(mp_cat(a, b, t))
    na, nb = a.shape[-1], b.shape[-1]
    c = math.sqrt((na + nb) / ((1.0 - t) ** 2 + t ** 2))
    return _CatSiluFn.apply(a, b, c * (1.0 - t) / math.sqrt(na), c * t / math.sqrt(nb))


class _ResampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = _c(x)
        N, H, W, C = x.shape
        ctx.mode = mode
        if mode == "down":
            out = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
            call("hdmoe_pool2", out, x, N, H // 2, W // 2, C, 0.25, _dt(x))
        else:
            out = torch.empty((N, H * 2, W * 2, C), dtype=x.dtype, device=x.device)
            call("hdmoe_upsample2", out, x, N, H * 2, W * 2, C, 1.0, _dt(x))
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        N, H, W, C = g.shape
        if ctx.mode == "down":
            dx = torch.empty((N, H * 2, W * 2, C), dtype=g.dtype, device=g.device)
            call("hdmoe_upsample2", dx, g, N, H * 2, W * 2, C, 0.25, _dt(g))
        else:
            dx = torch.empty((N, H // 2, W // 2, C), dtype=g.dtype, device=g.device)
            call("hdmoe_pool2", dx, g, N, H // 2, W // 2, C, 1.0, _dt(g))
        return dx, None


class _FirResampleFn(torch.autograd.Function):
    """resample(x, f, mode) for an even-length filter other than [1, 1] (model_internals.py:95-127): depthwise stride-2 correlation
    with outer(f, f) / f.sum()^2 ('down') or its transpose with 4x the taps ('up'); each is the other's backward."""

    @staticmethod
    def forward(ctx, x, taps, mode):
        x = _c(x)
        N, H, W, C = x.shape
        L = len(taps)
        pad = (L - 1) // 2
        if mode == "down":
            Ho, Wo = (H + 2 * pad - L) // 2 + 1, (W + 2 * pad - L) // 2 + 1
            out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
            call("hdmoe_fir_resample", out, x, taps, L, pad, 1.0, 0, N, H, W, Ho, Wo, C, _dt(x))
        else:
            Ho, Wo = (H - 1) * 2 - 2 * pad + L, (W - 1) * 2 - 2 * pad + L
            out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
            call("hdmoe_fir_resample", out, x, taps, L, pad, 4.0, 1, N, H, W, Ho, Wo, C, _dt(x))
        ctx.meta = (taps, mode, L, pad, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        taps, mode, L, pad, H, W = ctx.meta
        g = _c(g)
        N, Hg, Wg, C = g.shape
        dx = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
        if mode == "down":                                     # transpose of the strided correlation (no 4x)
            call("hdmoe_fir_resample", dx, g, taps, L, pad, 1.0, 1, N, Hg, Wg, H, W, C, _dt(g))
        else:
            call("hdmoe_fir_resample", dx, g, taps, L, pad, 4.0, 0, N, Hg, Wg, H, W, C, _dt(g))
        return dx, None, None


def resample(x: Tensor, mode: str = "keep", f=(1, 1)) -> Tensor:
    """resample (model_internals.py:95-127), channel-last.  f = [1, 1]: 'down' = 2x2 mean, 'up' = nearest x2 (vector kernels);
    any other even-length filter with up to 8 taps: the generic separable FIR kernels."""
    if mode == "keep":
        return x
    if mode not in ("down", "up"):
        raise ValueError(f"Invalid mode: {mode}")
    f = [float(v) for v in f]
    if len(f) % 2 or not f:
        raise AssertionError("resample: the filter must be 1-D with an even number of taps")
    if f == [1.0, 1.0] and not (mode == "down" and (x.shape[1] % 2 or x.shape[2] % 2)):
        return _ResampleFn.apply(x, mode)
    if len(f) > 8:
        raise NotImplementedError("resample: filters with more than 8 taps are not implemented")
    tot = sum(f)
    return _FirResampleFn.apply(x, tuple(v / tot for v in f), mode)


class _SeqReduceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        N, C = x.shape[0], x.shape[-1]
        S = x.numel() // (N * C)
        out = torch.zeros((N, C), dtype=torch.float32, device=x.device)
        call("hdmoe_seq_reduce", out, x, N, S, C, scale, _dt(x))
        ctx.meta = (x.shape, x.dtype, scale, S)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, dt, scale, S = ctx.meta
        g = _f32(g)
        dx = torch.empty(shape, dtype=dt, device=g.device)
        call("hdmoe_seq_bcast_add", dx, None, g, shape[0], S, shape[-1], scale, dtype_code(dt))
        return dx, None


def seq_mean(x: Tensor) -> Tensor:
    """Mean over all middle dims: (N, ..., C) -> float32 (N, C)  (AdaptiveAvgPool2d(1) / text.mean(1))."""
    N, C = x.shape[0], x.shape[-1]
    return _SeqReduceFn.apply(x, 1.0 / (x.numel() // (N * C)))


class _SeqBcastAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        ctx.pool_ok = not t.is_leaf
        x = _c(x); t = _f32(t)
        N, C = x.shape[0], x.shape[-1]
        S = x.numel() // (N * C)
        out = torch.empty_like(x)
        call("hdmoe_seq_bcast_add", out, x, t, N, S, C, 1.0, _dt(x))
        ctx.meta = (N, S, C)
        return out

    @staticmethod
    def backward(ctx, g):
        N, S, C = ctx.meta
        g = _c(g)
        dt_ = None
        if ctx.needs_input_grad[1]:
            dt_ = _zeros((N, C), torch.float32, g.device, ctx.pool_ok)
            call("hdmoe_seq_reduce", dt_, g, N, S, C, 1.0, _dt(g))
        return g, dt_


def seq_bcast_add(x: Tensor, t: Tensor) -> Tensor:
    """x[n, s, :] + t[n, :] (the q/k/v time biases of MP_Attention, model_internals.py:368-372)."""
    return _SeqBcastAddFn.apply(x, t)


class _BiasAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        x = _c(x); bias = _f32(bias)
        L = bias.numel()
        out = torch.empty_like(x)
        call("hdmoe_bias_add", out, x, bias, x.numel() // L, L, _dt(x))
        ctx.bshape = bias.shape
        ctx.bias_param = bias if (bias.is_leaf and bias.requires_grad) else None
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        db = None
        if ctx.needs_input_grad[1]:
            L = 1
            for d in ctx.bshape:
                L *= d
            bp = ctx.bias_param
            if _direct(bp):
                call("hdmoe_colsum", bp.grad, g, g.numel() // L, L, _dt(g))
            else:
                db = torch.zeros(ctx.bshape, dtype=torch.float32, device=g.device)
                call("hdmoe_colsum", db, g, g.numel() // L, L, _dt(g))
        return g, db


def bias_add(x: Tensor, bias: Tensor) -> Tensor:
    """x viewed as (rows, bias.numel()) + bias (pos_emb add, model_components.py:680)."""
    return _BiasAddFn.apply(x, bias)


class _LerpParamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha):
        a = _c(a); b = _c(b)
        out = torch.empty_like(a)
        call("hdmoe_lerp_param_fwd", out, a, b, alpha, a.numel(), _dt(a))
        ctx.save_for_backward(a, b, alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, alpha = ctx.saved_tensors
        g = _c(g)
        da, db = torch.empty_like(a), torch.empty_like(b)
        direct = _direct(alpha)
        dal = alpha.grad if direct else torch.zeros_like(alpha)
        call("hdmoe_lerp_param_bwd", da, db, dal, g, a, b, alpha, a.numel(), _dt(a))
        return da, db, (None if direct else dal)


def lerp_param(a: Tensor, b: Tensor, alpha: Tensor) -> Tensor:
    """a + alpha*(b - a) with a learnable float32 scalar (model_config2.py:291)."""
    return _LerpParamFn.apply(a, b, alpha)


class _GateMixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, U, A):
        logits = _c(logits); U = _c(U); A = _c(A)
        C = U.shape[-1]
        rows = U.numel() // C
        out = torch.empty_like(U)
        gate = torch.empty((*U.shape[:-1], 2), dtype=torch.float32, device=U.device)
        call("hdmoe_gate_mix_fwd", out, gate, logits, U, A, rows, C, _dt(U))
        ctx.save_for_backward(gate, U, A)
        return out, gate

    @staticmethod
    def backward(ctx, g, dgate):
        gate, U, A = ctx.saved_tensors
        g = _c(g)
        dgate = None if dgate is None else _f32(dgate)
        C = U.shape[-1]
        rows = U.numel() // C
        dU, dA = torch.empty_like(U), torch.empty_like(A)
        dl = torch.empty((*U.shape[:-1], 2), dtype=U.dtype, device=U.device)
        call("hdmoe_gate_mix_bwd", dU, dA, dl, g, dgate, gate, U, A, rows, C, _dt(U))
        return dl, dU, dA


def gate_mix(logits: Tensor, U: Tensor, A: Tensor):
    """softmax gate over 2 channels + mix + mp_sum (model_config2.py:297-301).  Returns (out, gate fp32)."""
    return _GateMixFn.apply(logits, U, A)


class _SoftmaxRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = _f32(x)
        out = torch.empty_like(x)
        call("hdmoe_softmax_rows_fwd", out, x, x.numel() // x.shape[-1], x.shape[-1], scale)
        ctx.save_for_backward(out)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _f32(g)
        dx = torch.empty_like(y)
        call("hdmoe_softmax_rows_bwd", dx, g, y, y.numel() // y.shape[-1], y.shape[-1], ctx.scale)
        return dx, None


def softmax_rows(x: Tensor, scale: float = 1.0) -> Tensor:
    return _SoftmaxRowsFn.apply(x, float(scale))


class _AdaLNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cond):
        x = _f32(x); cond = _f32(cond)
        B, F = x.shape
        out = torch.empty_like(x)
        call("hdmoe_adaln_fwd", out, x, cond, B, F)
        ctx.save_for_backward(x, cond)
        return out

    @staticmethod
    def backward(ctx, g):
        x, cond = ctx.saved_tensors
        g = _f32(g)
        B, F = x.shape
        dx, dcond = torch.empty_like(x), torch.empty_like(cond)
        call("hdmoe_adaln_bwd", dx, dcond, g, x, cond, B, F)
        return dx, dcond


def adaln(x: Tensor, cond: Tensor) -> Tensor:
    """x*(1+gamma)+beta with cond = [gamma | beta] (B, 2F) (model_components.py:148-151)."""
    return _AdaLNFn.apply(x, cond)


class _TakeColPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, e):
        w = _f32(w)
        B, E = w.shape
        out = torch.empty(B, dtype=torch.float32, device=w.device)
        call("hdmoe_take_col_pos_fwd", out, w, B, E, e)
        ctx.save_for_backward(w)
        ctx.e = e
        return out

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        B, E = w.shape
        dw = torch.zeros_like(w)
        call("hdmoe_take_col_pos_bwd", dw, _f32(g), w, B, E, ctx.e)
        return dw, None


def take_col_pos(w: Tensor, e: int) -> Tensor:
    """(B,) float32: w[:, e] where positive, else 0 (the `out_router[:, i] > 0` mask, model_config1.py:26,35)."""
    return _TakeColPosFn.apply(w, int(e))


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _c(x)
        out = torch.empty_like(x)
        call("hdmoe_dropout", out, x, seed, step_counter(x.device), p, x.numel(), _dt(x))
        ctx.meta = (p, seed)
        return out

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.meta
        g = _c(g)
        dx = torch.empty_like(g)
        call("hdmoe_dropout", dx, g, seed, step_counter(g.device), p, g.numel(), _dt(g))
        return dx, None, None


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    if not training or p == 0.0:
        return x
    return _DropoutFn.apply(x, float(p), _next_seed())


def randn_like(x: Tensor, scale: float) -> Tensor:
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("hdmoe_randn", out, _next_seed(), step_counter(x.device), float(scale), out.numel())
    return out


# =====================================================================================================
# layout / boundary
# =====================================================================================================
class _ToNHWCFn(torch.autograd.Function):
    """float32 NCHW (contiguous) -> NHWC in `dtype`, optionally scaled per sample."""

    @staticmethod
    def forward(ctx, x, s, dtype):
        x = _c(x)
        N, C, H, W = x.shape
        out = torch.empty((N, H, W, C), dtype=dtype, device=x.device)
        call("hdmoe_nchw_to_nhwc", out, x, s, N, C, H * W, dtype_code(dtype))
        ctx.save_for_backward(s)
        return out

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        g = _c(g)
        N, H, W, C = g.shape
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        call("hdmoe_nhwc_to_nchw", dx, g, s, None, None, N, C, H * W, _dt(g))
        return dx, None, None


class _FromNHWCFn(torch.autograd.Function):
    """out_nchw(fp32) = sf[n]*F_nhwc + sx[n]*x_nchw   (EDM D_x, model_config2.py:449)."""

    @staticmethod
    def forward(ctx, F, sf, x, sx):
        F = _c(F)
        N, H, W, C = F.shape
        x = None if x is None else _c(x)
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=F.device)
        call("hdmoe_nhwc_to_nchw", out, F, sf, x, sx, N, C, H * W, _dt(F))
        ctx.save_for_backward(sf, sx)
        ctx.meta = (F.dtype, x is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        sf, sx = ctx.saved_tensors
        fdt, has_x = ctx.meta
        g = _c(g)
        N, C, H, W = g.shape
        dF = torch.empty((N, H, W, C), dtype=fdt, device=g.device)
        call("hdmoe_nchw_to_nhwc", dF, g, sf, N, C, H * W, dtype_code(fdt))
        dx = None
        if has_x and ctx.needs_input_grad[2]:
            dx = torch.empty_like(g)
            if sx is None:
                call("hdmoe_axpby", dx, g, None, 1.0, 0.0, g.numel(), 0)
            else:
                call("hdmoe_scale_rows_fwd", dx, g, sx, N, C * H * W, 0)
        return dF, None, dx, None


def to_nhwc(x: Tensor, scale: Optional[Tensor] = None, dtype: Optional[torch.dtype] = None) -> Tensor:
    """Module-boundary ingest: logical NCHW -> contiguous (N,H,W,C)."""
    dtype = dtype or x.dtype
    if scale is None and dtype == x.dtype and x.is_contiguous(memory_format=torch.channels_last):
        return x.permute(0, 2, 3, 1)                      # already channel-last in memory: a view
    if x.dtype != torch.float32:
        # non-fp32 NCHW input at a module boundary: relayout only (no arithmetic)
        y = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        return cast(y, dtype) if scale is None else scale_rows(cast(y, dtype), scale)
    return _ToNHWCFn.apply(x, scale, dtype)


def from_nhwc(y: Tensor) -> Tensor:
    """(N,H,W,C) -> logical (N,C,H,W) (channels_last strides, zero-copy)."""
    return y.permute(0, 3, 1, 2)


def nhwc_to_nchw_f32(F: Tensor, sf: Optional[Tensor] = None, x: Optional[Tensor] = None, sx: Optional[Tensor] = None) -> Tensor:
    return _FromNHWCFn.apply(F, sf, x, sx)


class _PatchRelayoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, meta):
        N, H, W, C, p, hp, wp, order = meta
        tok = _c(tok)
        out = torch.empty((N, H, W, C), dtype=tok.dtype, device=tok.device)
        call("hdmoe_patch_relayout", out, tok, N, H, W, C, p, hp, wp, order, 1, _dt(tok))
        ctx.meta = meta
        ctx.tshape = tok.shape
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C, p, hp, wp, order = ctx.meta
        g = _c(g)
        # tokens cover the padded hp*p x wp*p canvas; positions outside H x W receive no gradient
        dtok = torch.zeros(ctx.tshape, dtype=g.dtype, device=g.device) if (hp * p != H or wp * p != W) else \
            torch.empty(ctx.tshape, dtype=g.dtype, device=g.device)
        call("hdmoe_patch_relayout", dtok, g, N, H, W, C, p, hp, wp, order, 0, _dt(g))
        return dtok, None


def pixel_shuffle_tokens(tok: Tensor, H: int, W: int, C: int, p: int) -> Tensor:
    """tokens (N, hp*wp, C*p*p) in PixelShuffle feature order -> image (N,H,W,C), cropped (model_components.py:698-704)."""
    N = tok.shape[0]
    hp, wp = -(-H // p), -(-W // p)
    return _PatchRelayoutFn.apply(tok, (N, H, W, C, p, hp, wp, 1))


# =====================================================================================================
# norms
# =====================================================================================================
class _PixelNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, with_silu, gx_scale=1.0):
        ctx.sx = float(gx_scale)
        x = _c(x)
        C = x.shape[-1]
        xn = torch.empty_like(x)
        h = torch.empty_like(x) if with_silu else None
        call("hdmoe_pixelnorm_fwd", xn, h, x, x.numel() // C, C, _dt(x))
        ctx.save_for_backward(x)
        ctx.with_silu = with_silu
        if with_silu:
            return xn, h
        return xn

    @staticmethod
    def backward(ctx, dxn, dh=None):
        (x,) = ctx.saved_tensors
        C = x.shape[-1]
        dx = torch.empty_like(x)
        call("hdmoe_pixelnorm_bwd", dx, _c(dxn), _c(dh), x, x.numel() // C, C, ctx.sx, _dt(x))
        return dx, None, None


def pixel_norm(x: Tensor) -> Tensor:
    """normalize(x, dim=[channel]) (model_internals.py:8-30)."""
    return _PixelNormFn.apply(x, False, 1.0)


def pixel_norm_silu(x: Tensor, gx_scale: float = 1.0):
    """Returns (normalize(x, channel), mp_silu(of that)) in one pass (model_components.py:238-240).  ``gx_scale``: factor applied to
    the gradient arriving for the FIRST output (see mp_conv(res_grad_raw=True))."""
    return _PixelNormFn.apply(x, True, float(gx_scale))


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, act, eps):
        x = _c(x)
        N, C = x.shape[0], x.shape[-1]
        S = x.numel() // (N * C)
        mean = torch.empty(N * G, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        parts = _gn_parts(N, S)
        if parts > 1:
            ws = torch.empty(2 * N * parts * G, dtype=torch.float32, device=x.device)
            call("hdmoe_groupnorm_fwd_split", y, mean, rstd, ws, parts, x, gamma, beta, N, S, C, G, act, eps, _dt(x))
        else:
            call("hdmoe_groupnorm_fwd", y, mean, rstd, x, gamma, beta, N, S, C, G, act, eps, _dt(x))
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.meta = (N, S, C, G, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        N, S, C, G, act = ctx.meta
        g = _c(g)
        dx = torch.empty_like(x)
        direct = _direct(gamma) and _direct(beta)
        dgamma = gamma.grad if direct else torch.zeros_like(gamma)
        dbeta = beta.grad if direct else torch.zeros_like(beta)
        parts = _gn_parts(N, S, forward=False)
        if parts > 1:
            ws = _zeros((2 * N * G,), torch.float32, x.device)
            call("hdmoe_groupnorm_bwd_split", dx, dgamma, dbeta, ws, parts, g, x, gamma, beta, mean, rstd, N, S, C, G, act, _dt(x))
        else:
            ws = torch.empty(2 * N * G, dtype=torch.float32, device=x.device)
            call("hdmoe_groupnorm_bwd", dx, dgamma, dbeta, ws, g, x, gamma, beta, mean, rstd, N, S, C, G, act, _dt(x))
        if direct:
            return dx, None, None, None, None, None
        return dx, dgamma, dbeta, None, None, None


def _gn_parts(N: int, S: int, forward: bool = True) -> int:
    """Row-range workgroups per sample for the GroupNorm statistics.  Forward: a function of S only -- the summation order, and
    with it every bit of the output, must not depend on how many samples share the batch.  Backward (which sums with float
    atomics anyway): only when one block per sample would leave the chip idle."""
    if forward:
        return max(1, min(4, S // 256))
    if N >= 128 or S < 256:
        return 1
    return max(1, min(64, 256 // N, S // 128))


ACT_NONE, ACT_RELU, ACT_MP_SILU = 0, 1, 2


def group_norm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, act: int = ACT_NONE, eps: float = 1e-5) -> Tensor:
    """nn.GroupNorm over a channel-last tensor (N, ..., C), optionally fused with ReLU / mp_silu."""
    return _GroupNormFn.apply(x, gamma, beta, int(groups), int(act), float(eps))


TRUNK_FUSED = _os.environ.get("HDMOE_TRUNK_FUSED", "1") != "0"
# bf16 compute mode only (the split kernels): the router trunks' BACKWARD (input + weight gradients of their convs) uses the hi halves of
# the split operands only -- bf16 operands, fp32 accumulation, one MFMA per product instead of three.  Bit-exact routing indices need
# the fp32-equivalent FORWARD (three products); gradients carry the bf16 mode's tolerance like every expert layer.  0: three products.
TRUNK_BWD_BF16 = _os.environ.get("HDMOE_TRUNK_BWD_BF16", "1") != "0"
# ... and, in that mode, as plain bf16 layers on the streaming kernels (round 4; 0: the split kernels' hi-only path, csrc/bwd6.hip bwd6s)
TRUNK_BWD7 = _os.environ.get("HDMOE_TRUNK_BWD7", "1") != "0"


class _TrunkFn(torch.autograd.Function):
    """Router.hard_route up to the average pool (reference model_components.py:100-112): three [MP_Conv 3x3 -> GroupNorm(1, C) -> ReLU]
    and AdaptiveAvgPool2d(1) over an fp32 channel-last tensor, with the GroupNorm + ReLU folded into the convs on either side
    (split-bf16 arithmetic, csrc/conv6s.hip):
      conv_l writes y_l and per-sample partial statistics of y_l -> hdmoe_gn1_finalize -> scale_l / shift_l [N][C];
      conv_{l+1} (and, in the backward, its weight gradient) stage relu(y_l * scale_l + shift_l) instead of a stored activation;
      the last GroupNorm + ReLU + pool is one read of y_3 (hdmoe_gn1_relu_mean).
    The normalised activations are never written.  ``tensors`` = (weight, gamma, beta) x 3; ``ents`` the ready weight-bank entries."""

    @staticmethod
    def forward(ctx, x, ents, bank, eps, *tensors):
        from ._lib import lib
        x = _c(x)
        N, H, W, _ = x.shape
        S = H * W
        inp, sc, sh = x, None, None
        saved = []
        for l in range(3):
            w, gamma, beta = tensors[3 * l:3 * l + 3]
            O, I = int(w.shape[0]), int(w.shape[1])
            ent = ents[l]
            slots = lib().hdmoe_conv_split_stats_slots(H, W, O)
            y = torch.empty((N, H, W, O), dtype=torch.float32, device=x.device)
            ws = torch.empty((N, max(slots, 1), 2), dtype=torch.float32, device=x.device)
            if slots < 1 or _timed("fused", dict(name=f"conv6_split_kernel<{2 if O % 64 == 0 else 1}> (GroupNorm folded)", dtype="split_bf16", seg=None, N=N, HW=S, O=O, I=I, taps=[9], mult=1.0),
                                   "hdmoe_conv_fwd_split_gn", inp, ent.wf, y, sc, sh, 1, ws, ent.wstride, ent.wstride, N, H, W, I, O, 1.0) != 0:
                raise RuntimeError("router trunk: layer outside the split conv kernel's domain (ops.trunk_ok should have said so)")
            sc = torch.empty((N, O), dtype=torch.float32, device=x.device)
            sh = torch.empty_like(sc)
            mean = torch.empty(N, dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            if l < 2:
                # (folding this step into the consuming conv as well -- every workgroup re-deriving mean / rstd from the slots in fp64 --
                #  was measured 0.08 ms/step SLOWER: register spills in the 128-channel kernel; tried and removed)
                call("hdmoe_gn1_finalize", sc, sh, mean, rstd, ws, gamma, beta, N, slots, O, S * O, eps[l])
            else:                                               # last layer: statistics -> scale / shift inside the pooled read's launch
                out = torch.empty((N, O), dtype=torch.float32, device=x.device)
                call("hdmoe_gn1_finalize_relu_mean", out, sc, sh, mean, rstd, y, ws, gamma, beta, N, slots, S, O, eps[l])
            saved += [y, sc, sh, mean, rstd]
            inp = y
        ctx.save_for_backward(x, *saved, *tensors)
        ctx.ents, ctx.bank = ents, bank
        STATS["trunk"] += 1
        return out

    @staticmethod
    def backward(ctx, g):
        from ._lib import lib, _int_array
        import ctypes
        x, *rest = ctx.saved_tensors
        saved, tensors = rest[:15], rest[15:]
        N, H, W, _ = x.shape
        S = H * W
        ents, bank = ctx.ents, ctx.bank
        params = [t for l in range(3) for t in tensors[3 * l + 1:3 * l + 3]]
        bufs, ret = _param_grads(params)
        k3 = ctypes.cast(_int_array([3]), ctypes.c_void_p)
        da = None
        # bf16-operand mode on 32 x 32 maps with enough samples: the trunk backward as PLAIN bf16 layers on the streaming kernels (csrc/conv7_body.h,
        # wgrad7_body.h) -- the GroupNorm backward writes dy_l in bf16, a small pass materialises the conv input relu(gn(y_{l-1})) in bf16
        # (the forward never stores it), and the fused dgrad + weight-gradient launch reads both by LDS-DMA.  Same arithmetic as the hi-only
        # split path (bf16 operands, fp32 accumulation) except that the input gradient between two layers is stored in bf16.
        use7 = (TRUNK_BWD7 and TRUNK_BWD_BF16 and CONV7 and H == 32 and W == 32 and N >= C7_MINN
                and all(int(tensors[3 * l].shape[0]) % 32 == 0 and int(tensors[3 * l].shape[1]) % 32 == 0 for l in range(3)))
        for l in (2, 1, 0):
            w, gamma, beta = tensors[3 * l:3 * l + 3]
            y, sc, sh, mean, rstd = saved[5 * l:5 * l + 5]
            O, I = int(w.shape[0]), int(w.shape[1])
            ent = ents[l]
            xin = x if l == 0 else saved[5 * (l - 1)]
            isc, ish = (None, None) if l == 0 else (saved[5 * (l - 1) + 1], saved[5 * (l - 1) + 2])
            wdstride = 9 * I * ((O + 15) // 16 * 16)
            ws = torch.empty(2 * N, dtype=torch.float32, device=x.device)
            if use7:
                dyb = torch.empty(y.shape, dtype=torch.bfloat16, device=x.device)
                call("hdmoe_gn1t_bwd", dyb, bufs[2 * l], bufs[2 * l + 1], ws, None if l == 2 else da, _f32(g) if l == 2 else None,
                     1.0 / S if l == 2 else 1.0, y, gamma, beta, mean, rstd, N, S, O)
                a_in = torch.empty(xin.shape, dtype=torch.bfloat16, device=x.device)
                call("hdmoe_gn1t_act", a_in, xin, isc, ish, N, S, I)
                kib = lib().hdmoe_conv_wgrad6_ws_kib(1, N, H, W, I, O, k3, k3, 1)
                arena = _w6_arena_take(x.device, 2 * kib * 256) if kib > 0 else None
                if arena is None and kib > 0 and not torch.cuda.is_current_stream_capturing():
                    arena = torch.empty(2 * kib * 256, dtype=torch.float32, device=x.device)
                if arena is None:
                    raise RuntimeError("router trunk backward: no weight-gradient workspace (arena exhausted inside a graph capture)")
                da = torch.empty(xin.shape, dtype=torch.bfloat16, device=x.device)
                if _timed("fused", dict(name="bwd7_trunk_kernel", dtype="bfloat16", seg=None, N=N, HW=S, O=O, I=I, taps=[9], mult=2.0),
                          "hdmoe_conv_bwd6", a_in, dyb, ent.wd, da, list(ent.G), None, 1, wdstride, N, H, W, I, O, [3], [3], [1], [1], 1.0,
                          arena, arena.numel() * 4, 1) != 0:
                    raise RuntimeError("router trunk backward: layer outside the streaming backward kernels' domain")
                bank.defer_w6(list(ent.G), None, arena, [1, N, H, W, I, O, 1, 0, 3, 0, 0, 0, 0, 0, 0, 0])
                bank.note_backward(ent)
                STATS["trunk_bwd"] += 1
                STATS["trunk_bwd7"] += 1
                if l == 0:                                     # the stem features are fp32: so is their gradient
                    da32 = torch.empty(xin.shape, dtype=torch.float32, device=x.device)
                    call("hdmoe_cast", da32, da, da.numel(), 1, 0)
                    da = da32
                continue
            dy = torch.empty_like(y)
            if l == 2:      # d(mean over S of a_3): g[n][c] / S at every position, read from the (N, C) tensor (no materialised broadcast)
                call("hdmoe_groupnorm_bwd_bcast", dy, bufs[2 * l], bufs[2 * l + 1], ws, _f32(g), 1.0 / S, y, gamma, beta, mean, rstd, N, S, O, 1, ACT_RELU, 0)
            else:
                call("hdmoe_groupnorm_bwd", dy, bufs[2 * l], bufs[2 * l + 1], ws, da, y, gamma, beta, mean, rstd, N, S, O, 1, ACT_RELU, 0)
            kib = lib().hdmoe_conv_wgrad6_ws_kib(1, N, H, W, I, O, k3, k3, F32S)
            arena = _w6_arena_take(x.device, 2 * kib * 256) if kib > 0 else None
            if arena is None and kib > 0 and not torch.cuda.is_current_stream_capturing():
                arena = torch.empty(2 * kib * 256, dtype=torch.float32, device=x.device)      # arena exhausted (it grows at the next step): own slabs
            if arena is None:
                raise RuntimeError("router trunk backward: no weight-gradient workspace (arena exhausted inside a graph capture)")
            da = torch.empty_like(xin)
            if _timed("fused", dict(name="bwd6s_kernel", dtype="bfloat16" if TRUNK_BWD_BF16 else "split_bf16", seg=None, N=N, HW=S, O=O, I=I, taps=[9], mult=2.0),
                      "hdmoe_conv_bwd6s", xin, dy, ent.wd, da, list(ent.G), None, 1, wdstride, wdstride, N, H, W, I, O, [3], [3], [1], [1], 1.0,
                      arena, arena.numel() * 4, isc, ish, 1, 1 if TRUNK_BWD_BF16 else 0) != 0:
                raise RuntimeError("router trunk backward: layer outside the fused backward kernel's domain")
            bank.defer_w6(list(ent.G), None, arena, [1, N, H, W, I, O, F32S, 0, 3, 0, 0, 0, 0, 0, 0, 0])
            bank.note_backward(ent)
            STATS["trunk_bwd"] += 1
        out = [da, None, None, None]
        for l in range(3):
            out += [None, ret[2 * l], ret[2 * l + 1]]
        return tuple(out)


def trunk_ok(x: Tensor, convs) -> bool:
    """Can Router.hard_route take the fused path?  bf16 compute mode with split-bf16 trunks, every conv a ready weight-bank entry inside the
    split kernels' domain, and the deferred weight-gradient path on."""
    if not (TRUNK_FUSED and ROUTER_SPLIT and BWD6 and W6_DEFER and _prof_ok() and _bank.ACTIVE is not None):
        return False
    if x.dtype != torch.float32 or x.ndim != 4:
        return False
    c = x.shape[-1]
    for w in convs:
        probe = torch.empty((1, x.shape[1], x.shape[2], c), dtype=torch.float32, device="meta")
        if not _split_ok(probe, [w], False) or int(w.shape[1]) != c:
            return False
        c = int(w.shape[0])
        # every layer's GroupNorm backward runs on the 16-byte-vector kernels only (the pooled-gradient form hdmoe_groupnorm_bwd_bcast has no
        # scalar fallback): csrc/norm.hip gn_vec_ok -- C / 4 lanes per pixel must divide the 512-thread block
        cv = c // 4
        if c % 4 or cv > 512 or 512 % cv:
            return False
    return c <= 1024                                         # hdmoe_gn1_finalize_relu_mean: one pixel row of <= 1024 channels per pass


def router_trunk(x: Tensor, convs, norms) -> Optional[Tensor]:
    """Fused Router.hard_route[0:10] -> (N, 4C) fp32, or None when the weight bank has not prepared the layers yet (the caller then
    runs the layers one by one; the lookup registers them for the next step)."""
    ents = [_bank.ACTIVE.lookup([w], "split", 1.0, 1.0, True) for w in convs]
    if any(e is None for e in ents):
        return None
    tensors = []
    for w, nm in zip(convs, norms):
        tensors += [w, nm.weight, nm.bias]
    return _TrunkFn.apply(x, ents, _bank.ACTIVE, [float(nm.eps) for nm in norms], *tensors)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        C = x.shape[-1]
        rows = x.numel() // C
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        call("hdmoe_layernorm_fwd", y, mean, rstd, x, gamma, beta, rows, C, eps, _dt(x))
        ctx.save_for_backward(x, gamma, mean, rstd, beta)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd, beta = ctx.saved_tensors
        g = _c(g)
        C = x.shape[-1]
        dx = torch.empty_like(x)
        direct = _direct(gamma) and _direct(beta)
        dgamma = gamma.grad if direct else torch.zeros_like(gamma)
        dbeta = beta.grad if direct else torch.zeros_like(gamma)
        call("hdmoe_layernorm_bwd", dx, dgamma, dbeta, g, x, gamma, mean, rstd, x.numel() // C, C, _dt(x))
        if direct:
            return dx, None, None, None
        return dx, dgamma, dbeta, None


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    return _LayerNormFn.apply(x, gamma, beta, float(eps))


# =====================================================================================================
# attention core
# =====================================================================================================
class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, bias, H):
        q = _c(q); k = _c(k); v = _c(v)
        B, Sq, E = q.shape
        Skv = k.shape[1]
        D = E // H
        Sb = 0
        if bias is not None:
            bias = _f32(bias)
            Sb = bias.shape[-1]
        out = torch.empty_like(q)
        lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
        _timed("attn", dict(dir="fwd", B=B, Sq=Sq, Skv=Skv, H=H, D=D, esz=q.element_size(), bias=bias is not None),
               "hdmoe_attn_fwd", out, lse, q, k, v, bias, B, Sq, Skv, H, D, Sb, _dt(q))
        ctx.save_for_backward(q, k, v, bias, out, lse)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, bias, out, lse = ctx.saved_tensors
        H = ctx.H
        g = _c(g)
        B, Sq, E = q.shape
        Skv = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        dbias = None
        Sb = 0
        direct = False
        if bias is not None:
            Sb = bias.shape[-1]
            if ctx.needs_input_grad[3]:
                direct = _direct(bias)
                dbias = bias.grad if direct else torch.zeros_like(bias)
        _timed("attn", dict(dir="bwd", B=B, Sq=Sq, Skv=Skv, H=H, D=E // H, esz=q.element_size(), bias=bias is not None),
               "hdmoe_attn_bwd", dq, dk, dv, dbias, delta, g, out, q, k, v, lse, bias, B, Sq, Skv, H, E // H, Sb, _dt(q))
        return dq, dk, dv, (None if direct else dbias), None


class _BicubicFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, S):
        table = _f32(table)
        H, S0, _ = table.shape
        out = torch.empty((H, S, S), dtype=torch.float32, device=table.device)
        call("hdmoe_bicubic_fwd", out, table, H, S0, S)
        ctx.dims = (H, S0, S)
        return out

    @staticmethod
    def backward(ctx, g):
        H, S0, S = ctx.dims
        dt = torch.zeros((H, S0, S0), dtype=torch.float32, device=g.device)
        call("hdmoe_bicubic_bwd", dt, _f32(g), H, S0, S)
        return dt, None


def bicubic_resize(table: Tensor, S: int) -> Tensor:
    """(H,S0,S0) -> (H,S,S): F.interpolate(mode='bicubic', align_corners=False) of the rel_pos_bias table
    (reference model_internals.py:388-397)."""
    return _BicubicFn.apply(table, int(S))


def attention(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor], num_heads: int) -> Tensor:
    """softmax(q k^T / sqrt(D) + bias) v per head; (B,S,E) tensors, head h = channels [h*D,(h+1)*D)."""
    return _AttnFn.apply(q, k, v, bias, int(num_heads))


# =====================================================================================================
# ragged tokens: the ViT expert bank (csrc/ragged.hip, hdmoe_attn_rag_* in csrc/attention.hip)
# =====================================================================================================
class RagLayout:
    """Routed rows of all ViT experts in one padded tensor (R, Sp, C): rows [seg[g], seg[g+1]) (device int32, from the
    dispatch plan) belong to expert g and hold ``lens[g]`` real tokens followed by padding."""

    def __init__(self, seg: Tensor, lens: Sequence[int], R: int):
        self.seg, self.lens, self.R = seg, [int(v) for v in lens], int(R)
        self.G, self.Sp = len(self.lens), max(int(v) for v in lens)


def _param_grads(params):
    """(buffers the kernels accumulate into, what to hand back to autograd) for small fp32 parameters."""
    direct = all(_direct(p) for p in params)
    bufs = [p.grad if direct else torch.zeros_like(p) for p in params]
    return bufs, ([None] * len(params) if direct else bufs)


class _RagPackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rag, *tensors):
        G = rag.G
        srcs, pos = [_c(t) for t in tensors[:G]], tensors[G:]
        C = srcs[0].shape[-1]
        dst = torch.empty((rag.R, rag.Sp, C), dtype=srcs[0].dtype, device=srcs[0].device)
        call("hdmoe_rag_pack", dst, srcs, [_f32(p) for p in pos], rag.seg, rag.lens, G, rag.R, rag.Sp, C, _dt(dst))
        ctx.rag, ctx.C = rag, C
        ctx.save_for_backward(*pos)
        return dst

    @staticmethod
    def backward(ctx, g):
        rag, C, pos = ctx.rag, ctx.C, ctx.saved_tensors
        g = _c(g)
        dsrcs = [torch.empty((rag.R, L, C), dtype=g.dtype, device=g.device) for L in rag.lens]
        bufs, ret = _param_grads(pos)
        call("hdmoe_rag_pack_bwd", dsrcs, bufs, g, rag.seg, rag.lens, rag.G, rag.R, rag.Sp, C, _dt(g))
        return (None, *dsrcs, *ret)


def rag_pack(srcs: Sequence[Tensor], pos: Sequence[Tensor], rag: RagLayout) -> Tensor:
    """Per-expert compact tokens (R, S_g, C) (+ that expert's pos_emb (1, S_g, C)) -> padded (R, Sp, C); row r takes expert g(r)'s."""
    return _RagPackFn.apply(rag, *srcs, *pos)


class _RagUnpackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, rag):
        tok = _c(tok)
        C = tok.shape[-1]
        dsts = [torch.empty((rag.R, L, C), dtype=tok.dtype, device=tok.device) for L in rag.lens]
        call("hdmoe_rag_unpack", dsts, tok, rag.seg, rag.lens, rag.G, rag.R, rag.Sp, C, _dt(tok))
        ctx.rag, ctx.C = rag, C
        return tuple(dsts)

    @staticmethod
    def backward(ctx, *gs):
        rag, C = ctx.rag, ctx.C
        ref = next(g for g in gs if g is not None)
        gs = [_c(g) if g is not None else torch.zeros((rag.R, L, C), dtype=ref.dtype, device=ref.device) for g, L in zip(gs, rag.lens)]
        d = torch.empty((rag.R, rag.Sp, C), dtype=ref.dtype, device=ref.device)
        call("hdmoe_rag_unpack_bwd", d, gs, rag.seg, rag.lens, rag.G, rag.R, rag.Sp, C, _dt(d))
        return d, None


def rag_unpack(tok: Tensor, rag: RagLayout):
    """Padded (R, Sp, C) -> one compact (R, S_g, C) tensor per expert (all rows: the caller selects each row's own expert later)."""
    return _RagUnpackFn.apply(tok, rag)


class _RagSelectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rag, *outs):
        outs = [_c(o) for o in outs]
        y = torch.empty_like(outs[0])
        nb = y.numel() // rag.R * y.element_size()
        call("hdmoe_rag_select", y, outs, rag.seg, rag.G, rag.R, nb)
        ctx.rag, ctx.nb = rag, nb
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        douts = [torch.empty_like(g) for _ in range(ctx.rag.G)]
        call("hdmoe_rag_select_bwd", douts, g, ctx.rag.seg, ctx.rag.G, ctx.rag.R, ctx.nb)
        return (None, *douts)


def rag_select(outs: Sequence[Tensor], rag: RagLayout) -> Tensor:
    """y[r] = outs[g(r)][r] over same-shaped per-expert tensors (R, ...)."""
    return _RagSelectFn.apply(rag, *outs)


class _GNRagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rag, groups, act, eps, *params):
        x = _c(x)
        G = rag.G
        C = x.shape[-1]
        mean = torch.empty(rag.R * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        call("hdmoe_gn_rag_fwd", y, mean, rstd, x, list(params[:G]), list(params[G:]), rag.seg, rag.lens, G, rag.R, rag.Sp, C, groups, act, eps, _dt(x))
        ctx.save_for_backward(x, mean, rstd, *params)
        ctx.meta = (rag, groups, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd, *params = ctx.saved_tensors
        rag, groups, act = ctx.meta
        G = rag.G
        g = _c(g)
        dx = torch.empty_like(x)
        bufs, ret = _param_grads(params)
        call("hdmoe_gn_rag_bwd", dx, bufs[:G], bufs[G:], g, x, list(params[:G]), list(params[G:]), mean, rstd, rag.seg, rag.lens, G, rag.R, rag.Sp,
             x.shape[-1], groups, act, _dt(x))
        return (dx, None, None, None, None, *ret)


def gn_rag(x: Tensor, gammas: Sequence[Tensor], betas: Sequence[Tensor], rag: RagLayout, groups: int, act: int = 0, eps: float = 1e-5) -> Tensor:
    """nn.GroupNorm (+ activation) over each row's REAL tokens with the row's expert's affine; padding -> 0."""
    return _GNRagFn.apply(x, rag, int(groups), int(act), float(eps), *gammas, *betas)


class _LNRagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rag, eps, *params):
        x = _c(x)
        G = rag.G
        C = x.shape[-1]
        mean = torch.empty(rag.R * rag.Sp, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        call("hdmoe_ln_rag_fwd", y, mean, rstd, x, list(params[:G]), list(params[G:]), rag.seg, G, rag.R, rag.Sp, C, eps, _dt(x))
        ctx.save_for_backward(x, mean, rstd, *params)
        ctx.rag = rag
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd, *params = ctx.saved_tensors
        rag = ctx.rag
        G = rag.G
        g = _c(g)
        dx = torch.empty_like(x)
        bufs, ret = _param_grads(params)
        call("hdmoe_ln_rag_bwd", dx, bufs[:G], bufs[G:], g, x, list(params[:G]), mean, rstd, rag.seg, G, rag.R, rag.Sp, x.shape[-1], _dt(x))
        return (dx, None, None, *ret)


def ln_rag(x: Tensor, gammas: Sequence[Tensor], betas: Sequence[Tensor], rag: RagLayout, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm per token with the row's expert's affine."""
    return _LNRagFn.apply(x, rag, float(eps), *gammas, *betas)


class _AttnRagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, rag, H, *biases):
        q = _c(q); k = _c(k); v = _c(v)
        R, Sp, E = q.shape
        sb = [int(b.shape[-1]) for b in biases]
        out = torch.empty_like(q)
        lse = torch.empty((R, H, Sp), dtype=torch.float32, device=q.device)
        call("hdmoe_attn_rag_fwd", out, lse, q, k, v, list(biases), rag.seg, rag.lens, sb, rag.G, R, Sp, H, E // H, _dt(q))
        ctx.save_for_backward(q, k, v, out, lse, *biases)
        ctx.meta = (rag, H, sb)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse, *biases = ctx.saved_tensors
        rag, H, sb = ctx.meta
        g = _c(g)
        R, Sp, E = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        bufs, ret = (None, [None] * rag.G)
        if any(ctx.needs_input_grad[5:]):
            bufs, ret = _param_grads(biases)
        call("hdmoe_attn_rag_bwd", dq, dk, dv, bufs, delta, g, out, q, k, v, lse, list(biases), rag.seg, rag.lens, sb, rag.G, R, Sp, H, E // H, _dt(q))
        return (dq, dk, dv, None, None, *ret)


def attention_rag(q: Tensor, k: Tensor, v: Tensor, biases: Sequence[Tensor], rag: RagLayout, num_heads: int) -> Tensor:
    """Self-attention over each row's real tokens with the row's expert's rel_pos_bias table (H, S_g, S_g)."""
    return _AttnRagFn.apply(q, k, v, rag, int(num_heads), *biases)


# =====================================================================================================
# router head + dispatch
# =====================================================================================================
class _RouterHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, noise, mask, k):
        logits = _f32(logits)
        B, E = logits.shape
        noise = None if noise is None else _f32(noise)
        mask = None if mask is None else _c(mask.to(torch.float32))
        sparse = torch.empty_like(logits)
        probs = torch.empty_like(logits)
        xout = torch.empty_like(logits)
        idx = torch.empty((B, k), dtype=torch.int32, device=logits.device)
        call("hdmoe_router_head_fwd", sparse, probs, xout, idx, logits, noise, mask, B, E, k)
        ctx.save_for_backward(sparse, probs, idx, mask)
        ctx.k = k
        ctx.mark_non_differentiable(idx)
        return sparse, probs, xout, idx

    @staticmethod
    def backward(ctx, dsparse, dprobs, dxout, _didx):
        sparse, probs, idx, mask = ctx.saved_tensors
        B, E = sparse.shape
        dl = torch.empty_like(sparse)
        call("hdmoe_router_head_bwd", dl, _c(dsparse), _c(dprobs), _c(dxout), sparse, probs, idx, mask, B, E, ctx.k)
        return dl, None, None, None


def router_head(logits: Tensor, noise: Optional[Tensor], mask: Optional[Tensor], k: int):
    """masked_fill -> softmax -> top-k -> softmax(top-k) -> sparse scatter (model_components.py:158-168).
    Returns (sparse_weights, gate_probs, masked_logits, topk_idx int32)."""
    return _RouterHeadFn.apply(logits, noise, mask, int(k))


class DispatchPlan:
    """Device-side expert-contiguous permutation of the routed (sample, expert) pairs."""

    def __init__(self, sparse: Tensor, kcap: int):
        sparse = _f32(sparse.detach())
        B, E = sparse.shape
        self.B, self.E, self.kcap = B, E, int(kcap)
        self.R = B * self.kcap
        dev = sparse.device
        self.perm = torch.empty(self.R, dtype=torch.int32, device=dev)
        self.row_expert = torch.empty(self.R, dtype=torch.int32, device=dev)
        self.row_w = torch.empty(self.R, dtype=torch.float32, device=dev)
        self.inv = torch.empty(self.R, dtype=torch.int32, device=dev)
        self.seg = torch.empty(E + 1, dtype=torch.int32, device=dev)
        call("hdmoe_dispatch_plan", self.perm, self.row_expert, self.row_w, self.inv, self.seg, sparse, B, E, self.kcap)


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan):
        x = _c(x)
        L = x.numel() // x.shape[0]
        out = torch.empty((plan.R, *x.shape[1:]), dtype=x.dtype, device=x.device)
        call("hdmoe_gather_rows", out, x, plan.perm, plan.R, L, _dt(x))
        ctx.plan = plan
        ctx.xshape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g = _c(g)
        L = g.numel() // plan.R
        dx = torch.empty(ctx.xshape, dtype=g.dtype, device=g.device)
        call("hdmoe_combine_rows_fwd", dx, g, plan.inv, None, plan.B, plan.kcap, L, _dt(g))
        return dx, None


def gather_rows(x: Tensor, plan: DispatchPlan) -> Tensor:
    """x[perm] : (B, ...) -> (R, ...) in expert-contiguous order (zeros in unused rows)."""
    return _GatherFn.apply(x, plan)


class _CombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys, sparse, plan):
        ctx.pool_ok = not sparse.is_leaf
        ys = _c(ys)
        L = ys.numel() // plan.R
        out = torch.empty((plan.B, *ys.shape[1:]), dtype=ys.dtype, device=ys.device)
        call("hdmoe_combine_rows_fwd", out, ys, plan.inv, plan.row_w, plan.B, plan.kcap, L, _dt(ys))
        ctx.save_for_backward(ys)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        (ys,) = ctx.saved_tensors
        plan = ctx.plan
        g = _c(g)
        L = ys.numel() // plan.R
        dys = torch.empty_like(ys)
        dsp = _zeros((plan.B, plan.E), torch.float32, g.device, ctx.pool_ok) if ctx.needs_input_grad[1] else None
        call("hdmoe_combine_rows_bwd", dys, dsp, g, ys, plan.perm, plan.row_expert, plan.row_w, plan.R, plan.E, L, _dt(ys))
        return dys, dsp, None


def combine_rows(ys: Tensor, sparse: Tensor, plan: DispatchPlan) -> Tensor:
    """out[b] = sum over the rows r routed from sample b of sparse[b, e(r)] * ys[r]  (output[mask] += out*w)."""
    return _CombineFn.apply(ys, sparse, plan)


# =====================================================================================================
# small vector-path helpers (no autograd needed: inputs are data, not parameters)
# =====================================================================================================
def fourier(x: Tensor, freqs: Tensor, phases: Tensor) -> Tensor:
    """MP_Fourier.forward (model_internals.py:158-175); x must be 1-D."""
    if x.ndim != 1:
        raise RuntimeError("MP_Fourier expects a 1-D input")
    x = _f32(x.detach().to(torch.float32))
    out = torch.empty((x.shape[0], freqs.shape[0]), dtype=torch.float32, device=x.device)
    call("hdmoe_fourier", out, x, _f32(freqs), _f32(phases), x.shape[0], freqs.shape[0])
    return out


def edm_coeffs(sigma: Tensor, sigma_data: float, B: int) -> Tensor:
    """(4, B) float32: c_skip, c_out, c_in, c_noise (model_config2.py:431-438)."""
    s = _f32(sigma.detach().to(torch.float32).reshape(-1))
    coef = torch.empty((4, B), dtype=torch.float32, device=s.device)
    call("hdmoe_edm_coeffs", coef, s, s.numel(), float(sigma_data), B)
    return coef


def sigmoid_scaling(c_noise: Tensor, transition_point: float, softness: float):
    """(s_vit (B,), s_unet (B,), scaling_factors (B,2)) (model_config2.py:244-249)."""
    c = _f32(c_noise)
    B = c.shape[0]
    sv = torch.empty(B, dtype=torch.float32, device=c.device)
    su = torch.empty_like(sv)
    pair = torch.empty((B, 2), dtype=torch.float32, device=c.device)
    call("hdmoe_sigmoid_scaling", sv, su, pair, c, float(transition_point), float(softness), B)
    return sv, su, pair


# =====================================================================================================
# EDM_LOSS (row N2: the step right after the path) -- fused, sync-free
# =====================================================================================================
class _EDMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, denoised, target, log_var, pU, pV, rU, rV, cfg):
        unet_bal, vit_bal, z_bal = cfg
        d = _f32(denoised); t = _f32(target.detach())
        B = d.shape[0]
        L = d.numel() // B
        E = pU.shape[1]
        lv = None if log_var is None else _f32(log_var).reshape(-1)
        pU, pV, rU, rV = _f32(pU), _f32(pV), _f32(rU), _f32(rV)
        out = torch.empty(5, dtype=torch.float32, device=d.device)
        aux = torch.empty(2 * E + 3, dtype=torch.float32, device=d.device)
        sse = torch.empty(B, dtype=torch.float32, device=d.device)
        call("hdmoe_edm_loss_fwd", out, aux, sse, d, t, lv, pU, pV, rU, rV, B, L, E, unet_bal, vit_bal, z_bal)
        ctx.save_for_backward(d, t, lv, rU, rV, aux, sse)
        ctx.meta = (B, L, E, cfg, None if log_var is None else log_var.shape)
        stats = out.detach()
        ctx.mark_non_differentiable(stats)
        return out[0], stats

    @staticmethod
    def backward(ctx, g, _gstats):
        d, t, lv, rU, rV, aux, sse = ctx.saved_tensors
        B, L, E, (unet_bal, vit_bal, z_bal), lv_shape = ctx.meta
        g = _f32(g.reshape(1))
        dD = torch.empty_like(d)
        dlv = None if lv is None else torch.empty_like(lv)
        dpU, dpV, drU, drV = (torch.empty_like(rU) for _ in range(4))
        call("hdmoe_edm_loss_bwd", dD, dlv, dpU, dpV, drU, drV, g, aux, sse, d, t, lv, rU, rV, B, L, E, unet_bal, vit_bal, z_bal)
        return dD, None, (None if dlv is None else dlv.reshape(lv_shape)), dpU, dpV, drU, drV, None


def edm_loss(denoised: Tensor, target: Tensor, log_var: Optional[Tensor], pU: Tensor, pV: Tensor, rU: Tensor, rV: Tensor,
             unet_bal: float, vit_bal: float, z_bal: float):
    """Returns (loss 0-dim with grad, stats (5,) detached = [loss, denoising, balance, z_loss, pure_loss])."""
    return _EDMLossFn.apply(denoised, target, log_var, pU, pV, rU, rV, (float(unet_bal), float(vit_bal), float(z_bal)))
