"""Magnitude-preserving primitives -- drop-in for the reference's ``models/model_internals.py``.

Same public names and call conventions (normalize, mp_silu, mp_sum, mp_cat, resample, MP_Fourier, MP_Conv,
MP_Attention); tensors at this public boundary are logical NCHW / (B,S,E) exactly as in the reference, on the
GPU.  Internally everything runs channel-last through ``hdmoe_hip.ops`` (HIP kernels); methods named ``_fwd``
take / return the internal layout so that composite modules never round-trip through NCHW.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from hdmoe_hip import ops

Tensor = torch.Tensor


def _same_dense_layout(a: Tensor, b: Tensor) -> bool:
    return a.shape == b.shape and a.stride() == b.stride() and \
        (a.is_contiguous() or (a.ndim == 4 and a.is_contiguous(memory_format=torch.channels_last)))


def _like_storage(out_flat: Tensor, ref: Tensor) -> Tensor:
    """Reinterpret a freshly produced dense buffer with the same (shape, strides) as ``ref``."""
    return out_flat.as_strided(ref.shape, ref.stride())


def _storage_view(t: Tensor) -> Tensor:
    """1-D view over the dense storage of a contiguous or channels_last tensor (element order irrelevant)."""
    return t.as_strided((t.numel(),), (1,))


def normalize(x: Tensor, dim: Optional[list] = None, eps: float = 1e-4) -> Tensor:
    """x / (eps + ||x||_2(dim) * sqrt(n_norm / n_x))  (reference model_internals.py:8-30).

    The kernel's eps is 1e-4 (every reference call site); another eps > 0 uses x / (eps + s |x|) == (x k) / (1e-4 + s |x k|) with
    k = 1e-4 / eps.  Reductions other than all-but-first / the channel dim of a 4-D tensor move the reduced dims to the end first."""
    if not eps > 0:
        raise NotImplementedError("normalize: eps must be positive")
    if eps != 1e-4:
        x = ops.axpby(x.contiguous(), None, 1e-4 / float(eps), 0.0)
    if dim is None or sorted(d % x.ndim for d in dim) == list(range(1, x.ndim)):
        flat = x.contiguous().reshape(x.shape[0], -1)
        return ops.pixel_norm(flat).reshape(x.shape)
    if x.ndim == 4 and [d % 4 for d in dim] == [1]:
        return ops.from_nhwc(ops.pixel_norm(ops.to_nhwc(x)))
    red = sorted({d % x.ndim for d in dim})
    keep = [d for d in range(x.ndim) if d not in red]
    perm = keep + red
    xp = x.permute(perm).contiguous()
    R = 1
    for d in red:
        R *= x.shape[d]
    out = ops.pixel_norm(xp.reshape(-1, R)).reshape(xp.shape)
    inv = [perm.index(d) for d in range(x.ndim)]
    return out.permute(inv)


def mp_silu(x: Tensor) -> Tensor:
    """silu(x) / 0.596 (reference model_internals.py:33-47)."""
    if x.ndim == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
        return _like_storage(ops.mp_silu(_storage_view(x)), x)
    return ops.mp_silu(x)


def mp_sum(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """lerp(a, b, t) / sqrt((1-t)^2 + t^2) (reference model_internals.py:50-66)."""
    if a.shape != b.shape:
        a, b = torch.broadcast_tensors(a, b)
    if _same_dense_layout(a, b) and not a.is_contiguous():
        return _like_storage(ops.mp_sum(_storage_view(a), _storage_view(b), t), a)
    return ops.mp_sum(a, b, t)


def mp_cat(a: Tensor, b: Tensor, dim: int = 1, t: float = 0.5) -> Tensor:
    """Magnitude-preserving concat (reference model_internals.py:69-92); channel dim of NCHW / last dim otherwise."""
    if a.ndim == 4 and dim % 4 == 1:
        return ops.from_nhwc(ops.mp_cat(ops.to_nhwc(a), ops.to_nhwc(b), t))
    d = dim % a.ndim
    if d == a.ndim - 1:
        return ops.mp_cat(a, b, t)
    # any other dim: concatenate along the last dim of the transposed tensors (the kernel's layout), transpose back
    return ops.mp_cat(a.transpose(d, -1).contiguous(), b.transpose(d, -1).contiguous(), t).transpose(d, -1)


def resample(x: Tensor, f=(1, 1), mode: Optional[str] = "keep") -> Tensor:
    """Separable-filter resampling (reference model_internals.py:95-127): f = [1, 1] is a 2x2 mean / nearest x2; any other even-length
    filter (up to 8 taps) runs on the generic FIR kernels."""
    if mode == "keep":
        return x
    if torch.is_tensor(f):
        f = f.detach().cpu().tolist()
    if mode not in ("down", "up"):
        raise ValueError(f"Invalid mode: {mode}")
    return ops.from_nhwc(ops.resample(ops.to_nhwc(x), mode, f=list(f)))


class MP_Fourier(nn.Module):
    """sqrt(2) * cos(x (x) freqs + phases), fp32 (reference model_internals.py:130-175)."""

    def __init__(self, num_channels: int, bandwidth: float = 1):
        super().__init__()
        self.register_buffer("freqs", 2 * torch.pi * torch.randn(num_channels) * bandwidth)
        self.register_buffer("phases", 2 * torch.pi * torch.rand(num_channels))

    def forward(self, x: Tensor) -> Tensor:
        y = ops.fourier(x, self.freqs, self.phases)
        return y if x.dtype == torch.float32 else ops.cast(y, x.dtype)


class Pos_encoding(nn.Module):
    """Sinusoidal timestep features -> Linear -> SiLU -> Linear (reference model_internals.py:178-206; unused by the reference's models,
    kept for its callers and tests).  Same parameters / buffer (`mlp.0`, `mlp.2`, `freq`); the features come from the Fourier kernel
    ([cos(t f), sin(t f)] = cos(t [f, f] + [0, -pi/2])), the two plain linears from the conv kernel with un-normalised weights."""

    def __init__(self, emb_dim: Optional[int] = 512, freq_emb_dim: Optional[int] = 256, max_period: Optional[int] = 10000):
        super().__init__()
        assert freq_emb_dim % 2 == 0
        self.half_dim = freq_emb_dim // 2
        self.max_period = max_period
        self.mlp = nn.Sequential(nn.Linear(in_features=freq_emb_dim, out_features=emb_dim), nn.SiLU(),
                                 nn.Linear(in_features=emb_dim, out_features=emb_dim))
        expo = -1 * np.log(self.max_period) * torch.arange(start=0, end=self.half_dim, dtype=torch.float32) / self.half_dim
        self.register_buffer("freq", torch.exp(expo))

    def forward(self, time_vec: Tensor) -> Tensor:
        if time_vec.ndim > 1:
            time_vec = time_vec.flatten()
        f2 = torch.cat([self.freq, self.freq])
        ph = torch.cat([torch.zeros_like(self.freq), torch.full_like(self.freq, -0.5 * math.pi)])
        emb = ops.axpby(ops.fourier(time_vec, f2, ph), None, 1.0 / math.sqrt(2.0), 0.0)        # the kernel returns sqrt(2) cos(.)
        h = ops.bias_add(ops.mp_conv(emb, self.mlp[0].weight, 1.0, normalize=False), self.mlp[0].bias)
        h = ops.axpby(ops.mp_silu(h), None, 0.596, 0.0)                                         # plain SiLU = 0.596 * mp_silu
        return ops.bias_add(ops.mp_conv(h, self.mlp[2].weight, 1.0, normalize=False), self.mlp[2].bias)


class MP_Conv(nn.Module):
    """Weight-normalised conv / linear without bias (reference model_internals.py:209-275).

    Parameter name ``weights`` and shape ``(out, in, *kernel)`` as in the reference; fp32 master weights.
    In training mode the forward re-normalises the stored weights in place (reference :254-256)."""

    def __init__(self, in_channels: int, out_channels: int, kernel: tuple, stride: int = 1):
        super().__init__()
        self.out_channels = out_channels
        self.weights = nn.Parameter(torch.randn(out_channels, in_channels, *kernel))
        assert self.weights.numel() != 0
        self.kernel = kernel
        self.stride = stride

    def _fwd(self, x: Tensor, gain=1.0, **kw) -> Tensor:
        """channel-last in / out; extra kwargs (res/alpha/beta/ones) are forwarded to ops.mp_conv."""
        if self.stride != 1:
            if kw or x.ndim != 4 or torch.is_tensor(gain):
                raise NotImplementedError("MP_Conv(stride > 1): plain 4-D forward only (no reference model uses a strided MP_Conv)")
            return ops.mp_conv_strided(x, self.weights, gain, self.stride, training=self.training)
        return ops.mp_conv(x, self.weights, gain, training=self.training, **kw)

    def forward(self, x: Tensor, gain: float = 1.0) -> Tensor:
        if x.ndim == 2:
            if self.weights.ndim != 2:
                raise RuntimeError("MP_Conv: 2-D input needs a linear (kernel=()) layer")
            return self._fwd(x, gain)
        assert x.ndim == 4
        if self.weights.ndim != 4:
            raise RuntimeError("MP_Conv: 4-D input needs a conv kernel")
        return ops.from_nhwc(self._fwd(ops.to_nhwc(x), gain))


class MP_Attention(nn.Module):
    """Magnitude-preserving multi-head attention, self or cross, optionally time-conditioned
    (reference model_internals.py:279-409).  The S x S score map is never materialised."""

    def __init__(self, num_heads: int, emb_dim: int, seq_ln: int, time_dim: Optional[int] = 0,
                 context_dim: Optional[int] = None, attn_balance: Optional[float] = 0.5,
                 is_cross_attn: Optional[bool] = False):
        super().__init__()
        self.num_heads = num_heads
        self.emb_dim = emb_dim
        self.head_dim = emb_dim // num_heads
        self.time_emb = time_dim
        assert emb_dim % num_heads == 0
        if context_dim is None:
            context_dim = emb_dim
        self.is_cross = is_cross_attn
        self.attn_balance = attn_balance
        self.time_dependent = True if time_dim > 0 else False
        self.rel_pos_bias = nn.Parameter(torch.zeros(self.num_heads, seq_ln, seq_ln)) if not is_cross_attn else None
        self.q_proj = MP_Conv(emb_dim, emb_dim, kernel=(1, 1))
        self.k_proj = MP_Conv(context_dim, emb_dim, kernel=(1, 1))
        self.v_proj = MP_Conv(context_dim, emb_dim, kernel=(1, 1))
        self.q_time = MP_Conv(time_dim, emb_dim, kernel=(1, 1)) if self.time_dependent else None
        self.k_time = MP_Conv(time_dim, emb_dim, kernel=(1, 1)) if self.time_dependent and not is_cross_attn else None
        self.v_time = MP_Conv(time_dim, emb_dim, kernel=(1, 1)) if self.time_dependent and not is_cross_attn else None
        self.out_proj = MP_Conv(emb_dim, emb_dim, kernel=(1, 1))

    def forward(self, query: Tensor, gain_s: float, gain_t: float, context: Optional[Tensor] = None,
                time_embedding: Optional[Tensor] = None) -> Tensor:
        batch_size, seq_len, emb_dim = query.shape
        assert emb_dim == self.emb_dim
        dt = query.dtype
        ctx = query if context is None else ops.cast(context, dt)
        q = self.q_proj._fwd(query, gain_s)
        k = self.k_proj._fwd(ctx, gain_s)
        v = self.v_proj._fwd(ctx, gain_s)
        if self.time_dependent and time_embedding is not None:
            te = ops.cast(time_embedding.reshape(batch_size, -1), torch.float32)
            q = ops.seq_bcast_add(q, self.q_time._fwd(te, gain_t))
            if not self.is_cross:
                k = ops.seq_bcast_add(k, self.k_time._fwd(te, gain_t))
                v = ops.seq_bcast_add(v, self.v_time._fwd(te, gain_t))
        bias = None
        if not self.is_cross:
            bias = self.rel_pos_bias                      # the kernel reads the [:S, :S] corner in place
            if seq_len > bias.shape[1]:                   # longer than trained: bicubic resize of the table (:388-397)
                bias = ops.bicubic_resize(bias, seq_len)
        o = ops.attention(q, k, v, bias, self.num_heads)
        t = self.attn_balance
        n = math.sqrt((1.0 - t) ** 2 + t ** 2)
        # out_proj with the mp_sum(res, out, attn_balance) residual fused into the conv epilogue
        return self.out_proj._fwd(o, gain_s, res=query, alpha=t / n, beta=(1.0 - t) / n)
