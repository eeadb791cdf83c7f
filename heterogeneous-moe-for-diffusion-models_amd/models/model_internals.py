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
    Supported reductions: all-but-first (default) and the channel dim of a 4-D tensor."""
    if eps != 1e-4:
        raise NotImplementedError("normalize: eps is fixed to 1e-4 in the HIP kernels")
    if dim is None or sorted(d % x.ndim for d in dim) == list(range(1, x.ndim)):
        flat = x.contiguous().reshape(x.shape[0], -1)
        return ops.pixel_norm(flat).reshape(x.shape)
    if x.ndim == 4 and [d % 4 for d in dim] == [1]:
        return ops.from_nhwc(ops.pixel_norm(ops.to_nhwc(x)))
    raise NotImplementedError(f"normalize: unsupported dim={dim} for a {x.ndim}-D tensor")


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
    if dim % a.ndim == a.ndim - 1:
        return ops.mp_cat(a, b, t)
    raise NotImplementedError("mp_cat: only the channel dim of NCHW tensors or the last dim is supported")


def resample(x: Tensor, f=(1, 1), mode: Optional[str] = "keep") -> Tensor:
    """Box-filter resampling (reference model_internals.py:95-127) for the default f=[1,1]."""
    if mode == "keep":
        return x
    if list(f) != [1, 1]:
        raise NotImplementedError("resample: only the default f=[1,1] filter is implemented")
    if mode not in ("down", "up"):
        raise ValueError(f"Invalid mode: {mode}")
    return ops.from_nhwc(ops.resample(ops.to_nhwc(x), mode))


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
    """Present for import compatibility only: unused by every model in the reference
    (model_internals.py:177 "not used right now") and therefore outside the accelerated path."""

    def __init__(self, emb_dim: Optional[int] = 512, freq_emb_dim: Optional[int] = 256, max_period: Optional[int] = 10000):
        super().__init__()
        raise NotImplementedError("Pos_encoding is not on the HDMOEM hot path and is not provided by this build")


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
            raise NotImplementedError("MP_Conv with stride > 1 is unused by the reference models and not implemented")
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
