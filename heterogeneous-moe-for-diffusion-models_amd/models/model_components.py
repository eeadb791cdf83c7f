"""Routers and experts -- drop-in for the reference's ``models/model_components.py``.

Class names, constructor / forward signatures, attribute names and state_dict keys follow the reference
(Scaling_router:7, Router:68, Unet_block:171, Unet_expert:255, Vit_block:435, Vit_expert:564).  Public ``forward``
methods speak the reference's logical layouts (NCHW images, (B,S,C) tokens); ``_fwd`` methods and the
``*_bank_forward`` functions work channel-last and can run a whole bank of heterogeneous experts in one grouped
launch per layer (``seg`` = device-side row offsets from the dispatch plan).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

import models.model_internals as m
from hdmoe_hip import ops

Tensor = torch.Tensor


class Scaling_router(nn.Module):
    """2-way soft gate whose rows sum to 2 (reference model_components.py:7-66)."""

    def __init__(self, emb_dim: Optional[int] = 3, num_experts: Optional[int] = 2, dropout: Optional[float] = 0.2):
        super().__init__()
        self.soft_route = nn.Sequential(
            m.MP_Conv(in_channels=emb_dim, out_channels=emb_dim * 2, kernel=()),
            nn.GroupNorm(1, emb_dim * 2),
            nn.ReLU(),
            m.MP_Conv(in_channels=emb_dim * 2, out_channels=emb_dim * 4, kernel=()),
            nn.GroupNorm(1, emb_dim * 4),
            nn.ReLU(),
            nn.Dropout(dropout),
        )
        self.linear = m.MP_Conv(in_channels=emb_dim * 4, out_channels=num_experts, kernel=())

    def forward(self, x: Tensor, zeta: Optional[float] = 1e-2) -> Tensor:
        if x.ndim == 3:
            x = x.squeeze(1)
        sr = self.soft_route
        x = ops.group_norm(sr[0]._fwd(x), sr[1].weight, sr[1].bias, 1, ops.ACT_RELU, sr[1].eps)
        x = ops.group_norm(sr[3]._fwd(x), sr[4].weight, sr[4].bias, 1, ops.ACT_RELU, sr[4].eps)
        x = ops.dropout(x, sr[6].p, self.training)
        x = self.linear._fwd(x)
        if self.training:
            x = ops.axpby(x, ops.randn_like(x, zeta), 1.0, 1.0)
        return ops.softmax_rows(x, 2.0)


class Router(nn.Module):
    """Sparse noisy top-k gate (reference model_components.py:68-168).
    Returns (sparse_gate_weights, gate_probs, masked_logits), each (B, num_experts)."""

    def __init__(self, in_channels: Optional[int] = 3, time_dim: Optional[int] = 256, top_k: Optional[int] = 1,
                 num_experts: Optional[int] = 5, dropout: Optional[float] = 0.2):
        super().__init__()
        self.hard_route = nn.Sequential(
            m.MP_Conv(in_channels=in_channels, out_channels=in_channels * 2, kernel=(3, 3)),
            nn.GroupNorm(1, in_channels * 2),
            nn.ReLU(),
            m.MP_Conv(in_channels=in_channels * 2, out_channels=in_channels * 4, kernel=(3, 3)),
            nn.GroupNorm(1, in_channels * 4),
            nn.ReLU(),
            m.MP_Conv(in_channels=in_channels * 4, out_channels=in_channels * 4, kernel=(3, 3)),
            nn.GroupNorm(1, in_channels * 4),
            nn.ReLU(),
            nn.AdaptiveAvgPool2d((1, 1)),
            nn.Dropout(dropout),
        )
        self.out_router = in_channels * 4
        self.time_linear = m.MP_Conv(in_channels=time_dim, out_channels=self.out_router * 2, kernel=())
        self.linear = m.MP_Conv(in_channels=in_channels * 4, out_channels=num_experts, kernel=())
        self.k = top_k

    def _fwd(self, x: Tensor, time_emb: Tensor, mask: Optional[Tensor], zeta):
        """x channel-last (B,H,W,C); returns (sparse, probs, logits, topk_idx)."""
        hr = self.hard_route
        # The trunk stays fp32 in HBM in every mode (its logits feed topk: indices must equal the fp32 reference).  In bf16 compute
        # mode its convs run on the bf16 matrix pipe as split-bf16 (hi + lo, ~5e-6 relative: csrc/conv6s.hip); in fp32 mode they
        # keep the exact fp32-input MFMA -- the trunk's GroupNorm + ReLU turns even 1e-5 perturbations into mask flips that show
        # in the trunk gradients at the fp32 tests' 3e-4 tolerance.
        import hdmoe_hip
        split = ops.ROUTER_SPLIT and hdmoe_hip.compute_dtype() == torch.bfloat16
        pooled = None
        convs = [hr[0].weights, hr[3].weights, hr[6].weights]
        if split and ops.trunk_ok(x, convs):
            # GroupNorm + ReLU folded into the neighbouring convs; None until the weight bank has prepared the layers (first step)
            pooled = ops.router_trunk(x, convs, [hr[1], hr[4], hr[7]])
        if pooled is None:
            for ci, gi in ((0, 1), (3, 4), (6, 7)):
                x = ops.group_norm(hr[ci]._fwd(x, split=split), hr[gi].weight, hr[gi].bias, 1, ops.ACT_RELU, hr[gi].eps)
            pooled = ops.seq_mean(x)                                    # AdaptiveAvgPool2d(1) -> fp32 (B, 4C)
        x = pooled
        x = ops.dropout(x, hr[10].p, self.training)
        if time_emb.ndim == 3:
            time_emb = time_emb.squeeze(1)
        cond = self.time_linear._fwd(ops.mp_silu(ops.cast(time_emb, torch.float32)))
        x = ops.adaln(x, cond)
        logits = self.linear._fwd(x)
        noise = ops.randn_like(logits, zeta) if self.training else None
        return ops.router_head(logits, noise, mask, self.k)

    def forward(self, x: Tensor, time_emb: Tensor, mask: Optional[Tensor] = None, zeta: Optional[float] = 1e-2):
        sparse, probs, logits, _ = self._fwd(ops.to_nhwc(x), time_emb, mask, zeta)
        return sparse, probs, logits


# ------------------------------------------------------------------------------------------------------------
# U-Net expert(s).  Every function takes a LIST of structurally identical modules (one per expert, possibly with
# different kernel sizes) and runs them as one grouped launch per layer.
# ------------------------------------------------------------------------------------------------------------
def _w(mods: Sequence[nn.Module], path: str) -> List[Tensor]:
    out = []
    for mod in mods:
        for p in path.split("."):
            mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
        out.append(mod)
    return out


def unet_block_bank_forward(blocks: Sequence["Unet_block"], x: Tensor, embedding: Tensor, seg: Optional[Tensor], film: Optional[Tensor] = None,
                            silu_x: Optional[Tensor] = None) -> Tensor:
    """Unet_block.forward (reference model_components.py:232-253) over a bank of same-shaped blocks.
    ``film``: the block's 1 + emb_layer(embedding) * gain when the caller computed it for all blocks at once (ops.multi_linear)."""
    b0 = blocks[0]
    tr = b0.training

    def conv(name, inp, gain=1.0, **kw):
        return ops.mp_conv(inp, [getattr(b, name).weights for b in blocks], gain, seg=seg, training=tr, **kw)

    emb = film if film is not None else ops.affine(conv("emb_layer", embedding, b0.emb_gain), 1.0, 1.0)   # 1 + emb_layer(e) * gain
    x = ops.resample(x, b0.resample)
    t = b0.residual_balance
    n = ((1.0 - t) ** 2 + t ** 2) ** 0.5
    # The residual x enters conv_res2's epilogue as beta * x.  Its gradient beta * dy would be a scaling pass per block: instead
    # beta is applied where the gradient is consumed anyway (the backward of the op that produced x), or -- when a conv_skip sits
    # in between -- folded into that conv's weights, and conv_res2 hands dy back as it is (res_grad_raw).
    beta = (1.0 - t) / n
    raw = False
    if b0.type == "enc":
        if b0.conv_skip is not None:
            x = conv("conv_skip", x)
        x, h = ops.pixel_norm_silu(x, gx_scale=beta)
        raw = True
    elif silu_x is not None:
        h = silu_x                                             # the caller fused mp_silu into the producer of x (ops.mp_cat_silu)
    else:
        skip_next = b0.conv_skip is not None
        x, h = ops.silu_branch(x, gx_scale=1.0 if skip_next else beta)     # main branch + skip / residual: one fused backward pass
        raw = not skip_next
    w1, w2 = [b.conv_res1.weights for b in blocks], [b.conv_res2.weights for b in blocks]
    if b0.type == "dec" and b0.conv_skip is not None:
        x = conv("conv_skip", x, alpha=beta)                   # beta folded into the skip projection's weight image
        rbeta, raw = 1.0, False
    else:
        rbeta = beta
    # bf16 bank path: conv_res1 -> FiLM * emb -> mp_silu -> F.dropout -> conv_res2 -> mp_sum as ONE launch, the activation tile kept in LDS (csrc/blk6.hip)
    yb = ops.unet_block_fused(h, x, w1, w2, b0.conv_gain1, b0.conv_gain2, emb, b0.dropout, tr, seg, alpha=t / n, beta=rbeta, res_grad_raw=raw)
    if yb is not None:
        return yb
    # conv_res1 -> FiLM * emb -> mp_silu -> F.dropout: the three elementwise steps as a second output of the conv's epilogue (eval), or one pass
    y = ops.mp_conv_film(h, w1, b0.conv_gain1, emb, b0.dropout, tr, seg=seg)
    # conv_res2 with mp_sum(x, main, residual_balance) fused into its epilogue
    return conv("conv_res2", y, b0.conv_gain2, res=x, alpha=t / n, beta=rbeta, res_grad_raw=raw)


class Unet_block(nn.Module):
    """EDM2-style residual block with a per-expert kernel size (reference model_components.py:171-253)."""

    def __init__(self, in_channels: int, out_channels: int, kernel: tuple, emb_size: int, resample: Optional[str] = "keep",
                 Type: Optional[str] = "enc", residual_balance: Optional[float] = 0.5, Dropout: Optional[float] = 0.2,
                 emb_gain: Optional[float] = 1.0, conv_gain: Optional[float] = 1.0):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.emb_size = emb_size
        self.residual_balance = residual_balance
        self.type = Type
        self.resample = resample
        self.kernel = kernel
        self.dropout = Dropout
        self.emb_gain = emb_gain
        self.conv_gain1 = conv_gain
        self.conv_gain2 = conv_gain
        self.conv_skip = m.MP_Conv(in_channels=in_channels, out_channels=out_channels, kernel=(1, 1)) \
            if in_channels != out_channels else None
        self.emb_layer = m.MP_Conv(in_channels=emb_size, out_channels=out_channels, kernel=())
        self.conv_res1 = m.MP_Conv(in_channels=out_channels if self.type == "enc" else in_channels,
                                   out_channels=out_channels, kernel=self.kernel)
        self.conv_res2 = m.MP_Conv(in_channels=out_channels, out_channels=out_channels, kernel=self.kernel)

    def forward(self, x: Tensor, embedding: Tensor) -> Tensor:
        emb = ops.cast(embedding, torch.float32)
        return ops.from_nhwc(unet_block_bank_forward([self], ops.to_nhwc(x), emb, None))


def unet_expert_bank_forward(experts: Sequence["Unet_expert"], x: Tensor, time_emb: Tensor, text_emb: Optional[Tensor],
                             seg: Optional[Tensor], stager=None) -> Tensor:
    """Unet_expert.forward (reference model_components.py:389-433) over a bank of experts.
    x: (R,H,W,C) channel-last rows in expert-contiguous order; time_emb (R,T) / text_emb (R,text_dim) fp32.
    ``stager`` (staged step, hdmoe_hip/graph.py SPLIT_UNET_BWD): the autograd graph is cut with detached leaves behind the full-resolution
    encoder entries, behind the encoder and in front of the full-resolution decoder entries, so that the backward runs as four sections
    whose weight gradients are final -- and can be all-reduced -- one after the other."""
    e0 = experts[0]
    tr = e0.training

    def conv(mods, inp, gain=1.0, **kw):
        return ops.mp_conv(inp, [mm.weights for mm in mods], gain, seg=seg, training=tr, **kw)

    emb = conv([e.map_noise for e in experts], time_emb)
    if e0.map_text is not None and text_emb is not None:
        if text_emb.ndim == 3:
            text_emb = ops.seq_mean(text_emb)
        emb = ops.mp_sum(emb, conv([e.map_text for e in experts], text_emb), e0.label_balance)
    emb = ops.mp_silu(emb)
    # every block's FiLM vector 1 + emb_layer(emb) * gain depends on emb only: all blocks in ONE launch (and one launch each for the
    # input and weight gradients) instead of a linear + an affine per block
    blk = [[e.encoders[n] for e in experts] for n in e0.encoders.keys() if "conv" not in n] + [[e.decoders[n] for e in experts] for n in e0.decoders.keys()]
    if len({b[0].emb_gain for b in blk}) == 1 and len(blk) <= 16:
        films = ops.multi_linear(emb, [[b.emb_layer.weights for b in bs] for bs in blk], blk[0][0].emb_gain, seg=seg, c=1.0, training=tr)
        embs = [None] * len(blk)
    else:
        films = [None] * len(blk)
        embs = list(ops.fanout(emb, len(blk)))                  # ONE gradient sum instead of nblk - 1 adds
    skips = []
    bi = 0
    top = next(iter(e0.encoders.keys())).split("_")[0]          # "32x32": the full-resolution level
    cut_on = stager is not None and getattr(stager, "SPLIT_UNET_BWD", False) and len(e0.block_channels) > 1

    def cut(k):
        # boundary k between two backward sections: everything alive here becomes a leaf (tensors that already are leaves stay)
        nonlocal x, skips, films, embs
        live = [x] + skips + [f for f in films[bi:] if f is not None] + [f for f in embs[bi:] if f is not None]
        out = stager.cut_local(**{f"unet_c{k}": tuple(live)})
        x, out = out[0], out[1:]
        skips, out = list(out[:len(skips)]), out[len(skips):]
        nf = sum(f is not None for f in films[bi:])
        fl, el = iter(out[:nf]), iter(out[nf:])
        films = films[:bi] + [None if f is None else next(fl) for f in films[bi:]]
        embs = embs[:bi] + [None if f is None else next(el) for f in embs[bi:]]

    films, embs = list(films), list(embs)
    names = list(e0.encoders.keys())
    for ni, name in enumerate(names):
        mods = [e.encoders[name] for e in experts]
        if "conv" in name:
            x = conv(mods, x, ones=True)                         # torch.cat([x, ones]) folded into the conv (:416)
        else:
            x = unet_block_bank_forward(mods, x, embs[bi], seg, films[bi])
            bi += 1
        x, sk = ops.fanout(x, 2)
        skips.append(sk)
        if cut_on and name.startswith(top) and ni + 1 < len(names) and not names[ni + 1].startswith(top):
            cut(0)                                               # behind the full-resolution encoder entries
    if cut_on:
        cut(1)                                                   # behind the encoder
    for name in e0.decoders.keys():
        mods = [e.decoders[name] for e in experts]
        if cut_on and name == f"{top}_up":
            cut(2)                                               # in front of the full-resolution decoder entries
        hx = None
        if "block" in name:
            if mods[0].resample == "keep":
                x, hx = ops.mp_cat_silu(x, skips.pop(), e0.concat_balance)       # the concatenation and the block's mp_silu of it: one pass
            else:
                x = ops.mp_cat(x, skips.pop(), e0.concat_balance)
        x = unet_block_bank_forward(mods, x, embs[bi], seg, films[bi], hx)
        bi += 1
    return conv([e.out_conv for e in experts], x, [e.out_gain for e in experts])


class Unet_expert(nn.Module):
    """Magnitude-preserving U-Net expert (reference model_components.py:255-433)."""

    def __init__(self, img_resolution: int, img_channels: int, time_emb_dim: int, text_emb_dim: int, channel_mult: list,
                 model_channels: Optional[int] = 192, channel_mult_emb: Optional[int] = None, num_blocks: Optional[int] = 3,
                 kernel_size: Optional[tuple] = (3, 3), label_balance: Optional[float] = 0.5,
                 concat_balance: Optional[float] = 0.5):
        super().__init__()
        self.block_channels = [model_channels * i for i in channel_mult]
        self.emb_size = model_channels * channel_mult_emb if channel_mult_emb is not None else max(self.block_channels)
        self.label_balance = label_balance
        self.concat_balance = concat_balance
        self.out_gain = nn.Parameter(torch.zeros([]))
        self.map_noise = m.MP_Conv(in_channels=time_emb_dim, out_channels=self.emb_size, kernel=())
        self.map_text = m.MP_Conv(in_channels=text_emb_dim, out_channels=self.emb_size, kernel=()) if text_emb_dim > 0 else None
        self.encoders = nn.ModuleDict()
        self.out_channels = img_channels + 1
        for level, channel in enumerate(self.block_channels):
            res = img_resolution >> level
            if level == 0:
                cin = self.out_channels
                self.out_channels = channel
                self.encoders[f"{res}x{res}_conv"] = m.MP_Conv(in_channels=cin, out_channels=self.out_channels, kernel=kernel_size)
            else:
                self.encoders[f"{res}x{res}_down"] = Unet_block(in_channels=self.out_channels, out_channels=self.out_channels,
                                                                kernel=kernel_size, Type="enc", resample="down",
                                                                emb_size=self.emb_size)
            for i in range(num_blocks):
                cin = self.out_channels
                self.out_channels = channel
                self.encoders[f"{res}x{res}_block{i}"] = Unet_block(in_channels=cin, out_channels=self.out_channels,
                                                                    emb_size=self.emb_size, Type="enc", resample="keep",
                                                                    kernel=kernel_size)
        self.decoders = nn.ModuleDict()
        skips = [block.out_channels for _, block in self.encoders.items()]
        for level, channel in reversed(list(enumerate(self.block_channels))):
            res = img_resolution >> level
            if level == len(self.block_channels) - 1:
                for tag in ("in0", "in1"):
                    self.decoders[f"{res}x{res}_{tag}"] = Unet_block(in_channels=self.out_channels, out_channels=self.out_channels,
                                                                     emb_size=self.emb_size, Type="dec", resample="keep",
                                                                     kernel=kernel_size)
            else:
                self.decoders[f"{res}x{res}_up"] = Unet_block(in_channels=self.out_channels, out_channels=self.out_channels,
                                                              emb_size=self.emb_size, Type="dec", resample="up",
                                                              kernel=kernel_size)
            for i in range(num_blocks + 1):
                cin = self.out_channels + skips.pop()
                self.out_channels = channel
                self.decoders[f"{res}x{res}_block{i}"] = Unet_block(in_channels=cin, out_channels=self.out_channels,
                                                                    emb_size=self.emb_size, Type="dec", resample="keep",
                                                                    kernel=kernel_size)
        self.out_conv = m.MP_Conv(in_channels=self.out_channels, out_channels=img_channels, kernel=kernel_size)

    def forward(self, x: Tensor, time_emb: Tensor, text_emb: Tensor) -> Tensor:
        # fp16 callers (`model.half()`, reference tests/test_model/test_Unet_expert.py:106-115): converted at ingest, computed in fp32 (the
        # parameters through ops.f32_params), returned in fp16
        half = x.dtype == torch.float16
        if half:
            x = ops.cast(x.contiguous(), torch.float32)
        te = ops.cast(time_emb, torch.float32)
        tx = None if text_emb is None else ops.cast(text_emb, torch.float32)
        y = ops.from_nhwc(unet_expert_bank_forward([self], ops.to_nhwc(x), te, tx, None))
        return ops.cast(y, torch.float16) if half else y


class Vit_block(nn.Module):
    """DiffiT-style block: GN -> silu -> linear -> LN -> TMSA -> LN -> MLP with MP residuals
    (reference model_components.py:435-562)."""

    def __init__(self, num_heads: int, num_groups: int, num_channels: int, seq_ln: int, emb_dim: int,
                 resample: Optional[str] = "keep", time_dim: Optional[int] = 0, res_balance: Optional[float] = 0.5,
                 attn_balance: Optional[float] = 0.5, gain_s: Optional[float] = 1.0, gain_t: Optional[float] = 1.0):
        super().__init__()
        self.res_balance = res_balance
        self.gain_s = gain_s
        self.gain_t = gain_t
        self.emb_dim = emb_dim
        self.resample = resample
        self.GN = nn.GroupNorm(num_groups=num_groups, num_channels=num_channels)
        self.skip_proj = m.MP_Conv(num_channels, emb_dim, kernel=()) if num_channels != emb_dim else None
        self.linear1 = m.MP_Conv(num_channels, emb_dim, kernel=())
        self.norm1 = nn.LayerNorm(emb_dim)
        self.norm2 = nn.LayerNorm(emb_dim)
        self.TMSA = m.MP_Attention(num_heads=num_heads, emb_dim=emb_dim, seq_ln=seq_ln, time_dim=time_dim,
                                   attn_balance=attn_balance)
        self.linear2 = m.MP_Conv(emb_dim, emb_dim * 4, kernel=())
        self.linear3 = m.MP_Conv(emb_dim * 4, emb_dim, kernel=())

    def forward(self, x: Tensor, time_embedding: Optional[Tensor] = None) -> Tensor:
        if self.resample != "keep":
            raise NotImplementedError("Vit_block: only resample='keep' is used by the reference models")
        t = self.res_balance
        n = ((1.0 - t) ** 2 + t ** 2) ** 0.5
        res_main = x
        h = ops.group_norm(x, self.GN.weight, self.GN.bias, self.GN.num_groups, ops.ACT_MP_SILU, self.GN.eps)
        h = self.linear1._fwd(h, self.gain_s)
        res_attn = h
        y = ops.layer_norm(h, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if time_embedding is not None and time_embedding.ndim == 2:
            time_embedding = time_embedding[:, None, :]
        y = self.TMSA(y, time_embedding=time_embedding, gain_s=self.gain_s, gain_t=self.gain_t)
        y = ops.mp_sum(y, res_attn, t)
        h = ops.layer_norm(y, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.mp_silu(self.linear2._fwd(h, self.gain_s))
        h = self.linear3._fwd(h, self.gain_s, res=y, alpha=(1.0 - t) / n, beta=t / n)          # mp_sum(linear3(.), y, t)
        if self.skip_proj is not None:
            # mp_sum(skip_proj(res_main), h, t): the projection gets weight (1-t), h the weight t
            return self.skip_proj._fwd(res_main, self.gain_s, res=h, alpha=(1.0 - t) / n, beta=t / n)
        return ops.mp_sum(res_main, h, t)


class Vit_expert(nn.Module):
    """Isotropic ViT expert with a per-expert patch size (reference model_components.py:564-706)."""

    def __init__(self, num_heads: int, num_groups: int, in_channels: int, seq_ln: int, emb_dim: int, num_blocks: int,
                 patch_size: int, time_dim: Optional[int] = 0, text_dim: Optional[int] = 0, res_balance: Optional[float] = 0.5,
                 attn_balance: Optional[float] = 0.5, emb_balance: Optional[float] = 0.5, gain_s: Optional[float] = 1.0,
                 gain_t: Optional[float] = 1.0):
        super().__init__()
        self.seq_ln = seq_ln
        self.emb_balance = emb_balance
        self.emb_dim = emb_dim
        self.patch = nn.Conv2d(in_channels=in_channels, out_channels=emb_dim, kernel_size=patch_size, stride=patch_size)
        self.map_txt = m.MP_Conv(in_channels=text_dim, out_channels=time_dim, kernel=()) \
            if text_dim != time_dim and text_dim != 0 else None
        self.pos_emb = nn.Parameter(torch.zeros(1, seq_ln, emb_dim))
        self.diffit = nn.ModuleList()
        for _ in range(num_blocks):
            self.diffit.append(Vit_block(num_heads=num_heads, num_groups=num_groups, num_channels=emb_dim, seq_ln=seq_ln,
                                         emb_dim=emb_dim, resample="keep", time_dim=time_dim, res_balance=res_balance,
                                         attn_balance=attn_balance, gain_s=gain_s, gain_t=gain_t))
        self.norm = nn.LayerNorm(emb_dim)
        self.unpatch_proj = m.MP_Conv(in_channels=emb_dim, out_channels=in_channels * (patch_size ** 2), kernel=())
        self.unpatch = nn.PixelShuffle(upscale_factor=patch_size)

    def _fwd(self, x: Tensor, time_emb: Optional[Tensor], text_emb: Optional[Tensor]) -> Tensor:
        """x channel-last (B,H,W,C) -> (B,H,W,C)."""
        B, H, W, C = x.shape
        p = self.patch.kernel_size[0]
        hp, wp = -(-H // p), -(-W // p)
        assert hp * wp == self.seq_ln, f"Sequence length mismatch: Got {hp * wp}, expected {self.seq_ln}, shape: {(B, self.emb_dim, hp, wp)}"
        tok = ops.patch_embed(x, self.patch.weight, self.patch.bias)              # zero-pad + strided conv + bias
        tok = ops.bias_add(tok, self.pos_emb)
        if time_emb is not None:
            time_emb = ops.cast(time_emb, torch.float32)
        if text_emb is not None:
            text_emb = ops.cast(text_emb, torch.float32)
            if self.map_txt is not None:
                if text_emb.ndim == 3:
                    text_emb = ops.seq_mean(text_emb)
                text_emb = self.map_txt._fwd(text_emb)
            time_emb = ops.mp_sum(time_emb, text_emb, self.emb_balance)
        for block in self.diffit:
            tok = block(tok, time_embedding=time_emb)
        tok = ops.layer_norm(tok, self.norm.weight, self.norm.bias, self.norm.eps)
        tok = self.unpatch_proj._fwd(tok)
        return ops.pixel_shuffle_tokens(tok, H, W, C, p)

    def forward(self, x: Tensor, time_emb: Tensor = None, text_emb: Optional[Tensor] = None) -> Tensor:
        return ops.from_nhwc(self._fwd(ops.to_nhwc(x), time_emb, text_emb))


# ------------------------------------------------------------------------------------------------------------
# ViT expert BANK: all experts of a layer in one launch over ragged (padded) token rows -- see csrc/ragged.hip.
# ------------------------------------------------------------------------------------------------------------
def vit_bank_compatible(experts: Sequence[nn.Module], H: int, W: int) -> bool:
    """True when the experts differ only in patch size (the reference's HDMOEM construction, model_config1.py:120-127)."""
    if not (1 <= len(experts) <= 8 and all(isinstance(e, Vit_expert) for e in experts)):
        return False
    e0 = experts[0]
    b0 = e0.diffit[0] if len(e0.diffit) else None

    def sig(e):
        blk = [(b.res_balance, b.gain_s, b.gain_t, b.emb_dim, b.GN.num_groups, b.GN.num_channels, b.GN.eps, b.skip_proj is None,
                b.TMSA.num_heads, b.TMSA.attn_balance, b.TMSA.time_dependent, b.norm1.eps, b.norm2.eps, b.resample) for b in e.diffit]
        return (e.emb_dim, e.emb_balance, e.map_txt is None, e.norm.eps, e.patch.in_channels, tuple(blk),
                None if e.map_txt is None else tuple(e.map_txt.weights.shape))
    if any(sig(e) != sig(e0) for e in experts) or b0 is None or any(b.resample != "keep" for b in e0.diffit):
        return False
    # domains of the ragged kernels (csrc/ragged.hip, attn_rag_* in csrc/attention.hip): outside them the per-expert path runs instead
    heads = b0.TMSA.num_heads
    if e0.emb_dim > 256 or b0.GN.num_groups > 32 or e0.emb_dim % heads or (e0.emb_dim // heads) not in (1, 2, 4, 8, 16, 32):
        return False
    if 2 * max(e.seq_ln for e in experts) * (e0.emb_dim // heads) * 4 > 60 * 1024:      # K / V of one (row, head) in LDS
        return False
    for e in experts:
        p = e.patch.kernel_size[0]
        L = (-(-H // p)) * (-(-W // p))
        if L != e.seq_ln or e.pos_emb.shape[1] != L or any(b.TMSA.rel_pos_bias.shape[1] < L for b in e.diffit):
            return False
    return True


def vit_block_bank_forward(blocks: Sequence["Vit_block"], tok: Tensor, t, rag, tproj=None) -> Tensor:
    """Vit_block.forward (reference model_components.py:520-562) over a bank of blocks; tok (R, Sp, C) ragged rows, t (R, T)."""
    b0 = blocks[0]
    tr = b0.training
    seg = rag.seg
    bal = b0.res_balance
    n = ((1.0 - bal) ** 2 + bal ** 2) ** 0.5

    def lin(mods, inp, gain, **kw):
        return ops.mp_conv(inp, [mm.weights for mm in mods], gain, seg=seg, training=tr, **kw)

    # every tensor with several consumers goes through ops.fanout: its gradient is then ONE fused sum, not n - 1 add launches
    tok, res_main = ops.fanout(tok, 2)
    h = ops.gn_rag(tok, [b.GN.weight for b in blocks], [b.GN.bias for b in blocks], rag, b0.GN.num_groups, ops.ACT_MP_SILU, b0.GN.eps)
    h = lin([b.linear1 for b in blocks], h, b0.gain_s)
    h, res_attn = ops.fanout(h, 2)
    y = ops.ln_rag(h, [b.norm1.weight for b in blocks], [b.norm1.bias for b in blocks], rag, b0.norm1.eps)
    # TMSA: time-modulated self-attention (MP_Attention.forward, reference model_internals.py:338-409)
    at = [b.TMSA for b in blocks]
    a0 = at[0]
    ab = a0.attn_balance
    an = ((1.0 - ab) ** 2 + ab ** 2) ** 0.5
    # (the residual consumers hand back an unscaled gradient -- res_grad_raw -- and the factor is applied inside the fan-out's sum)
    yq, yk, yv, yr = ops.fanout(y, 4, scales=(1.0, 1.0, 1.0, (1.0 - ab) / an))
    q = lin([a.q_proj for a in at], yq, b0.gain_s)
    k = lin([a.k_proj for a in at], yk, b0.gain_s)
    v = lin([a.v_proj for a in at], yv, b0.gain_s)
    if tproj is not None:                                         # q_time / k_time / v_time of all blocks were computed in one launch
        q, k, v = ops.seq_bcast_add(q, tproj[0]), ops.seq_bcast_add(k, tproj[1]), ops.seq_bcast_add(v, tproj[2])
    elif a0.time_dependent and t is not None:
        tq, tk, tv = t if isinstance(t, (list, tuple)) else ops.fanout(t, 3)
        q = ops.seq_bcast_add(q, lin([a.q_time for a in at], tq, b0.gain_t))
        k = ops.seq_bcast_add(k, lin([a.k_time for a in at], tk, b0.gain_t))
        v = ops.seq_bcast_add(v, lin([a.v_time for a in at], tv, b0.gain_t))
    o = ops.attention_rag(q, k, v, [a.rel_pos_bias for a in at], rag, a0.num_heads)
    y = lin([a.out_proj for a in at], o, b0.gain_s, res=yr, alpha=ab / an, beta=(1.0 - ab) / an, res_grad_raw=True)
    y = ops.mp_sum(y, res_attn, bal)
    y, yn = ops.fanout(y, 2, scales=(bal / n, 1.0))
    h = ops.ln_rag(yn, [b.norm2.weight for b in blocks], [b.norm2.bias for b in blocks], rag, b0.norm2.eps)
    h = ops.mp_silu(lin([b.linear2 for b in blocks], h, b0.gain_s))
    h = lin([b.linear3 for b in blocks], h, b0.gain_s, res=y, alpha=(1.0 - bal) / n, beta=bal / n, res_grad_raw=True)
    if b0.skip_proj is not None:
        return lin([b.skip_proj for b in blocks], res_main, b0.gain_s, res=h, alpha=(1.0 - bal) / n, beta=bal / n)
    return ops.mp_sum(res_main, h, bal)


def vit_expert_bank_forward(experts: Sequence["Vit_expert"], x: Tensor, time_emb: Optional[Tensor], text_emb: Optional[Tensor],
                            seg: Tensor) -> Tensor:
    """Vit_expert.forward (reference model_components.py:655-706) over a bank of experts that differ in patch size.
    x: (R,H,W,C) channel-last rows in expert-contiguous order (rows [seg[g], seg[g+1]) belong to expert g); time_emb (R,T) /
    text_emb (R,text_dim) fp32.  Patch embedding and un-patching have per-expert shapes and run per expert over all rows (they
    are <2 % of the bank's arithmetic); every layer in between is one launch for all experts."""
    e0 = experts[0]
    tr = e0.training
    R, H, W, C = x.shape
    ps = [e.patch.kernel_size[0] for e in experts]
    rag = ops.RagLayout(seg, [(-(-H // p)) * (-(-W // p)) for p in ps], R)
    pes = [ops.patch_embed(xe, e.patch.weight, e.patch.bias) for xe, e in zip(ops.fanout(x, len(experts)), experts)]
    tok = ops.rag_pack(pes, [e.pos_emb for e in experts], rag)
    t = time_emb
    if text_emb is not None:
        tx = text_emb
        if e0.map_txt is not None:
            tx = ops.mp_conv(tx, [e.map_txt.weights for e in experts], seg=seg, training=tr)
        t = ops.mp_sum(t, tx, e0.emb_balance)
    nb = len(e0.diffit)
    timed = t is not None and e0.diffit[0].TMSA.time_dependent
    tps = ts = None
    if timed and 3 * nb <= 16 and len({b.gain_t for b in e0.diffit}) == 1:
        # the time projections of every block depend on t only: 3 * nb linear layers in one launch
        layers = [[getattr(e.diffit[i].TMSA, nm).weights for e in experts] for i in range(nb) for nm in ("q_time", "k_time", "v_time")]
        tps = ops.multi_linear(t, layers, e0.diffit[0].gain_t, seg=seg, training=tr)
    elif timed:
        ts = list(ops.fanout(t, 3 * nb))
    for i in range(nb):
        tok = vit_block_bank_forward([e.diffit[i] for e in experts], tok, t if ts is None else ts[3 * i:3 * i + 3], rag,
                                     None if tps is None else tps[3 * i:3 * i + 3])
    tok = ops.ln_rag(tok, [e.norm.weight for e in experts], [e.norm.bias for e in experts], rag, e0.norm.eps)
    outs = []
    for e, p, part in zip(experts, ps, ops.rag_unpack(tok, rag)):
        outs.append(ops.pixel_shuffle_tokens(e.unpatch_proj._fwd(part), H, W, C, p))
    return ops.rag_select(outs, rag)
