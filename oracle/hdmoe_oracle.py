"""CPU oracle for the HDMOEM denoising hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a functional, plain-PyTorch fp32 restatement of the reference's
MoE-UNet/ViT denoiser (forward; backward comes from torch autograd on CPU).
It is *not* part of the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker /
reported baseline.  The product path (``heterogeneous-moe-for-diffusion-models_amd``)
never imports anything from ``oracle/``.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own
Python modules (in the build container only) and writes golden vectors to
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against every one of them (outputs, top-k indices, gradients).

Shape of the API: every function takes ``P`` -- a flat ``{name: tensor}``
mapping with the reference's ``state_dict`` key layout -- plus a key prefix,
so the oracle can be driven by the state_dict of either the reference modules
or the product's drop-in modules.  Architecture (skip convs, enc/dec type,
resample mode) is inferred from the key names exactly as the reference's
constructors lay them out.

Reference citations are ``path:line`` relative to the reference repo root.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

MP_SILU_DIV = 0.596

# Train-mode switch for the timed CPU baseline (bench.py): dropout p = 0.2 in every Unet_block (model_components.py:245-246), in
# the router trunks (:111) and in Scaling_router (:37), logit noise randn * zeta (:61-62, :155-156).  Parity tests run with it off
# (torch's CPU RNG stream cannot be matched by the device RNG); the in-place weight renormalisation (model_internals.py:254-256) is
# a no-op on the value of the forward and is not re-stated.
TRAIN = {"on": False, "p": 0.2, "zeta": 0.0}


def _drop(x: Tensor) -> Tensor:
    return F.dropout(x, TRAIN["p"], training=True) if TRAIN["on"] else x


def _noise(x: Tensor) -> Tensor:
    return x + torch.randn_like(x) * TRAIN["zeta"] if TRAIN["on"] and TRAIN["zeta"] else x


# --------------------------------------------------------------------------
# L0: magnitude-preserving primitives            models/model_internals.py
# --------------------------------------------------------------------------
def normalize(x: Tensor, dim: Optional[Sequence[int]] = None, eps: float = 1e-4) -> Tensor:
    """x / (eps + ||x||_2 * sqrt(n_norm / n_x)); norm in fp32 (model_internals.py:8-30)."""
    if dim is None:
        dim = list(range(1, x.ndim))
    n = torch.linalg.vector_norm(x, dim=list(dim), keepdim=True, dtype=torch.float32)
    n = eps + n * math.sqrt(n.numel() / x.numel())
    return x / n.to(x.dtype)


def mp_silu(x: Tensor) -> Tensor:
    """silu(x)/0.596 (model_internals.py:33-47)."""
    return F.silu(x) / MP_SILU_DIV


def mp_sum(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """lerp(a,b,t)/sqrt((1-t)^2+t^2) (model_internals.py:50-66)."""
    return torch.lerp(a, b, t) / math.sqrt((1.0 - t) ** 2 + t ** 2)


def mp_cat_scales(na: int, nb: int, t: float = 0.5) -> Tuple[float, float]:
    """The two channel-block weights of mp_cat (model_internals.py:87-91)."""
    c = math.sqrt((na + nb) / ((1.0 - t) ** 2 + t ** 2))
    return c * (1.0 - t) / math.sqrt(na), c * t / math.sqrt(nb)


def mp_cat(a: Tensor, b: Tensor, dim: int = 1, t: float = 0.5) -> Tensor:
    """Weighted concat (model_internals.py:69-92)."""
    wa, wb = mp_cat_scales(a.shape[dim], b.shape[dim], t)
    return torch.cat([wa * a, wb * b], dim=dim)


def resample(x: Tensor, mode: str = "keep", f: Sequence[float] = (1.0, 1.0)) -> Tensor:
    """Separable even-length FIR resampling (model_internals.py:95-127).  f=[1,1]:
    'down' is the depthwise stride-2 conv with 0.25 taps (== 2x2 mean pool); 'up'
    the depthwise transposed conv with unit taps (== nearest x2)."""
    if mode == "keep":
        return x
    f1 = torch.tensor(list(f), dtype=torch.float32)
    assert f1.ndim == 1 and f1.shape[0] % 2 == 0
    pad = (f1.shape[0] - 1) // 2
    f1 = f1 / f1.sum()
    c = x.shape[1]
    k = torch.outer(f1, f1)[None, None].to(x.dtype).repeat(c, 1, 1, 1)
    if mode == "down":
        return F.conv2d(x, k, stride=2, groups=c, padding=pad)
    if mode == "up":
        return F.conv_transpose2d(x, k * 4, stride=2, groups=c, padding=pad)
    raise ValueError(mode)


def mp_fourier(x: Tensor, freqs: Tensor, phases: Tensor) -> Tensor:
    """sqrt(2)*cos(x (x) freqs + phases), fp32 (model_internals.py:158-175)."""
    if x.ndim != 1:
        raise RuntimeError("mp_fourier expects a 1-D input")
    y = torch.outer(x.float(), freqs.float()) + phases.float()
    return (y.cos() * math.sqrt(2.0)).to(x.dtype)


def mp_weight(w: Tensor, gain=1.0) -> Tensor:
    """Effective weight of an MP_Conv: normalize(w) * gain / sqrt(fan_in)
    (model_internals.py:258-259)."""
    w = normalize(w.float())
    return w * (gain / math.sqrt(w[0].numel()))


def mp_conv(x: Tensor, w: Tensor, gain=1.0, stride: int = 1) -> Tensor:
    """MP_Conv.forward in eval mode (model_internals.py:253-275): linear for
    2-D inputs; stride-1 conv with explicit 'same' padding (left (k-1)//2,
    right the rest, k = last kernel dim) for 4-D inputs; stride > 1: padding
    k // 2 (:272-275).  No bias."""
    we = mp_weight(w, gain).to(x.dtype)
    if x.ndim == 2:
        return F.linear(x, we)
    assert x.ndim == 4
    k = we.shape[-1]
    if stride != 1:
        return F.conv2d(x, we, padding=k // 2, stride=stride)
    lo = (k - 1) // 2
    hi = (k - 1) - lo
    return F.conv2d(F.pad(x, (lo, hi, lo, hi)), we)


def pos_encoding(P: "Params", pre: str, time_vec: Tensor) -> Tensor:
    """Pos_encoding.forward (model_internals.py:178-206): [cos(t f), sin(t f)] -> Linear -> SiLU -> Linear."""
    if time_vec.ndim > 1:
        time_vec = time_vec.flatten()
    args = time_vec[:, None].float() * P[pre + "freq"][None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    h = F.silu(F.linear(emb, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"]))
    return F.linear(h, P[pre + "mlp.2.weight"], P[pre + "mlp.2.bias"])


def _sub(P: Params, prefix: str) -> bool:
    """True if any key lives under prefix."""
    return any(k.startswith(prefix) for k in P)


def mp_attention(P: Params, pre: str, query: Tensor, gain_s: float, gain_t: float,
                 num_heads: int, context: Optional[Tensor] = None,
                 time_embedding: Optional[Tensor] = None, attn_balance: float = 0.5) -> Tensor:
    """MP_Attention.forward (model_internals.py:354-409).  Cross-vs-self is
    inferred from the presence of ``rel_pos_bias`` (self only, :324)."""
    B, S, E = query.shape
    is_cross = (pre + "rel_pos_bias") not in P
    ctx = query if context is None else context
    D = E // num_heads

    def proj(name: str, inp: Tensor, g: float) -> Tensor:            # 1x1 conv over (B,C,S,1) == per-token linear
        w = mp_weight(P[pre + name + ".weights"], g).flatten(1).to(inp.dtype)
        return inp @ w.t()                                           # (B,S,E)

    q = proj("q_proj", query, gain_s)
    k = proj("k_proj", ctx, gain_s)
    v = proj("v_proj", ctx, gain_s)
    if (pre + "q_time.weights") in P and time_embedding is not None:   # :368-372
        te = time_embedding.reshape(B, -1)
        q = q + proj("q_time", te, gain_t)[:, None, :]
        if not is_cross:
            k = k + proj("k_time", te, gain_t)[:, None, :]
            v = v + proj("v_time", te, gain_t)[:, None, :]

    def heads(t: Tensor) -> Tensor:                                   # channel c = h*D + d  (:375-377)
        return t.reshape(B, -1, num_heads, D).permute(0, 2, 1, 3)     # (B,H,S,D)

    qh, kh, vh = heads(q), heads(k), heads(v)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(D)                   # :380-381
    if not is_cross:                                                  # :382-399
        bias = P[pre + "rel_pos_bias"]
        if S <= bias.shape[1]:
            bias = bias[:, :S, :S]
        else:
            bias = F.interpolate(bias[None], size=(S, S), mode="bicubic", align_corners=False)[0]
        s = s + bias
    p = s.softmax(dim=-1)
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B, S, E)                 # :402-404
    out = proj("out_proj", o, gain_s)
    return mp_sum(query, out, attn_balance)


# --------------------------------------------------------------------------
# L1: components                                  models/model_components.py
# --------------------------------------------------------------------------
def group_norm(P: Params, pre: str, x: Tensor, groups: int) -> Tensor:
    return F.group_norm(x, groups, P[pre + "weight"], P[pre + "bias"], eps=1e-5)


def layer_norm(P: Params, pre: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], eps=1e-5)


def scaling_router(P: Params, pre: str, x: Tensor, logit_noise: Optional[Tensor] = None) -> Tensor:
    """Scaling_router.forward, eval (model_components.py:41-66): 2-way soft
    gate whose rows sum to 2."""
    if x.ndim == 3:
        x = x.squeeze(1)
    x = F.relu(group_norm(P, pre + "soft_route.1.", mp_conv(x, P[pre + "soft_route.0.weights"]), 1))
    x = _drop(F.relu(group_norm(P, pre + "soft_route.4.", mp_conv(x, P[pre + "soft_route.3.weights"]), 1)))
    x = _noise(mp_conv(x, P[pre + "linear.weights"]))
    if logit_noise is not None:
        x = x + logit_noise
    return F.softmax(x, dim=-1) * 2.0


def router_head(logits: Tensor, mask: Optional[Tensor], k: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Noisy-top-k head after the noise add (model_components.py:158-168).
    Returns (sparse_weights, gate_probs, masked_logits, topk_indices)."""
    if mask is not None:
        logits = logits.masked_fill(mask == 0, float("-inf"))
    probs = F.softmax(logits, dim=-1)
    vals, idx = torch.topk(logits, k, dim=-1)
    w = F.softmax(vals, dim=-1)
    sparse = torch.zeros_like(logits).scatter(-1, idx, w)
    return sparse, probs, logits, idx


def router_logits(P: Params, pre: str, x: Tensor, time_emb: Tensor) -> Tensor:
    """Router trunk up to the pre-noise logits, eval (model_components.py:140-153)."""
    B = x.shape[0]
    for conv_i, gn_i in ((0, 1), (3, 4), (6, 7)):
        x = mp_conv(x, P[f"{pre}hard_route.{conv_i}.weights"])
        x = F.relu(group_norm(P, f"{pre}hard_route.{gn_i}.", x, 1))
    x = _drop(x.mean(dim=(2, 3)).reshape(B, -1))
    if time_emb.ndim == 3:
        time_emb = time_emb.squeeze(1)
    cond = mp_conv(mp_silu(time_emb), P[pre + "time_linear.weights"])
    gamma, beta = cond.chunk(2, dim=1)
    x = x * (1.0 + gamma) + beta
    return mp_conv(x, P[pre + "linear.weights"])


def router(P: Params, pre: str, x: Tensor, time_emb: Tensor, mask: Optional[Tensor], k: int,
           logit_noise: Optional[Tensor] = None):
    """Router.forward (model_components.py:118-168).  ``logit_noise`` stands in
    for the train-mode ``randn*zeta`` draw so it can be supplied explicitly."""
    logits = _noise(router_logits(P, pre, x, time_emb))
    if logit_noise is not None:
        logits = logits + logit_noise
    sparse, probs, logits, _ = router_head(logits, mask, k)
    return sparse, probs, logits


def unet_block(P: Params, pre: str, x: Tensor, emb: Tensor, kind: str, mode: str,
               residual_balance: float = 0.5) -> Tensor:
    """Unet_block.forward, eval (model_components.py:232-253)."""
    e = 1.0 + mp_conv(emb, P[pre + "emb_layer.weights"])
    x = resample(x, mode)
    has_skip = (pre + "conv_skip.weights") in P
    if kind == "enc":
        if has_skip:
            x = mp_conv(x, P[pre + "conv_skip.weights"])
        x = normalize(x, dim=[1])
    y = mp_conv(mp_silu(x), P[pre + "conv_res1.weights"])
    y = _drop(mp_silu(y * e[:, :, None, None].to(x.dtype)))
    y = mp_conv(y, P[pre + "conv_res2.weights"])
    if kind == "dec" and has_skip:
        x = mp_conv(x, P[pre + "conv_skip.weights"])
    return mp_sum(x, y, residual_balance)


def _ordered_children(P: Params, pre: str) -> List[str]:
    """Child-module names directly under ``pre`` in state_dict (== registration) order."""
    seen: List[str] = []
    for k in P:
        if k.startswith(pre):
            name = k[len(pre):].split(".")[0]
            if name not in seen:
                seen.append(name)
    return seen


def unet_expert(P: Params, pre: str, x: Tensor, time_emb: Tensor, text_emb: Optional[Tensor],
                label_balance: float = 0.5, concat_balance: float = 0.5) -> Tensor:
    """Unet_expert.forward (model_components.py:389-433)."""
    emb = mp_conv(time_emb, P[pre + "map_noise.weights"])
    if (pre + "map_text.weights") in P and text_emb is not None:
        if text_emb.ndim == 3:
            text_emb = text_emb.mean(dim=1)
        emb = mp_sum(emb, mp_conv(text_emb, P[pre + "map_text.weights"]), label_balance)
    emb = mp_silu(emb)
    x = torch.cat([x, torch.ones_like(x[:, :1])], dim=1)
    skips: List[Tensor] = []
    for name in _ordered_children(P, pre + "encoders."):
        bp = f"{pre}encoders.{name}."
        if "conv" in name:                                   # :419-420  '{res}x{res}_conv'
            x = mp_conv(x, P[bp + "weights"])
        else:
            x = unet_block(P, bp, x, emb, "enc", "down" if name.endswith("_down") else "keep")
        skips.append(x)
    for name in _ordered_children(P, pre + "decoders."):
        bp = f"{pre}decoders.{name}."
        if "block" in name:                                  # :426-428
            x = mp_cat(x, skips.pop(), t=concat_balance)
        x = unet_block(P, bp, x, emb, "dec", "up" if name.endswith("_up") else "keep")
    return mp_conv(x, P[pre + "out_conv.weights"], gain=P[pre + "out_gain"])


def vit_block(P: Params, pre: str, x: Tensor, time_embedding: Optional[Tensor], num_heads: int,
              num_groups: int, res_balance: float = 0.5, attn_balance: float = 0.5,
              gain_s: float = 1.0, gain_t: float = 1.0) -> Tensor:
    """Vit_block.forward (model_components.py:525-562)."""
    B, S, C = x.shape
    res_main = x
    h = mp_silu(group_norm(P, pre + "GN.", x.transpose(1, 2), num_groups)).transpose(1, 2)
    h = mp_conv(h.reshape(B * S, C), P[pre + "linear1.weights"], gain_s)
    E = h.shape[-1]
    res_attn = h
    y = layer_norm(P, pre + "norm1.", h).reshape(B, S, E)
    if time_embedding is not None and time_embedding.ndim == 2:
        time_embedding = time_embedding[:, None, :]
    y = mp_attention(P, pre + "TMSA.", y, gain_s, gain_t, num_heads,
                     time_embedding=time_embedding, attn_balance=attn_balance)
    y = mp_sum(y.reshape(B * S, E), res_attn, res_balance)
    h = layer_norm(P, pre + "norm2.", y)
    h = mp_silu(mp_conv(h, P[pre + "linear2.weights"], gain_s))
    h = mp_conv(h, P[pre + "linear3.weights"], gain_s)
    h = mp_sum(h, y, res_balance).reshape(B, S, E)
    if (pre + "skip_proj.weights") in P:
        r = mp_conv(res_main.reshape(B * S, C), P[pre + "skip_proj.weights"], gain_s).reshape(B, S, E)
        return mp_sum(r, h, res_balance)
    return mp_sum(res_main, h, res_balance)


def vit_expert(P: Params, pre: str, x: Tensor, time_emb: Tensor, text_emb: Optional[Tensor],
               num_heads: int, num_groups: int, emb_balance: float = 0.5) -> Tensor:
    """Vit_expert.forward (model_components.py:649-706)."""
    B, C, H0, W0 = x.shape
    pw = P[pre + "patch.weight"]
    p = pw.shape[-1]
    ph, pwd = (p - H0 % p) % p, (p - W0 % p) % p
    if ph or pwd:
        x = F.pad(x, (0, pwd, 0, ph))
    x = F.conv2d(x, pw, P[pre + "patch.bias"], stride=p)
    _, E, hp, wp = x.shape
    S = hp * wp
    assert S == P[pre + "pos_emb"].shape[1], "Sequence length mismatch"
    x = x.flatten(2).transpose(1, 2) + P[pre + "pos_emb"]
    if text_emb is not None:
        if (pre + "map_txt.weights") in P:
            if text_emb.ndim == 3:
                text_emb = text_emb.mean(dim=1)
            text_emb = mp_conv(text_emb, P[pre + "map_txt.weights"])
        time_emb = mp_sum(time_emb, text_emb, emb_balance)
    for name in _ordered_children(P, pre + "diffit."):
        x = vit_block(P, f"{pre}diffit.{name}.", x, time_emb, num_heads, num_groups)
    x = layer_norm(P, pre + "norm.", x).reshape(B * S, E)
    x = mp_conv(x, P[pre + "unpatch_proj.weights"]).reshape(B, S, -1)
    x = F.pixel_shuffle(x.transpose(1, 2).reshape(B, -1, hp, wp), p)
    return x[:, :, :H0, :W0]


# --------------------------------------------------------------------------
# L2: assembly                                    models/model_config{1,2}.py
# --------------------------------------------------------------------------
def dispatch_experts(x: Tensor, sparse_w: Tensor, time_emb: Tensor, text_emb: Optional[Tensor],
                     expert_fn) -> Tensor:
    """router_to_unet_experts (model_config1.py:11-39): per-sample gather by
    ``sparse_w[:, e] > 0``, run expert ``e``, weighted scatter-add."""
    text = text_emb.mean(dim=1) if (text_emb is not None and text_emb.ndim == 3) else text_emb
    out = torch.zeros_like(x)
    for e in range(sparse_w.shape[1]):
        sel = sparse_w[:, e] > 0
        if not bool(sel.any()):
            continue
        y = expert_fn(e, x[sel], time_emb[sel], None if text is None else text[sel])
        out = out.index_put((sel.nonzero(as_tuple=True)[0],), y * sparse_w[sel, e].view(-1, 1, 1, 1),
                            accumulate=True)
    return out


def hdmoem(P: Params, cfg: dict, variant: int, x: Tensor, time_vec: Tensor, text_emb: Tensor,
           unet_mask: Tensor, vit_mask: Tensor, transition_point: float = 0.0, softness: float = 1.0,
           alpha_routing: float = 10.0, pre: str = "net."):
    """HDMOEM.forward, eval.  variant 2: model_config2.py:239-303 (closed-form
    sigmoid scaling, query=UNet / context=ViT); variant 1: model_config1.py:241-309
    (learned Scaling_router + soft query/context swap)."""
    B, _, H, W = x.shape
    C = cfg["internal_channels"]
    nh, ng, k = cfg["VIT_num_heads"], cfg["VIT_num_groups"], cfg["top_k"]
    te = mp_fourier(time_vec, P[pre + "Fourier_emb.freqs"], P[pre + "Fourier_emb.phases"])
    te = mp_conv(te, P[pre + "out_fourier1.weights"])
    te = mp_conv(mp_silu(te), P[pre + "out_fourier2.weights"])
    feats = mp_conv(x, P[pre + "input_proj.weights"])
    if variant == 2:
        vw = torch.sigmoid((time_vec * 4 - transition_point) / softness).view(-1, 1, 1, 1)
        s_vit = (vw + 1e-2) * 2
        s_unet = ((1.0 - vw) + 1e-2) * 2
        scaling = torch.cat([s_vit, s_unet], dim=1).view(-1, 2)
    else:
        scaling = scaling_router(P, pre + "scaling_net.", te)
        s_vit = scaling[:, 0:1].view(-1, 1, 1, 1)
        s_unet = scaling[:, 1:2].view(-1, 1, 1, 1)
    in_unet = s_unet * feats
    in_vit = s_vit * feats
    w_vit, p_vit, raw_vit = router(P, pre + "vit_router.", in_vit, te, vit_mask, k)
    w_unet, p_unet, raw_unet = router(P, pre + "Unet_router.", in_unet, te, unet_mask, k)
    lb = cfg.get("Unet_label_balance", 0.5)
    cb = cfg.get("Unet_concat_balance", 0.5)
    out_u = dispatch_experts(in_unet, w_unet, te, text_emb,
                             lambda e, xs, ts, tx: unet_expert(P, f"{pre}Unet_experts.{e}.", xs, ts, tx, lb, cb))
    out_v = dispatch_experts(in_vit, w_vit, te, text_emb,
                             lambda e, xs, ts, tx: vit_expert(P, f"{pre}VIT_experts.{e}.", xs, ts, tx, nh, ng))
    fu = out_u.flatten(2).transpose(1, 2)
    fv = out_v.flatten(2).transpose(1, 2)
    if variant == 2:
        q, ctx = fu, fv
    else:
        sw = torch.sigmoid(alpha_routing * (s_vit - s_unet)).view(-1, 1, 1)
        q = sw * fv + (1 - sw) * fu
        ctx = sw * fu + (1 - sw) * fv
    a = mp_attention(P, pre + "cross_attn.", q, 1.0, 1.0, nh, context=ctx)
    at = mp_attention(P, pre + "cross_attn_text.", a, 1.0, 1.0, nh, context=text_emb)
    a = a + P[pre + "alpha_txt"] * (at - a)
    a_img = a.transpose(1, 2).reshape(B, C, H, W)
    g = mp_conv(mp_cat(out_u, a_img, dim=1), P[pre + "gate1.weights"])
    g = F.softmax(mp_conv(mp_silu(g), P[pre + "gate2.weights"]), dim=1)
    mixed = g[:, 0:1] * out_u + g[:, 1:2] * a_img
    out = mp_conv(mp_sum(out_u, mixed, 0.5), P[pre + "output_proj.weights"])
    return out, p_unet, raw_unet, p_vit, raw_vit, scaling, g


def preconditioned_hdmoem(P: Params, cfg: dict, variant: int, x: Tensor, sigma: Tensor, text_emb: Tensor,
                          unet_mask: Tensor, vit_mask: Tensor, transition_point: float = 0.0,
                          softness: float = 1.0, return_log_var: bool = False,
                          alpha_routing: float = 10.0) -> Dict[str, Optional[Tensor]]:
    """preconditioned_HDMOEM.forward (model_config2.py:431-468): EDM c_skip /
    c_out / c_in / c_noise; NB ``D_x`` uses the already c_in-scaled x (:440,:449)."""
    sd = cfg.get("sigma_data", 0.5)
    sigma = sigma.to(torch.float32)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
    c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.flatten().log() / 4
    if c_noise.shape[0] == 1 and x.shape[0] > 1:
        c_noise = c_noise.expand(x.shape[0])
    x = x * c_in
    out, p_u, raw_u, p_v, raw_v, scaling, gate = hdmoem(
        P, cfg, variant, x, c_noise, text_emb, unet_mask, vit_mask, transition_point, softness, alpha_routing)
    res = {"denoised": c_skip * x + c_out * out, "Unet_router_loss": p_u, "Unet_raw": raw_u,
           "vit_router_loss": p_v, "vit_raw": raw_v, "scaling_net_out": scaling, "out_gate": gate,
           "log_var": None}
    if return_log_var:
        lv = mp_fourier(c_noise, P["log_var_fourier.freqs"], P["log_var_fourier.phases"])
        res["log_var"] = mp_conv(lv, P["log_var_linear.weights"]).reshape(-1, 1, 1, 1)
    return res


# --------------------------------------------------------------------------
# L3 neighbours of the path (loss / input generation)          Utils/utils.py
# --------------------------------------------------------------------------
def edm_loss(out: Dict[str, Optional[Tensor]], target: Tensor, num_experts: int, unet_bal: float,
             vit_bal: float, z_bal: float) -> Dict[str, Tensor]:
    """EDM_LOSS.__call__ (utils.py:127-156) with lamda = 1."""
    err = (out["denoised"] - target) ** 2
    if out["log_var"] is None:
        pure = err.mean()
    else:
        lv = out["log_var"].clamp(min=-10, max=10)
        pure = (err / lv.exp() + lv).mean()
    pure = pure.clamp(max=50)

    def balance(p: Tensor) -> Tensor:                       # utils.py:158-161
        return num_experts * (p.mean(dim=0) ** 2).sum()

    def z(l: Tensor) -> Tensor:                             # utils.py:167-172
        return (torch.logsumexp(l.clamp(min=-50, max=50), dim=-1) ** 2).clamp(max=100).mean()

    bal = (unet_bal * balance(out["Unet_router_loss"]) + vit_bal * balance(out["vit_router_loss"])).clamp(max=50)
    zl = (z_bal * z(out["Unet_raw"]) + z_bal * z(out["vit_raw"])).clamp(max=50)
    return {"loss": (pure + zl + bal).clamp(max=50), "denoising": err.mean(), "balance": bal,
            "z_loss": zl, "pure_loss": pure}


def sample_sigma_hybrid(batch: int, sigma_min=0.002, sigma_max=80.0, p_mean=-0.4, p_std=1.0,
                        extreme_prob=0.2, generator: Optional[torch.Generator] = None) -> Tensor:
    """utils.py:26-61 (log-normal core + log-uniform tail, shuffled)."""
    n_ln = int(batch * (1 - extreme_prob))
    ln = (torch.randn(n_ln, 1, 1, 1, generator=generator) * p_std + p_mean).exp()
    u = torch.rand(batch - n_ln, 1, 1, 1, generator=generator)
    lu = (u * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min)).exp()
    s = torch.cat([ln, lu], 0).clamp(sigma_min, sigma_max)
    return s[torch.randperm(batch, generator=generator)]


def mask_generator(sigma: Tensor, attrs: Sequence[float], p_mean: float, p_std: float, bandwidth: float,
                   min_active: int = 1, noise_range=(0.0, 1.0)) -> Tensor:
    """MaskGenerator.__call__ at a fixed bandwidth (utils.py:262-309)."""
    a = torch.tensor(list(attrs), dtype=torch.float32)
    order = torch.sort(a, stable=True).indices
    centers = torch.zeros_like(a)
    centers[order] = torch.linspace(noise_range[0], noise_range[1], steps=len(a))
    pct = (0.5 * (1 + torch.erf((sigma.flatten().log() - p_mean) / (p_std * math.sqrt(2))))).clamp(0, 1)
    dist = (pct.view(-1, 1) - centers.view(1, -1)).abs()
    mask = (dist <= bandwidth).float()
    mask.scatter_(1, torch.topk(-dist, k=min_active, dim=-1).indices, 1.0)
    return mask


def edm_sampler(denoise, noise: Tensor, num_steps: int, sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7.0) -> Tensor:
    """Deterministic (S_churn = 0) Heun sampler, Utils/EDM_sampler.py:73-109.  ``denoise(x, t)`` -> D(x; t) with ``t`` a
    0-dim tensor; classifier-free guidance (:57-70) is the caller's ``ref.lerp(D, g)`` inside ``denoise``."""
    i = torch.arange(num_steps, dtype=noise.dtype)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    t = torch.cat([t, torch.zeros_like(t[:1])])
    x = noise * t[0]
    for k in range(num_steps):
        d = (x - denoise(x, t[k])) / t[k]
        xn = x + (t[k + 1] - t[k]) * d
        if k < num_steps - 1:
            dp = (xn - denoise(xn, t[k + 1])) / t[k + 1]
            xn = x + (t[k + 1] - t[k]) * (0.5 * d + 0.5 * dp)
        x = xn
    return x


def cfg_lerp(d_model: Tensor, d_guide: Tensor, guidance: float) -> Tensor:
    """Utils/EDM_sampler.py:57-70: ref_D_x.lerp(D_x, guidance)."""
    return d_guide.lerp(d_model, guidance) if guidance != 1.0 else d_model
