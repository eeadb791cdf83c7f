"""Shared implementation behind models/model_config1.py and models/model_config2.py (the reference keeps two
near-identical 467-line files; here they are thin subclasses of the bases below).

Dispatch (reference model_config1.py:11-39) is done without the reference's per-expert host syncs
(``mask.any()`` + boolean indexing): a device-side plan permutes the routed samples into expert-contiguous rows,
the U-Net bank runs as grouped launches over those rows, and a weighted combine un-permutes the result.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

import hdmoe_hip
from hdmoe_hip import bank as wbank
from hdmoe_hip import graph as hgraph
from hdmoe_hip import ops
from models import model_components as m
from models import model_internals as util

Tensor = torch.Tensor


# ViT experts as one routed bank over ragged token rows (one launch per layer for all experts; csrc/ragged.hip).
# HDMOE_VIT_BANK=0: every expert on the whole batch on its own side stream (the round-1 path, kept for A/B timing).
VIT_BANK = __import__("os").environ.get("HDMOE_VIT_BANK", "1") != "0"


def _dispatch_nhwc(x: Tensor, experts: nn.ModuleList, out_router: Tensor, time_emb: Tensor, text2d: Optional[Tensor],
                   kcap: Optional[int] = None, stager=None, cut_name: str = "unet") -> Tensor:
    """x channel-last (B,H,W,C) -> (B,H,W,C).  ``stager``: cut the autograd graph between the bank and the combine (staged step); the bank's
    output rows then belong to the backward section of producer ``cut_name``."""
    mods = list(experts)
    E = len(mods)

    def note_usage(plan):
        # rows routed to each expert since the last optimizer step, for the optimizer: an expert without a sample is left out of the update
        # like a grad-None tensor in the reference loop (hdmoe_hip/optim.py FusedAdamW.track_expert_usage).  Every dispatch path writes
        # them -- a pre-installed counter that stayed zero would freeze the expert's parameters -- and they accumulate over the forwards
        # of a step (FusedAdamW.step clears them behind the update).
        if torch.is_grad_enabled() and isinstance(experts, nn.ModuleList):
            u = getattr(experts, "_hdmoe_usage", None)
            if u is None or u.device != x.device or u.numel() != E:
                if torch.cuda.is_current_stream_capturing():
                    return
                u = torch.zeros(E, dtype=torch.float32, device=x.device)
                object.__setattr__(experts, "_hdmoe_usage", u)
            if plan is not None:
                ops.call("hdmoe_seg_counts", u, plan.seg, E)
            else:
                sp = out_router.detach()
                ops.call("hdmoe_route_counts", u, sp if sp.dtype == torch.float32 and sp.is_contiguous() else sp.float().contiguous(), sp.shape[0], E)

    if all(isinstance(e, m.Unet_expert) for e in mods) and E <= 8:
        plan = ops.DispatchPlan(out_router, kcap if kcap is not None else E)
        note_usage(plan)
        xs = ops.gather_rows(x, plan)
        ts = ops.gather_rows(time_emb, plan)
        tx = None if text2d is None else ops.gather_rows(text2d, plan)
        ys = m.unet_expert_bank_forward(mods, xs, ts, tx, plan.seg, stager=stager if cut_name == "unet" else None)
        if stager is not None:
            (ys,) = stager.cut_local(**{cut_name: (ys,)})
        return ops.combine_rows(ys, out_router, plan)
    if VIT_BANK and m.vit_bank_compatible(mods, x.shape[1], x.shape[2]):
        plan = ops.DispatchPlan(out_router, kcap if kcap is not None else E)
        note_usage(plan)
        xs = ops.gather_rows(x, plan)
        ts = ops.gather_rows(time_emb, plan)
        tx = None if text2d is None else ops.gather_rows(text2d, plan)
        ys = m.vit_expert_bank_forward(mods, xs, ts, tx, plan.seg)
        if stager is not None:
            (ys,) = stager.cut_local(**{cut_name: (ys,)})
        return ops.combine_rows(ys, out_router, plan)
    note_usage(None)
    return _combine_weighted(_run_experts(x, mods, time_emb, text2d), out_router)


def _note_usage_sparse(experts: nn.ModuleList, out_router: Tensor) -> None:
    """Per-expert routed-row counts from the sparse gate weights (the path without a dispatch plan); see _dispatch_nhwc.note_usage."""
    if not (torch.is_grad_enabled() and isinstance(experts, nn.ModuleList)):
        return
    E = len(experts)
    u = getattr(experts, "_hdmoe_usage", None)
    if u is None or u.device != out_router.device or u.numel() != E:
        if torch.cuda.is_current_stream_capturing():
            return
        u = torch.zeros(E, dtype=torch.float32, device=out_router.device)
        object.__setattr__(experts, "_hdmoe_usage", u)
    sp = out_router.detach()
    ops.call("hdmoe_route_counts", u, sp if sp.dtype == torch.float32 and sp.is_contiguous() else sp.float().contiguous(), sp.shape[0], E)


def _run_experts(x: Tensor, mods: List[nn.Module], time_emb: Tensor, text2d: Optional[Tensor]):
    """ViT experts have per-expert token counts (heterogeneous patch sizes) and are FLOP-trivial (<1 % of a step) but
    launch-latency-bound (hundreds of few-microsecond kernels): each is evaluated on the whole batch -- still sync-free --
    on its own side stream, so the experts overlap each other and whatever the caller's stream does until the outputs are
    combined.  Returns (outputs, streams to join)."""
    def one(expert):
        if isinstance(expert, m.Vit_expert):
            return expert._fwd(x, time_emb, text2d)
        return ops.to_nhwc(expert(x=ops.from_nhwc(x), time_emb=time_emb, text_emb=text2d))
    if not (ops.SIDE_STREAMS and x.is_cuda and len(mods) > 1):
        return [one(e) for e in mods], None
    main = torch.cuda.current_stream()
    streams = ops.side_streams(x.device, len(mods))
    ys = []
    for s, expert in zip(streams, mods):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            ys.append(one(expert))
    wbank.note_forked_streams(streams)
    return ys, streams


def _combine_weighted(job, out_router: Tensor) -> Tensor:
    ys, streams = job
    if streams is not None:
        main = torch.cuda.current_stream()
        for s in streams:
            main.wait_stream(s)
    out = None
    for i, y in enumerate(ys):
        w = ops.take_col_pos(out_router, i)                       # (B,): weight where routed, exact 0 elsewhere
        y = ops.scale_rows(y, w)
        out = y if out is None else ops.axpby(out, y, 1.0, 1.0)
    return out


def router_to_unet_experts(x: Tensor, experts: nn.ModuleList, out_router: Tensor, time_emb: Tensor,
                           text_emb: Tensor) -> Tensor:
    """Drop-in for the reference helper (model_config1.py:11-39): x logical NCHW, out_router (B,E) sparse weights."""
    text2d = text_emb
    if text_emb is not None:
        text2d = ops.cast(text_emb, torch.float32)
        if text2d.ndim == 3:
            text2d = ops.seq_mean(text2d)
    te = ops.cast(time_emb, torch.float32)
    return ops.from_nhwc(_dispatch_nhwc(ops.to_nhwc(x), experts, out_router, te, text2d))


class _HDMOEMBase(nn.Module):
    """Stem -> two noisy-top-k routers -> U-Net bank + ViT bank -> cross-attention fusion -> text cross-attention
    -> soft gate -> head (reference model_config2.py:42-303 / model_config1.py:42-309)."""

    _has_scaling_net = False

    def __init__(self, IN_in_channels: int, IN_img_resolution: int, internal_channels: int, time_emb_dim: int,
                 text_emb_dim: int, num_experts: int, top_k: int, Fourier_bandwidth: float, VIT_num_blocks: int,
                 VIT_patch_sizes: List[int], VIT_num_groups: int, VIT_num_heads: int, VIT_emb_size: int, Unet_num_blocks: int,
                 Unet_channel_mult: list, Unet_kernel_sizes: List[Tuple[int, int]], Unet_model_channels: Optional[int] = 192,
                 Unet_channel_mult_emb: Optional[int] = None, Unet_label_balance: Optional[float] = 0.5,
                 Unet_concat_balance: Optional[float] = 0.5):
        super().__init__()
        self.internal_channels = internal_channels
        self.top_k = top_k
        self.input_proj = util.MP_Conv(in_channels=IN_in_channels, out_channels=self.internal_channels, kernel=(3, 3))
        self.Fourier_emb = util.MP_Fourier(num_channels=time_emb_dim // 2, bandwidth=Fourier_bandwidth)
        self.out_fourier1 = util.MP_Conv(in_channels=time_emb_dim // 2, out_channels=time_emb_dim * 2, kernel=())
        self.out_fourier2 = util.MP_Conv(in_channels=time_emb_dim * 2, out_channels=time_emb_dim, kernel=())
        if self._has_scaling_net:                                            # registration order as in the reference
            self.scaling_net = m.Scaling_router(emb_dim=time_emb_dim, num_experts=2)
        self.Unet_router = m.Router(in_channels=self.internal_channels, time_dim=time_emb_dim, top_k=top_k, num_experts=num_experts)
        self.vit_router = m.Router(in_channels=self.internal_channels, time_dim=time_emb_dim, top_k=top_k, num_experts=num_experts)
        self.alpha_txt = nn.Parameter(torch.tensor(0.0))
        self.Unet_experts = nn.ModuleList()
        for i in range(num_experts):
            self.Unet_experts.append(m.Unet_expert(img_resolution=IN_img_resolution, img_channels=self.internal_channels,
                                                   time_emb_dim=time_emb_dim, text_emb_dim=text_emb_dim,
                                                   num_blocks=Unet_num_blocks, channel_mult=Unet_channel_mult,
                                                   kernel_size=Unet_kernel_sizes[i], label_balance=Unet_label_balance,
                                                   concat_balance=Unet_concat_balance, model_channels=Unet_model_channels,
                                                   channel_mult_emb=Unet_channel_mult_emb))
        self.VIT_experts = nn.ModuleList()
        for i in range(num_experts):
            self.VIT_experts.append(m.Vit_expert(num_heads=VIT_num_heads, num_groups=VIT_num_groups, in_channels=self.internal_channels,
                                                 seq_ln=math.ceil(IN_img_resolution / VIT_patch_sizes[i]) ** 2,
                                                 emb_dim=VIT_emb_size, num_blocks=VIT_num_blocks, patch_size=VIT_patch_sizes[i],
                                                 text_dim=text_emb_dim, time_dim=time_emb_dim))
        self.cross_attn = util.MP_Attention(num_heads=VIT_num_heads, emb_dim=self.internal_channels, seq_ln=IN_img_resolution ** 2,
                                            context_dim=self.internal_channels, attn_balance=0.5, is_cross_attn=True)
        self.cross_attn_text = util.MP_Attention(num_heads=VIT_num_heads, emb_dim=self.internal_channels,
                                                 seq_ln=IN_img_resolution ** 2, context_dim=text_emb_dim, attn_balance=0.5,
                                                 is_cross_attn=True)
        self.gate1 = util.MP_Conv(in_channels=self.internal_channels * 2, out_channels=self.internal_channels, kernel=(1, 1))
        self.gate2 = util.MP_Conv(in_channels=self.internal_channels, out_channels=2, kernel=(1, 1))
        self.output_proj = util.MP_Conv(in_channels=self.internal_channels, out_channels=IN_in_channels, kernel=(3, 3))

    # -- path scaling: the only place the two variants differ before the fusion ------------------------------------------
    def _scaling(self, time_vec: Tensor, time_embed: Tensor, zeta, **kw):
        raise NotImplementedError

    def _fusion_inputs(self, fu: Tensor, fv: Tensor, s_vit: Tensor, s_unet: Tensor, **kw):
        raise NotImplementedError

    def _fwd(self, x: Tensor, time_vec: Tensor, text_emb: Tensor, Unet_router_mask: Tensor, Vit_router_mask: Tensor, zeta, **kw):
        """x: channel-last fp32 (B,H,W,C_in).  Returns the 7-tuple with `out` and `out_gate` still channel-last."""
        B, H, W, _ = x.shape
        cdt = hdmoe_hip.compute_dtype()
        time_vec = ops.cast(time_vec, torch.float32)
        te = self.Fourier_emb(time_vec)
        te = self.out_fourier1._fwd(te)
        te = self.out_fourier2._fwd(ops.mp_silu(te))
        feats = self.input_proj._fwd(x)                                     # stem stays fp32 (feeds both router trunks)
        s_vit, s_unet, scaling = self._scaling(time_vec, te, zeta, **kw)
        in_unet = ops.scale_rows(feats, s_unet)
        in_vit = ops.scale_rows(feats, s_vit)
        text2d = None
        text_c = None
        if text_emb is not None:
            text32 = ops.cast(text_emb, torch.float32)
            text2d = ops.seq_mean(text32) if text32.ndim == 3 else text32
            text_c = ops.cast(text32, cdt)
        # the ViT experts need only the stem features: fork them now, they run beside the routers and the U-Net bank
        vit_mods = list(self.VIT_experts)
        vit_job = None
        banked = (all(isinstance(e, m.Unet_expert) for e in vit_mods) and len(vit_mods) <= 8) or \
            (VIT_BANK and m.vit_bank_compatible(vit_mods, H, W))
        if not banked:
            vit_job = _run_experts(ops.cast(in_vit, cdt), vit_mods, te, text2d)
        st = hgraph.current()
        if st is not None and banked and x.is_cuda:
            # staged step (hdmoe_hip/graph.py): each branch is its own hipGraph on its own stream, cut out of autograd with detached
            # leaves at the two boundaries; the backward sections are driven by Stager.backward
            if st.SPLIT_ROUTER:
                # the U-Net router on its own stream / graphs: its backward then runs beside the bank's (hdmoe_hip/graph.py SPLIT_ROUTER)
                te_r, in_r = st.cut("ur", pre=(te, in_unet))
                w_unet, p_unet, raw_unet, _ = self.Unet_router._fwd(in_r, te_r, Unet_router_mask, zeta)
                te_b, in_b, w_b = st.cut("unet", pre=(te, in_unet), ur=(w_unet,))
                out_u = _dispatch_nhwc(ops.cast(in_b, cdt), self.Unet_experts, w_b, te_b, text2d, kcap=self.top_k, stager=st)
            else:
                te_u, in_u = st.cut("unet", pre=(te, in_unet))
                (te_r, te_b), (in_r, in_b) = ops.fanout(te_u, 2), ops.fanout(in_u, 2)
                w_unet, p_unet, raw_unet, _ = self.Unet_router._fwd(in_r, te_r, Unet_router_mask, zeta)
                out_u = _dispatch_nhwc(ops.cast(in_b, cdt), self.Unet_experts, w_unet, te_b, text2d, kcap=self.top_k)
            split_vr = st.SPLIT_ROUTER and st.SPLIT_VROUTER
            te_v, in_v = st.cut("vit", pre=(te, in_vit))
            if split_vr:
                # one forward graph, three backward sections (hdmoe_hip/graph.py SPLIT_VROUTER): the router sees its own leaves of the stem
                # tensors, the bank a detached copy of the routing weights; the combine is the boundary between them
                w_vit, p_vit, raw_vit, _ = self.vit_router._fwd(in_v, te_v, Vit_router_mask, zeta)
                te_b, in_b, w_b = st.cut_local(pre=(te, in_vit), vr=(w_vit,))
                out_v = _dispatch_nhwc(ops.cast(in_b, cdt), self.VIT_experts, w_b, te_b, text2d, kcap=self.top_k, stager=st, cut_name="vit")
            else:
                (te_r, te_b), (in_r, in_b) = ops.fanout(te_v, 2), ops.fanout(in_v, 2)
                w_vit, p_vit, raw_vit, _ = self.vit_router._fwd(in_r, te_r, Vit_router_mask, zeta)
                out_v = _dispatch_nhwc(ops.cast(in_b, cdt), self.VIT_experts, w_vit, te_b, text2d, kcap=self.top_k)
            if split_vr:
                out_u, p_unet, raw_unet, out_v, p_vit, raw_vit, s_vit, s_unet, scaling = st.cut(
                    "post", ucomb=(out_u,), ur=(p_unet, raw_unet), vcomb=(out_v,), vr=(p_vit, raw_vit), pre=(s_vit, s_unet, scaling))
            elif st.SPLIT_ROUTER:
                out_u, p_unet, raw_unet, out_v, p_vit, raw_vit, s_vit, s_unet, scaling = st.cut(
                    "post", ucomb=(out_u,), ur=(p_unet, raw_unet), vit=(out_v, p_vit, raw_vit), pre=(s_vit, s_unet, scaling))
            else:
                out_u, p_unet, raw_unet, out_v, p_vit, raw_vit, s_vit, s_unet, scaling = st.cut(
                    "post", unet=(out_u, p_unet, raw_unet), vit=(out_v, p_vit, raw_vit), pre=(s_vit, s_unet, scaling))
        elif banked and ops.SIDE_STREAMS and x.is_cuda:
            w_vit, p_vit, raw_vit, _ = self.vit_router._fwd(in_vit, te, Vit_router_mask, zeta)
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream())                        # the ViT bank depends on nothing issued after this point
            w_unet, p_unet, raw_unet, _ = self.Unet_router._fwd(in_unet, te, Unet_router_mask, zeta)
            out_u = _dispatch_nhwc(ops.cast(in_unet, cdt), self.Unet_experts, w_unet, te, text2d, kcap=self.top_k)
            # the routed ViT bank (a few hundred small launches) on its own stream, beside the U-Net router and bank
            side = ops.side_streams(x.device, 1)[0]
            side.wait_event(fork)
            with torch.cuda.stream(side):
                out_v = _dispatch_nhwc(ops.cast(in_vit, cdt), self.VIT_experts, w_vit, te, text2d, kcap=self.top_k)
            wbank.note_forked_streams([side])
            torch.cuda.current_stream().wait_stream(side)
            out_v.record_stream(torch.cuda.current_stream())               # allocated on the side stream, consumed on this one
        else:
            w_vit, p_vit, raw_vit, _ = self.vit_router._fwd(in_vit, te, Vit_router_mask, zeta)
            w_unet, p_unet, raw_unet, _ = self.Unet_router._fwd(in_unet, te, Unet_router_mask, zeta)
            out_u = _dispatch_nhwc(ops.cast(in_unet, cdt), self.Unet_experts, w_unet, te, text2d, kcap=self.top_k)
            if vit_job is not None:
                _note_usage_sparse(self.VIT_experts, w_vit)
                out_v = _combine_weighted(vit_job, w_vit)
            else:
                out_v = _dispatch_nhwc(ops.cast(in_vit, cdt), self.VIT_experts, w_vit, te, text2d, kcap=self.top_k)
        C = self.internal_channels
        fu = out_u.reshape(B, H * W, C)                                     # channel-last image == (B, S, C) tokens
        fv = out_v.reshape(B, H * W, C)
        q, ctx = self._fusion_inputs(fu, fv, s_vit, s_unet, **kw)
        a = self.cross_attn(query=q, context=ctx, gain_s=1.0, gain_t=1.0)
        at = self.cross_attn_text(query=a, context=text_c, gain_s=1.0, gain_t=1.0)
        a = ops.lerp_param(a, at, self.alpha_txt)                           # a + alpha_txt * (at - a)
        a_img = a.reshape(B, H, W, C)
        g = self.gate1._fwd(ops.mp_cat(out_u, a_img, 0.5))
        g = self.gate2._fwd(ops.mp_silu(g))
        mixed, gate = ops.gate_mix(g, out_u, a_img)                         # softmax gate + mix + mp_sum(out_u, ., 0.5)
        out = self.output_proj._fwd(mixed)
        return out, p_unet, raw_unet, p_vit, raw_vit, scaling, gate

    def _public(self, res):
        out, p_u, raw_u, p_v, raw_v, scaling, gate = res
        return ops.nhwc_to_nchw_f32(out), p_u, raw_u, p_v, raw_v, scaling, ops.from_nhwc(gate)


class _PrecondBase(nn.Module):
    """EDM preconditioning wrapper (reference model_config2.py:306-468)."""

    _net_cls = None

    def __init__(self, sigma_data: Optional[float] = 0.5, log_var_channels: Optional[int] = 128, **net_kwargs):
        super().__init__()
        self.sigma_data = sigma_data
        self.log_var_channels = log_var_channels
        self.num_experts = net_kwargs["num_experts"]
        self.log_var_fourier = util.MP_Fourier(num_channels=self.log_var_channels)
        self.log_var_linear = util.MP_Conv(in_channels=self.log_var_channels, out_channels=1, kernel=())
        self.net = self._net_cls(**net_kwargs)

    def _forward(self, x: Tensor, sigma: Tensor, text_emb: Tensor, Unet_router_mask: Tensor, Vit_router_mask: Tensor, zeta,
                 return_log_var: bool, **kw):
        B = x.shape[0]
        wbank.bank_for(self).begin_step(self.training)                      # all weight images: one launch
        try:
            return self._forward_impl(x, sigma, text_emb, Unet_router_mask, Vit_router_mask, zeta, return_log_var, **kw)
        finally:
            wbank.deactivate()

    def _forward_impl(self, x: Tensor, sigma: Tensor, text_emb: Tensor, Unet_router_mask: Tensor, Vit_router_mask: Tensor, zeta,
                      return_log_var: bool, **kw):
        B = x.shape[0]
        coef = ops.edm_coeffs(sigma, self.sigma_data, B)                    # rows: c_skip, c_out, c_in, c_noise
        c_skip, c_out, c_in, c_noise = coef[0], coef[1], coef[2], coef[3]
        x32 = ops.cast(x, torch.float32)
        xs = ops.to_nhwc(x32, scale=c_in, dtype=torch.float32)              # x * c_in, channel-last
        out, p_u, raw_u, p_v, raw_v, scaling, gate = self.net._fwd(xs, c_noise, text_emb, Unet_router_mask, Vit_router_mask,
                                                                    zeta, **kw)
        # D_x = c_skip * (x * c_in) + c_out * F   (the reference uses the already-scaled x, :440/:449)
        D_x = ops.nhwc_to_nchw_f32(out, c_out, x32, ops.mul(c_skip, c_in))
        log_var = None
        if return_log_var:
            log_var = self.log_var_linear._fwd(self.log_var_fourier(c_noise)).reshape(-1, 1, 1, 1)
        return {"denoised": D_x, "Unet_router_loss": p_u, "Unet_raw": raw_u, "vit_router_loss": p_v, "vit_raw": raw_v,
                "scaling_net_out": scaling, "out_gate": ops.from_nhwc(gate), "log_var": log_var}
