"""Generate the golden vectors in tests/golden/ from the REFERENCE's own Python.

TEST INFRASTRUCTURE ONLY -- runs in the build container (where /root/reference
is mounted), never on the GPU box and never from the product path.

    python oracle/make_golden.py            # rewrites tests/golden/*.pt

The reference modules are imported from /root/reference unmodified; every
fixture stores inputs, the module's state_dict, outputs and (where stated)
gradients produced by the reference on CPU in fp32, eval() mode.  The fixtures
are data only (tensors + plain-python configs): no reference source is copied.
"""
import os
import sys

import torch

REF = os.environ.get("HDMOE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "Utils"))

import models.model_internals as mi          # noqa: E402  (reference)
import models.model_components as mc         # noqa: E402  (reference)
import models.model_config1 as c1            # noqa: E402  (reference)
import models.model_config2 as c2            # noqa: E402  (reference)
import utils as ru                           # noqa: E402  (reference Utils/utils.py)
import EDM_sampler as rs                     # noqa: E402  (reference Utils/EDM_sampler.py)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

TINY = dict(IN_in_channels=4, IN_img_resolution=16, internal_channels=8, time_emb_dim=16, text_emb_dim=32,
            num_experts=4, top_k=2, Fourier_bandwidth=1.0, VIT_num_blocks=1, VIT_patch_sizes=[2, 4, 4, 8],
            VIT_num_groups=2, VIT_num_heads=2, VIT_emb_size=8, Unet_num_blocks=1, Unet_channel_mult=[1, 2],
            Unet_kernel_sizes=[(3, 3), (3, 3), (5, 5), (5, 5)], Unet_model_channels=8,
            Unet_channel_mult_emb=2, sigma_data=0.5, log_var_channels=8)
LOSS = dict(unet_bal=0.05, vit_bal=0.1, z_bal=0.005)


def sd(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


def wake_zero_inits(mod, gen):
    """Zero-initialised learnables would make expert outputs identically 0 and
    the fixture vacuous: give them seeded non-trivial values."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if name.endswith("out_gain"):
                p.fill_(0.7)
            elif name.endswith("alpha_txt"):
                p.fill_(0.3)
            elif name.endswith("rel_pos_bias") or name.endswith("pos_emb"):
                p.copy_(0.3 * torch.randn(p.shape, generator=gen))
            elif name.endswith(".weight") and p.ndim == 1:          # GroupNorm / LayerNorm affine
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif name.endswith(".bias") and p.ndim == 1:
                p.copy_(0.2 * torch.randn(p.shape, generator=gen))


def grads_of(mod, names):
    named = dict(mod.named_parameters())
    return {n: (named[n].grad.detach().clone() if named[n].grad is not None else None) for n in names}


def full_model(variant):
    torch.manual_seed(1)
    gen = torch.Generator().manual_seed(7)
    cls = c1.preconditioned_HDMOEM if variant == 1 else c2.preconditioned_HDMOEM
    model = cls(**TINY).eval()
    wake_zero_inits(model, gen)
    B, E = 6, TINY["num_experts"]
    x0 = 0.5 * torch.randn(B, 4, 16, 16, generator=gen)
    sigma = torch.tensor([0.05, 0.3, 0.9, 2.5, 11.0, 60.0]).view(B, 1, 1, 1)
    x = (x0 + sigma * torch.randn(B, 4, 16, 16, generator=gen)).requires_grad_(True)
    text = torch.randn(B, 5, TINY["text_emb_dim"], generator=gen)
    um = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 0], [0, 1, 1, 1], [1, 1, 1, 1], [0, 0, 1, 1], [1, 0, 1, 1.]])
    vm = torch.tensor([[1, 1, 1, 1], [1, 0, 1, 1], [1, 1, 0, 1], [0, 1, 1, 0], [1, 1, 1, 0], [0, 1, 1, 1.]])
    kw = dict(x=x, sigma=sigma, text_emb=text, Unet_router_mask=um, Vit_router_mask=vm, zeta=0.0,
              return_log_var=True)
    extra = {}
    if variant == 2:
        extra = dict(transition_point=-1.2, softness=1.6)
    out = model(**kw, **extra)
    crit = ru.EDM_LOSS(num_experts=E, sigma_data=0.5, Unet_bal=LOSS["unet_bal"], vit_bal=LOSS["vit_bal"],
                       z_bal=LOSS["z_bal"], prior_bal=0.0)
    loss = crit(sigma_vec=sigma, x=x0, sigma=sigma, out_model=out)
    loss["loss"].backward()
    k = TINY["top_k"]
    idx, margin = {}, {}
    for key in ("Unet_raw", "vit_raw"):
        vals, ind = torch.topk(out[key].detach(), k + 1, dim=-1)
        idx[key] = ind[:, :k].clone()
        margin[key] = (vals[:, k - 1] - vals[:, k]).clone()
    pnames = ["net.input_proj.weights", "net.out_fourier1.weights", "net.Unet_router.hard_route.0.weights",
              "net.Unet_router.hard_route.4.weight", "net.vit_router.linear.weights",
              "net.Unet_experts.0.encoders.16x16_conv.weights", "net.Unet_experts.2.decoders.8x8_in0.conv_res1.weights",
              "net.Unet_experts.1.out_gain", "net.VIT_experts.0.diffit.0.TMSA.rel_pos_bias",
              "net.VIT_experts.1.patch.weight", "net.VIT_experts.3.unpatch_proj.weights",
              "net.cross_attn.q_proj.weights", "net.cross_attn_text.k_proj.weights", "net.alpha_txt",
              "net.gate2.weights", "net.output_proj.weights", "log_var_linear.weights"]
    if variant == 1:
        pnames.append("net.scaling_net.soft_route.0.weights")
    fx = dict(variant=variant, cfg=TINY, loss_cfg=LOSS, state=sd(model), x0=x0, sigma=sigma, x=x.detach().clone(),
              text=text, unet_mask=um, vit_mask=vm, extra=extra,
              out={k_: (v.detach().clone() if v is not None else None) for k_, v in out.items()},
              loss={k_: (v.detach().clone() if torch.is_tensor(v) else v) for k_, v in loss.items()},
              topk_idx=idx, topk_margin=margin, x_grad=x.grad.detach().clone(), param_grads=grads_of(model, pnames))
    torch.save(fx, os.path.join(OUT, f"full_config{variant}.pt"))
    print(f"full_config{variant}: denoised {tuple(out['denoised'].shape)} loss {float(loss['loss']):.6f} "
          f"min margin U {float(margin['Unet_raw'].min()):.4f} V {float(margin['vit_raw'].min()):.4f}")


def components():
    gen = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=gen)
    cases = {}

    # ---- L0 free functions -------------------------------------------------
    x = rn(3, 6, 5, 5)
    cases["normalize_default"] = dict(x=x, out=mi.normalize(x))
    cases["normalize_dim1"] = dict(x=x, out=mi.normalize(x, dim=[1]))
    cases["mp_silu"] = dict(x=x, out=mi.mp_silu(x))
    a, b = rn(2, 4, 3, 3), rn(2, 4, 3, 3)
    cases["mp_sum_t03"] = dict(a=a, b=b, t=0.3, out=mi.mp_sum(a, b, 0.3))
    b2 = rn(2, 7, 3, 3)
    cases["mp_cat_t05"] = dict(a=a, b=b2, t=0.5, out=mi.mp_cat(a, b2, dim=1, t=0.5))
    cases["mp_cat_t07"] = dict(a=a, b=b2, t=0.7, out=mi.mp_cat(a, b2, dim=1, t=0.7))
    y = rn(2, 3, 8, 6)
    cases["resample_down"] = dict(x=y, out=mi.resample(y, mode="down"))
    cases["resample_up"] = dict(x=y, out=mi.resample(y, mode="up"))
    torch.manual_seed(3)
    fo = mi.MP_Fourier(10, bandwidth=1.5)
    t = rn(5)
    cases["mp_fourier"] = dict(state=sd(fo), x=t, out=fo(t))

    # ---- MP_Conv -------------------------------------------------------------
    for name, cin, cout, kern, shape in [("lin", 12, 7, (), (5, 12)), ("1x1", 6, 9, (1, 1), (2, 6, 5, 4)),
                                         ("3x3", 5, 8, (3, 3), (2, 5, 7, 6)), ("5x5", 4, 6, (5, 5), (2, 4, 9, 8)),
                                         ("4x4even", 3, 5, (4, 4), (2, 3, 6, 7)), ("7x7", 3, 4, (7, 7), (1, 3, 8, 8))]:
        torch.manual_seed(5)
        conv = mi.MP_Conv(cin, cout, kern).eval()
        xx = rn(*shape).requires_grad_(True)
        out = conv(xx, gain=1.3)
        go = rn(*out.shape)
        out.backward(go)
        cases[f"mp_conv_{name}"] = dict(state=sd(conv), x=xx.detach().clone(), gain=1.3, out=out.detach().clone(),
                                        grad_out=go, x_grad=xx.grad.clone(), w_grad=conv.weights.grad.clone())

    # ---- MP_Attention ----------------------------------------------------------
    def attn_case(name, heads, emb, s0, time_dim, ctx_dim, cross, sq, skv, balance=0.5, gain_t=0.8):
        torch.manual_seed(9)
        at = mi.MP_Attention(heads, emb, s0, time_dim=time_dim, context_dim=ctx_dim, attn_balance=balance,
                             is_cross_attn=cross).eval()
        wake_zero_inits(at, gen)
        q = rn(3, sq, emb).requires_grad_(True)
        ctx = rn(3, skv, ctx_dim if ctx_dim else emb).requires_grad_(True) if cross else None
        te = rn(3, 1, time_dim).requires_grad_(True) if time_dim else None
        out = at(q, 1.1, gain_t, context=ctx, time_embedding=te)
        go = rn(*out.shape)
        out.backward(go)
        names = [n for n, _ in at.named_parameters()]
        cases[f"attn_{name}"] = dict(
            state=sd(at), heads=heads, cross=cross, balance=balance, gain_s=1.1, gain_t=gain_t,
            q=q.detach().clone(), ctx=None if ctx is None else ctx.detach().clone(),
            te=None if te is None else te.detach().clone(), out=out.detach().clone(), grad_out=go,
            q_grad=q.grad.clone(), ctx_grad=None if ctx is None else ctx.grad.clone(),
            te_grad=None if te is None else te.grad.clone(), param_grads=grads_of(at, names))

    attn_case("self_time", 2, 8, 9, 6, None, False, 9, 9)
    attn_case("self_slice", 2, 8, 9, 6, None, False, 6, 6)
    attn_case("self_bicubic", 2, 8, 6, 0, None, False, 10, 10)
    attn_case("cross", 4, 16, 12, 0, 16, True, 12, 12)
    attn_case("cross_text", 4, 16, 12, 0, 24, True, 12, 5, balance=0.3)
    attn_case("cross_time_q", 2, 8, 7, 6, 8, True, 7, 11)

    # ---- routers ----------------------------------------------------------------
    for name, k, mask in [("k1", 1, None), ("k2", 2, None),
                          ("k2_masked", 2, torch.tensor([[1, 1, 0, 1, 1], [0, 1, 1, 0, 1], [1, 1, 1, 1, 1], [1, 0, 0, 1, 0.]])),
                          ("k1_allmasked_row", 1, torch.tensor([[1, 1, 1, 1, 1], [0, 0, 0, 0, 0], [1, 0, 1, 0, 1], [0, 0, 0, 1, 1.]]))]:
        torch.manual_seed(13)
        r = mc.Router(in_channels=4, time_dim=6, top_k=k, num_experts=5).eval()
        wake_zero_inits(r, gen)
        xx = rn(4, 4, 8, 8).requires_grad_(True)
        te = rn(4, 6)
        sw, gp, lg = r(x=xx, time_emb=te, zeta=0.0, mask=mask)
        fin = torch.isfinite(gp)
        ((torch.where(fin, gp, torch.zeros_like(gp)) ** 2).sum() + (sw[torch.isfinite(sw)] * 0.37).sum()).backward()
        cases[f"router_{name}"] = dict(state=sd(r), k=k, x=xx.detach().clone(), te=te, mask=mask, sparse=sw.detach().clone(),
                                       probs=gp.detach().clone(), logits=lg.detach().clone(), x_grad=xx.grad.clone(),
                                       idx=torch.topk(lg.detach(), k, dim=-1).indices)
    torch.manual_seed(17)
    s = mc.Scaling_router(emb_dim=6, num_experts=2).eval()
    wake_zero_inits(s, gen)
    te = rn(5, 6)
    cases["scaling_router"] = dict(state=sd(s), x=te, out=s(te, zeta=0.0).detach().clone())

    # ---- UNet blocks / expert -------------------------------------------------------
    def block_case(name, cin, cout, kern, res, typ, hw):
        torch.manual_seed(19)
        blk = mc.Unet_block(cin, cout, kern, emb_size=10, resample=res, Type=typ).eval()
        xx = rn(2, cin, *hw).requires_grad_(True)
        e = rn(2, 10).requires_grad_(True)
        out = blk(xx, e)
        go = rn(*out.shape)
        out.backward(go)
        cases[f"unet_block_{name}"] = dict(state=sd(blk), kind=typ, mode=res, x=xx.detach().clone(), emb=e.detach().clone(),
                                           out=out.detach().clone(), grad_out=go, x_grad=xx.grad.clone(),
                                           emb_grad=e.grad.clone(),
                                           param_grads=grads_of(blk, [n for n, _ in blk.named_parameters()]))

    block_case("enc_keep", 6, 6, (3, 3), "keep", "enc", (8, 8))
    block_case("enc_skip", 4, 8, (3, 3), "keep", "enc", (8, 8))
    block_case("enc_down", 6, 6, (5, 5), "down", "enc", (8, 8))
    block_case("dec_skip", 10, 6, (3, 3), "keep", "dec", (6, 6))
    block_case("dec_up", 6, 6, (3, 3), "up", "dec", (4, 4))

    torch.manual_seed(23)
    ue = mc.Unet_expert(img_resolution=8, img_channels=4, time_emb_dim=6, text_emb_dim=5, channel_mult=[1, 2],
                        model_channels=8, channel_mult_emb=2, num_blocks=1, kernel_size=(3, 3)).eval()
    wake_zero_inits(ue, gen)
    xx, te, tx = rn(2, 4, 8, 8).requires_grad_(True), rn(2, 6), rn(2, 3, 5)
    out = ue(xx, te, tx)
    go = rn(*out.shape)
    out.backward(go)
    cases["unet_expert"] = dict(state=sd(ue), x=xx.detach().clone(), te=te, text=tx, out=out.detach().clone(),
                                grad_out=go, x_grad=xx.grad.clone(),
                                param_grads=grads_of(ue, ["out_gain", "map_text.weights", "encoders.8x8_conv.weights",
                                                          "decoders.4x4_block0.conv_skip.weights"]))
    out_nt = ue(xx.detach(), te, None)
    cases["unet_expert_notext"] = dict(state=sd(ue), x=xx.detach().clone(), te=te, text=None, out=out_nt.detach().clone())

    # ---- ViT block / expert -----------------------------------------------------------
    for name, cch, emb in [("same", 8, 8), ("skip_proj", 6, 8)]:
        torch.manual_seed(29)
        vb = mc.Vit_block(num_heads=2, num_groups=2, num_channels=cch, seq_ln=9, emb_dim=emb, time_dim=6).eval()
        wake_zero_inits(vb, gen)
        xx, te = rn(3, 9, cch).requires_grad_(True), rn(3, 6).requires_grad_(True)
        out = vb(xx, te)
        go = rn(*out.shape)
        out.backward(go)
        cases[f"vit_block_{name}"] = dict(state=sd(vb), heads=2, groups=2, x=xx.detach().clone(), te=te.detach().clone(),
                                          out=out.detach().clone(), grad_out=go, x_grad=xx.grad.clone(),
                                          te_grad=te.grad.clone(),
                                          param_grads=grads_of(vb, [n for n, _ in vb.named_parameters()]))
    for name, res, p in [("div", 8, 4), ("ragged", 10, 4)]:
        torch.manual_seed(31)
        hp = -(-res // p)
        ve = mc.Vit_expert(num_heads=2, num_groups=2, in_channels=4, seq_ln=hp * hp, emb_dim=8, num_blocks=2,
                           patch_size=p, time_dim=6, text_dim=5).eval()
        wake_zero_inits(ve, gen)
        xx, te, tx = rn(2, 4, res, res).requires_grad_(True), rn(2, 6), rn(2, 5)
        out = ve(xx, te, tx)
        go = rn(*out.shape)
        out.backward(go)
        cases[f"vit_expert_{name}"] = dict(state=sd(ve), heads=2, groups=2, x=xx.detach().clone(), te=te, text=tx,
                                           out=out.detach().clone(), grad_out=go, x_grad=xx.grad.clone(),
                                           param_grads=grads_of(ve, ["patch.weight", "patch.bias", "pos_emb",
                                                                     "unpatch_proj.weights", "norm.weight"]))

    # ---- dispatch with an expert that receives no sample ---------------------------------
    torch.manual_seed(37)
    bank = torch.nn.ModuleList([mc.Unet_expert(img_resolution=8, img_channels=4, time_emb_dim=6, text_emb_dim=5,
                                               channel_mult=[1], model_channels=8, channel_mult_emb=1, num_blocks=1,
                                               kernel_size=ks) for ks in [(3, 3), (5, 5), (3, 3)]]).eval()
    wake_zero_inits(bank, gen)
    xx, te, tx = rn(5, 4, 8, 8), rn(5, 6), rn(5, 2, 5)
    w = torch.tensor([[0.6, 0.0, 0.4], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.3, 0.0, 0.7], [0.0, 0.0, 0.0]])
    out = c2.router_to_unet_experts(xx, bank, w, te, tx)
    cases["dispatch_empty_expert"] = dict(state=sd(bank), x=xx, te=te, text=tx, w=w, out=out.detach().clone())

    # ---- loss / input generators (rows N2) ------------------------------------------------
    sig = torch.tensor([0.01, 0.2, 0.5, 1.0, 3.0, 20.0, 79.0])
    mg = ru.MaskGenerator(expert_attributes=[3, 3, 5, 5], p_mean=-1.2, p_std=1.6, bandwidth=0.3, max_bandwidth=0.8,
                          min_active=1, total_steps=5000, step_size=0.1, noise_range=(0.0, 0.6), strat_band="step")
    cases["mask_generator"] = dict(sigma=sig, attrs=[3, 3, 5, 5], p_mean=-1.2, p_std=1.6, bandwidth=0.3,
                                   noise_range=(0.0, 0.6), out=mg(sig, 0))
    torch.save(cases, os.path.join(OUT, "components.pt"))
    print(f"components: {len(cases)} cases")


WIDE_GRADS = ["net.input_proj.weights", "net.output_proj.weights", "net.gate2.weights", "net.alpha_txt",
              "net.Unet_router.hard_route.0.weights", "net.Unet_router.linear.weights", "net.vit_router.time_linear.weights",
              "net.cross_attn.q_proj.weights", "net.cross_attn_text.v_proj.weights", "log_var_linear.weights",
              "net.Unet_experts.{u}.encoders.{R}x{R}_block0.conv_res1.weights", "net.Unet_experts.{u}.out_gain",
              "net.Unet_experts.{u}.decoders.{R}x{R}_block2.conv_skip.weights", "net.Unet_experts.{u}.map_text.weights",
              "net.VIT_experts.{v}.diffit.1.TMSA.q_proj.weights", "net.VIT_experts.{v}.diffit.3.linear2.weights",
              "net.VIT_experts.{v}.patch.bias", "net.VIT_experts.{v}.norm.weight"]


def wide_model(cfg_id, B, seed):
    """BASELINE config `cfg_id` at its real widths (heterogeneous-moe-for-diffusion-models_amd/Utils/configs.py restates the
    reference's Utils/configs.py:3-35 plus the builder-defined 8-expert lists), weights from oracle/recipe.py; stores outputs only."""
    import importlib.util
    from recipe import fill_state, make_inputs
    spec = importlib.util.spec_from_file_location(
        "hdmoe_cfgs", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "heterogeneous-moe-for-diffusion-models_amd", "Utils", "configs.py"))
    cfgs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfgs)
    bc = cfgs.BASELINE_CONFIGS[cfg_id]
    kw = cfgs.model_kwargs(**bc["over"])
    cls = c1.preconditioned_HDMOEM if bc["module"] == 1 else c2.preconditioned_HDMOEM
    model = cls(**kw).eval()
    model.load_state_dict(fill_state(model.state_dict(), seed))
    E, k, R = kw["num_experts"], kw["top_k"], kw["IN_img_resolution"]
    inp = make_inputs(B, kw["IN_in_channels"], R, E, 77, kw["text_emb_dim"], seed)
    x = inp["x"].clone().requires_grad_(True)
    extra = dict(transition_point=-1.2, softness=1.6) if bc["module"] == 2 else {}
    out = model(x=x, sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["unet_mask"], Vit_router_mask=inp["vit_mask"],
                zeta=0.0, return_log_var=True, **extra)
    crit = ru.EDM_LOSS(num_experts=E, sigma_data=0.5, Unet_bal=LOSS["unet_bal"], vit_bal=LOSS["vit_bal"], z_bal=LOSS["z_bal"], prior_bal=0.0)
    loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
    loss["loss"].backward()
    idx, margin = {}, {}
    for key in ("Unet_raw", "vit_raw"):
        vals, ind = torch.topk(out[key].detach(), k + 1, dim=-1)
        idx[key] = ind[:, :k].clone()
        margin[key] = (vals[:, k - 1] - vals[:, k]).clone()
    u = int(idx["Unet_raw"][0, 0]); v = int(idx["vit_raw"][0, 0])          # experts that certainly received a sample
    names = [n.format(u=u, v=v, R=R) for n in WIDE_GRADS]
    fx = dict(cfg_id=cfg_id, B=B, seed=seed, extra=extra, loss_cfg=LOSS,
              out={k_: (v_.detach().clone() if v_ is not None else None) for k_, v_ in out.items()},
              loss={k_: (v_.detach().clone() if torch.is_tensor(v_) else v_) for k_, v_ in loss.items()},
              topk_idx=idx, topk_margin=margin, x_grad=x.grad.detach().clone(), param_grads=grads_of(model, names))
    torch.save(fx, os.path.join(OUT, f"wide_config{cfg_id}.pt"))
    print(f"wide_config{cfg_id}: B={B} loss {float(loss['loss']):.6f} min margin U {float(margin['Unet_raw'].min()):.4f} "
          f"V {float(margin['vit_raw'].min()):.4f} |denoised| {float(out['denoised'].abs().max()):.3f}")


class SamplerMock(torch.nn.Module):
    """Deterministic stand-in for the denoiser (same role as the reference's tests/test_utilities/test_sampler.py:6-23 mock),
    but sensitive to every argument the sampler forwards: sigma (0-dim), the text embedding and transition_point/softness."""

    def __init__(self, a, b):
        super().__init__()
        self.num_experts = 4
        self.a, self.b = a, b

    def forward(self, x, sigma, text_emb, Unet_router_mask, Vit_router_mask, zeta, transition_point, softness,
                return_log_var=False):
        assert sigma.ndim == 0 and Unet_router_mask.shape == (x.shape[0], self.num_experts) and zeta == 0
        s = sigma.to(x.dtype)
        t = text_emb.mean(dim=(1, 2)).view(-1, 1, 1, 1)
        return {"denoised": x * (self.a / (1.0 + s * s)) + self.b * t * torch.tanh(s) + 0.01 * transition_point * softness}


def sampler_fixture():
    """Row N1: trajectories of the REFERENCE EDM_Sampler (Utils/EDM_sampler.py:73-109, CFG :57-70) over SamplerMock."""
    gen = torch.Generator().manual_seed(41)
    noise = torch.randn(3, 4, 8, 8, generator=gen)
    text = torch.randn(3, 5, 16, generator=gen)
    text2 = torch.randn(3, 5, 16, generator=gen)
    unc = torch.randn(3, 5, 16, generator=gen)
    cases = []
    for guide, N, use_unc in [(1.0, 6, False), (2.5, 6, False), (2.5, 5, True), (0.0, 4, True)]:
        m, gnet = SamplerMock(0.9, 0.3), SamplerMock(0.5, -0.2)
        s = rs.EDM_Sampler(m, gnet, num_solve_steps=N, guidance=guide)
        out = s.sample(noise, text, -1.2, 1.6, uncond_text_emb=unc if use_unc else None)
        out2 = s.sample(noise, text2, -1.2, 1.6, uncond_text_emb=unc if use_unc else None)
        den = s.denoise(noise, torch.tensor(1.7), text, -1.2, 1.6, unc if use_unc else None)
        cases.append(dict(guide=guide, N=N, use_unc=use_unc, out=out.clone(), out_text2=out2.clone(), denoise_at_1p7=den.clone()))
    torch.save(dict(noise=noise, text=text, text2=text2, unc=unc, mock=dict(model=(0.9, 0.3), gnet=(0.5, -0.2)), tp=-1.2,
                    softness=1.6, cases=cases), os.path.join(OUT, "sampler.pt"))
    print("sampler:", [(c["guide"], c["N"], float(c["out"].abs().max())) for c in cases])


def logger_fixture():
    """Records written by the reference's graphs/logger.py for a seeded 25-step stream (row N4)."""
    import json, tempfile
    sys.path.insert(0, os.path.join(REF, "graphs"))
    import logger as rlog                                # noqa: E402  (reference graphs/logger.py)
    gen = torch.Generator().manual_seed(31)
    B, E = 12, 4
    steps = []
    for step in range(25):
        sigma = torch.exp(torch.randn(B, generator=gen) * 1.6 - 1.2)
        loss = {k: torch.rand((), generator=gen) for k in ("loss", "denoising", "pure_loss", "balance", "z_loss", "entropy")}
        steps.append(dict(step=step, loss=loss, zeta=0.1 + 0.01 * step, log_var=float(torch.randn((), generator=gen)),
                          lr=1e-3 * (1 - step / 50), sigma=sigma,
                          unet_probs=torch.softmax(3 * torch.randn(B, E, generator=gen), -1),
                          vit_probs=torch.softmax(torch.randn(B, E, generator=gen), -1),
                          scaling=2 * torch.softmax(torch.randn(B, 2, generator=gen), -1),
                          gate=torch.softmax(torch.randn(B, 2, generator=gen), -1)))
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.Unet_experts = torch.nn.ModuleList([torch.nn.Conv2d(3, 5, 3) for _ in range(2)])
            self.VIT_experts = torch.nn.ModuleList([torch.nn.Linear(7, 4) for _ in range(3)])
            self.Unet_router = torch.nn.Linear(6, 4)
            self.vit_router = torch.nn.Linear(6, 4)
            self.cross_attn = torch.nn.Linear(5, 5)
    torch.manual_seed(17)
    net = Net()
    grads = [[torch.randn(p.shape, generator=gen) for p in net.parameters()] for _ in range(25)]
    with tempfile.TemporaryDirectory() as d:
        lg = rlog.Logger(log_dir=d, run_name="fx", log_interval=10)
        for st, gs in zip(steps, grads):
            for p, g in zip(net.parameters(), gs):
                p.grad = g.clone()
            lg.log_training_step(step=st["step"], loss_dict=st["loss"], zeta=st["zeta"], log_var=st["log_var"], lr=st["lr"],
                                 sigma=st["sigma"], p_mean=-1.2, p_std=1.6)
            lg.log_router_statistics(step=st["step"], unet_probs=st["unet_probs"], vit_probs=st["vit_probs"], sigma=st["sigma"],
                                     p_mean=-1.2, p_std=1.6)
            lg.log_scaling_gating(scaling_factors=st["scaling"], gate_weights=st["gate"], sigma=st["sigma"])
            lg.log_gradients(step=st["step"], model=net)
            lg.log_weight_statistics(step=st["step"], model=net)
        files = {k: [json.loads(l) for l in open(getattr(lg, k))] for k in
                 ("main_log_file", "router_log_file", "gradient_log_file", "weight_log_file")}
    torch.save(dict(steps=steps, grads=grads, net_state=sd(net), files=files), os.path.join(OUT, "logger.pt"))
    print("logger:", {k: len(v) for k, v in files.items()})


def round4_pins():
    """Boundary cases SURVEY.md section 8(c) lists that had no fixture before round 4: a non-square 64 x 128 Unet_expert and the
    `.half()` forward (reference tests/test_model/test_Unet_expert.py:95-115), Pos_encoding (tests/test_model/test_encoding_scheme.py),
    and the public-API argument values the shipped configs never use (normalize eps / dims, mp_cat dims, resample filters, MP_Conv stride).
    Written to its own file so the earlier fixtures stay byte-identical."""
    gen = torch.Generator().manual_seed(404)
    rn = lambda *s: torch.randn(*s, generator=gen)
    cases = {}
    torch.manual_seed(41)
    ue = mc.Unet_expert(img_resolution=64, img_channels=4, time_emb_dim=6, text_emb_dim=5, channel_mult=[1, 2],
                        model_channels=8, channel_mult_emb=2, num_blocks=1, kernel_size=(3, 3)).eval()
    wake_zero_inits(ue, gen)
    xx, te, tx = rn(1, 4, 64, 128).requires_grad_(True), rn(1, 6), rn(1, 5)
    out = ue(xx, te, tx)
    go = rn(*out.shape)
    out.backward(go)
    cases["unet_expert_64x128"] = dict(state=sd(ue), x=xx.detach().clone(), te=te, text=tx, out=out.detach().clone(), grad_out=go,
                                       x_grad=xx.grad.clone(),
                                       param_grads=grads_of(ue, ["out_gain", "encoders.64x64_conv.weights", "decoders.32x32_in0.conv_res1.weights"]))
    torch.manual_seed(43)
    uh = mc.Unet_expert(img_resolution=8, img_channels=4, time_emb_dim=6, text_emb_dim=5, channel_mult=[1, 2],
                        model_channels=8, channel_mult_emb=2, num_blocks=1, kernel_size=(3, 3)).eval()
    wake_zero_inits(uh, gen)
    st32 = sd(uh)
    xh, th, txh = rn(2, 4, 8, 8), rn(2, 6), rn(2, 5)
    out32 = uh(xh, th, txh).detach().clone()
    uh = uh.half()
    try:
        outh = uh(xh.half(), th.half(), txh.half()).detach().clone()
    except Exception as e:                                      # CPU half kernels missing in this torch build: dtype / shape pin only
        print("half forward on CPU failed:", type(e).__name__, e)
        outh = None
    cases["unet_expert_half"] = dict(state=st32, x=xh, te=th, text=txh, out_fp32=out32, out_half=outh)
    torch.manual_seed(47)
    pe = mi.Pos_encoding(emb_dim=16, freq_emb_dim=8, max_period=10000).eval()
    t1 = (3.0 * rn(5)).requires_grad_(True)
    o1 = pe(t1)
    g1 = rn(*o1.shape)
    o1.backward(g1)
    cases["pos_encoding"] = dict(state=sd(pe), emb_dim=16, freq_emb_dim=8, t=t1.detach().clone(), out=o1.detach().clone(), grad_out=g1,
                                 param_grads=grads_of(pe, [n for n, _ in pe.named_parameters()]),
                                 out_2d=pe(t1.detach().reshape(5, 1)).detach().clone())
    x = rn(3, 6, 5, 4)
    cases["normalize_eps"] = dict(x=x, eps=1e-2, out=mi.normalize(x, eps=1e-2))
    cases["normalize_dim23"] = dict(x=x, dim=[2, 3], out=mi.normalize(x, dim=[2, 3]))
    cases["normalize_dim1_eps"] = dict(x=x, dim=[1], eps=3e-3, out=mi.normalize(x, dim=[1], eps=3e-3))
    a, b0, b2 = rn(2, 4, 3, 5), rn(3, 4, 3, 5), rn(2, 4, 6, 5)
    cases["mp_cat_dim0"] = dict(a=a, b=b0, dim=0, t=0.3, out=mi.mp_cat(a, b0, dim=0, t=0.3))
    cases["mp_cat_dim2"] = dict(a=a, b=b2, dim=2, t=0.6, out=mi.mp_cat(a, b2, dim=2, t=0.6))
    y = rn(2, 3, 8, 6).requires_grad_(True)
    for mode in ("down", "up"):
        o = mi.resample(y, f=[1, 3, 3, 1], mode=mode)
        g = rn(*o.shape)
        (gy,) = torch.autograd.grad(o, y, g)
        cases[f"resample_f1331_{mode}"] = dict(x=y.detach().clone(), f=[1, 3, 3, 1], out=o.detach().clone(), grad_out=g, x_grad=gy.clone())
    torch.manual_seed(53)
    cs = mi.MP_Conv(5, 8, (3, 3), stride=2).eval()
    xs = rn(2, 5, 9, 8)
    cases["mp_conv_stride2"] = dict(state=sd(cs), x=xs, out=cs(xs, gain=1.1).detach().clone(), gain=1.1)
    torch.save(cases, os.path.join(OUT, "round4.pt"))
    print("round4:", list(cases.keys()), "half:", None if cases["unet_expert_half"]["out_half"] is None else cases["unet_expert_half"]["out_half"].dtype)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if "--round4-only" in sys.argv:
        round4_pins()
        sys.exit(0)
    if "--logger-only" in sys.argv:
        logger_fixture()
        sys.exit(0)
    if "--sampler-only" in sys.argv:
        sampler_fixture()
        sys.exit(0)
    if "--config1-only" in sys.argv:
        wide_model(1, 8, 41)
        sys.exit(0)
    if "--wide-only" not in sys.argv:
        components()
        full_model(1)
        full_model(2)
        logger_fixture()
        sampler_fixture()
        round4_pins()
    wide_model(1, 8, 41)                                  # BASELINE configs[0]: 3-channel, top-1, B = 8
    wide_model(2, 4, 21)
    wide_model(3, 4, 32)
    wide_model(4, 2, 23)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
