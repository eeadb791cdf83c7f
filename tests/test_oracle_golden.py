"""Pin the CPU oracle (oracle/hdmoe_oracle.py) against golden vectors produced
by the reference's own Python (oracle/make_golden.py).  CPU only."""
import torch

from oracle import hdmoe_oracle as O

TOL = dict(rtol=1e-5, atol=1e-6)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    torch.testing.assert_close(a, b, equal_nan=True, **kw)


def close_scaled(a, b, rel=2e-5):
    """max|a-b| <= rel * max|b|: for gradients through deep fp32 chains, whose
    element-wise error is set by the largest terms of the sums, not by each element."""
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= rel * max(scale, 1e-6), (float((a - b).abs().max()), scale)


def test_free_functions(golden_components):
    g = golden_components
    close(O.normalize(g["normalize_default"]["x"]), g["normalize_default"]["out"])
    close(O.normalize(g["normalize_dim1"]["x"], dim=[1]), g["normalize_dim1"]["out"])
    close(O.mp_silu(g["mp_silu"]["x"]), g["mp_silu"]["out"])
    c = g["mp_sum_t03"]
    close(O.mp_sum(c["a"], c["b"], c["t"]), c["out"])
    for name in ("mp_cat_t05", "mp_cat_t07"):
        c = g[name]
        close(O.mp_cat(c["a"], c["b"], 1, c["t"]), c["out"])
    close(O.resample(g["resample_down"]["x"], "down"), g["resample_down"]["out"])
    close(O.resample(g["resample_up"]["x"], "up"), g["resample_up"]["out"])
    c = g["mp_fourier"]
    close(O.mp_fourier(c["x"], c["state"]["freqs"], c["state"]["phases"]), c["out"])


def test_mp_conv(golden_components):
    for name in ("lin", "1x1", "3x3", "5x5", "4x4even", "7x7"):
        c = golden_components[f"mp_conv_{name}"]
        x = c["x"].clone().requires_grad_(True)
        w = c["state"]["weights"].clone().requires_grad_(True)
        out = O.mp_conv(x, w, c["gain"])
        close(out, c["out"])
        out.backward(c["grad_out"])
        close(x.grad, c["x_grad"])
        close(w.grad, c["w_grad"], atol=1e-5)


def test_attention(golden_components):
    for name in ("self_time", "self_slice", "self_bicubic", "cross", "cross_text", "cross_time_q"):
        c = golden_components[f"attn_{name}"]
        P = {k: v.clone().requires_grad_(True) for k, v in c["state"].items()}
        q = c["q"].clone().requires_grad_(True)
        ctx = None if c["ctx"] is None else c["ctx"].clone().requires_grad_(True)
        te = None if c["te"] is None else c["te"].clone().requires_grad_(True)
        out = O.mp_attention(P, "", q, c["gain_s"], c["gain_t"], c["heads"], context=ctx, time_embedding=te,
                             attn_balance=c["balance"])
        close(out, c["out"])
        out.backward(c["grad_out"])
        close(q.grad, c["q_grad"], atol=1e-5)
        if ctx is not None:
            close(ctx.grad, c["ctx_grad"], atol=1e-5)
        if te is not None:
            close(te.grad, c["te_grad"], atol=1e-5)
        for n, gref in c["param_grads"].items():
            if gref is not None:
                close(P[n].grad, gref, atol=1e-5)


def test_routers(golden_components):
    for name in ("k1", "k2", "k2_masked", "k1_allmasked_row"):
        c = golden_components[f"router_{name}"]
        x = c["x"].clone().requires_grad_(True)
        sw, gp, lg = O.router(c["state"], "", x, c["te"], c["mask"], c["k"])
        close(lg, c["logits"])
        close(gp, c["probs"])
        rows_ok = torch.isfinite(c["sparse"]).all(dim=1)
        close(sw[rows_ok], c["sparse"][rows_ok])
        assert torch.equal(torch.topk(lg[rows_ok], c["k"], dim=-1).indices, c["idx"][rows_ok])
        fin = torch.isfinite(gp)
        ((torch.where(fin, gp, torch.zeros_like(gp)) ** 2).sum() + (sw[torch.isfinite(sw)] * 0.37).sum()).backward()
        close(x.grad, c["x_grad"], atol=1e-5)
    c = golden_components["scaling_router"]
    close(O.scaling_router(c["state"], "", c["x"]), c["out"])


def test_unet(golden_components):
    for name in ("enc_keep", "enc_skip", "enc_down", "dec_skip", "dec_up"):
        c = golden_components[f"unet_block_{name}"]
        P = {k: v.clone().requires_grad_(True) for k, v in c["state"].items()}
        x = c["x"].clone().requires_grad_(True)
        e = c["emb"].clone().requires_grad_(True)
        out = O.unet_block(P, "", x, e, c["kind"], c["mode"])
        close(out, c["out"])
        out.backward(c["grad_out"])
        close(x.grad, c["x_grad"], atol=1e-5)
        close(e.grad, c["emb_grad"], atol=1e-5)
        for n, gref in c["param_grads"].items():
            close(P[n].grad, gref, atol=1e-5)
    c = golden_components["unet_expert"]
    P = {k: v.clone().requires_grad_(True) for k, v in c["state"].items()}
    x = c["x"].clone().requires_grad_(True)
    out = O.unet_expert(P, "", x, c["te"], c["text"])
    close(out, c["out"], rtol=1e-4, atol=1e-4)
    out.backward(c["grad_out"])
    close_scaled(x.grad, c["x_grad"])                    # ~40-layer fp32 chain: summation-order noise
    for n, gref in c["param_grads"].items():
        close_scaled(P[n].grad, gref)
    c = golden_components["unet_expert_notext"]
    close(O.unet_expert(c["state"], "", c["x"], c["te"], None), c["out"], rtol=1e-4, atol=1e-4)


def test_vit(golden_components):
    for name in ("same", "skip_proj"):
        c = golden_components[f"vit_block_{name}"]
        P = {k: v.clone().requires_grad_(True) for k, v in c["state"].items()}
        x = c["x"].clone().requires_grad_(True)
        te = c["te"].clone().requires_grad_(True)
        out = O.vit_block(P, "", x, te, c["heads"], c["groups"])
        close(out, c["out"])
        out.backward(c["grad_out"])
        close(x.grad, c["x_grad"], atol=1e-5)
        close(te.grad, c["te_grad"], atol=1e-5)
        for n, gref in c["param_grads"].items():
            close(P[n].grad, gref, atol=1e-5)
    for name in ("div", "ragged"):
        c = golden_components[f"vit_expert_{name}"]
        P = {k: v.clone().requires_grad_(True) for k, v in c["state"].items()}
        x = c["x"].clone().requires_grad_(True)
        out = O.vit_expert(P, "", x, c["te"], c["text"], c["heads"], c["groups"])
        close(out, c["out"])
        out.backward(c["grad_out"])
        close_scaled(x.grad, c["x_grad"])
        for n, gref in c["param_grads"].items():
            close_scaled(P[n].grad, gref)


def test_dispatch_and_masks(golden_components):
    c = golden_components["dispatch_empty_expert"]
    out = O.dispatch_experts(c["x"], c["w"], c["te"], c["text"],
                             lambda e, xs, ts, tx: O.unet_expert(c["state"], f"{e}.", xs, ts, tx))
    close(out, c["out"])
    assert float(out[4].abs().max()) == 0.0            # the un-routed sample gets exactly zero
    c = golden_components["mask_generator"]
    m = O.mask_generator(c["sigma"], c["attrs"], c["p_mean"], c["p_std"], c["bandwidth"], 1, c["noise_range"])
    assert torch.equal(m, c["out"])


def test_full_model(golden_full):
    g = golden_full
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g["state"].items()}
    x = g["x"].clone().requires_grad_(True)
    out = O.preconditioned_hdmoem(P, g["cfg"], g["variant"], x, g["sigma"], g["text"], g["unet_mask"],
                                  g["vit_mask"], return_log_var=True, **g["extra"])
    for key, ref in g["out"].items():
        close(out[key], ref, rtol=1e-4, atol=1e-5)
    k = g["cfg"]["top_k"]
    for key in ("Unet_raw", "vit_raw"):                # router indices: bit-exact
        assert torch.equal(torch.topk(out[key], k, dim=-1).indices, g["topk_idx"][key])
    lc = g["loss_cfg"]
    loss = O.edm_loss(out, g["x0"], g["cfg"]["num_experts"], lc["unet_bal"], lc["vit_bal"], lc["z_bal"])
    for key in ("loss", "denoising", "balance", "z_loss", "pure_loss"):
        close(loss[key], g["loss"][key], rtol=1e-5, atol=1e-6)
    loss["loss"].backward()
    close(x.grad, g["x_grad"], rtol=1e-3, atol=1e-6)
    for n, gref in g["param_grads"].items():
        if gref is None:
            assert P[n].grad is None or float(P[n].grad.abs().max()) == 0.0
        else:
            close(P[n].grad, gref, rtol=1e-3, atol=1e-6)


def test_full_model_real_widths(golden_wide):
    """BASELINE configs 2 / 3 / 4 at cfg32 widths (4 and 8 heterogeneous experts incl. 7x7, R = 32 and 64, text 77x768):
    the oracle against the reference's outputs for the same recipe weights."""
    from conftest import wide_setup
    g = golden_wide
    variant, _, kw, state, inp = wide_setup(g)
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in state.items()}
    x = inp["x"].clone().requires_grad_(True)
    out = O.preconditioned_hdmoem(P, kw, variant, x, inp["sigma"], inp["text"], inp["unet_mask"], inp["vit_mask"],
                                  return_log_var=True, **g["extra"])
    k = kw["top_k"]
    for key in ("Unet_raw", "vit_raw"):
        assert torch.equal(torch.topk(out[key], k, dim=-1).indices, g["topk_idx"][key])
    for key, ref in g["out"].items():
        close(out[key], ref, rtol=1e-4, atol=2e-5)
    lc = g["loss_cfg"]
    loss = O.edm_loss(out, inp["x0"], kw["num_experts"], lc["unet_bal"], lc["vit_bal"], lc["z_bal"])
    close(loss["loss"], g["loss"]["loss"], rtol=1e-5, atol=1e-6)
    loss["loss"].backward()
    close_scaled(x.grad, g["x_grad"], 1e-4)
    for n, gref in g["param_grads"].items():
        close_scaled(P[n].grad, gref, 1e-4)


def _mock(a, b, tp, soft):
    return lambda x, s, text: x * (a / (1.0 + s * s)) + b * text.mean(dim=(1, 2)).view(-1, 1, 1, 1) * torch.tanh(s) + 0.01 * tp * soft


def test_sampler_fixture(golden_sampler):
    """Row N1: the oracle's Heun loop + CFG lerp against trajectories of the reference EDM_Sampler (oracle/make_golden.py)."""
    g = golden_sampler
    m, gn = _mock(*g["mock"]["model"], g["tp"], g["softness"]), _mock(*g["mock"]["gnet"], g["tp"], g["softness"])
    for c in g["cases"]:
        unc = g["unc"] if c["use_unc"] else None
        for text, key in ((g["text"], "out"), (g["text2"], "out_text2")):
            den = lambda x, t: O.cfg_lerp(m(x, t, text), gn(x, t, unc if unc is not None else text), c["guide"])
            out = O.edm_sampler(den, g["noise"], c["N"])
            close_scaled(out, c[key], 1e-5)
        den17 = O.cfg_lerp(m(g["noise"], torch.tensor(1.7), g["text"]), gn(g["noise"], torch.tensor(1.7), unc if unc is not None else g["text"]), c["guide"])
        close(den17, c["denoise_at_1p7"])


def test_product_mask_generator_matches_reference(golden_components):
    """Row N2: the product's MaskGenerator (plain torch, device-agnostic host logic) against the reference's masks."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "heterogeneous-moe-for-diffusion-models_amd", "Utils"))
    import utils as U
    c = golden_components["mask_generator"]
    mg = U.MaskGenerator(expert_attributes=c["attrs"], p_mean=c["p_mean"], p_std=c["p_std"], bandwidth=c["bandwidth"], max_bandwidth=0.8,
                         min_active=1, total_steps=5000, step_size=0.1, noise_range=c["noise_range"], strat_band="step")
    assert torch.equal(mg(c["sigma"], 0), c["out"])


def test_round4_boundary_cases():
    """tests/golden/round4.pt (oracle/make_golden.py round4_pins, from the reference): non-square U-Net expert, Pos_encoding, FIR resampling,
    general normalize / mp_cat arguments, strided MP_Conv."""
    import os
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round4.pt"), weights_only=False)
    c = g["unet_expert_64x128"]
    P = {k: v.clone().requires_grad_(True) for k, v in c["state"].items()}
    x = c["x"].clone().requires_grad_(True)
    out = O.unet_expert(P, "", x, c["te"], c["text"])
    assert out.shape == (1, 4, 64, 128)
    close(out, c["out"], rtol=1e-4, atol=1e-4)
    out.backward(c["grad_out"])
    close_scaled(x.grad, c["x_grad"])
    for n, gref in c["param_grads"].items():
        close_scaled(P[n].grad, gref)
    c = g["unet_expert_half"]
    close(O.unet_expert(c["state"], "", c["x"], c["te"], c["text"]), c["out_fp32"], rtol=1e-4, atol=1e-4)
    c = g["pos_encoding"]
    P = {k: v.clone().requires_grad_(v.is_floating_point() and k != "freq") for k, v in c["state"].items()}
    out = O.pos_encoding(P, "", c["t"])
    close(out, c["out"], rtol=1e-5, atol=1e-6)
    out.backward(c["grad_out"])
    for n, gref in c["param_grads"].items():
        close(P[n].grad, gref, rtol=1e-4, atol=1e-6)
    close(O.pos_encoding(c["state"], "", c["t"].reshape(5, 1)), c["out_2d"], rtol=1e-5, atol=1e-6)
    c = g["normalize_eps"]
    close(O.normalize(c["x"], eps=c["eps"]), c["out"])
    c = g["normalize_dim23"]
    close(O.normalize(c["x"], dim=c["dim"]), c["out"])
    c = g["normalize_dim1_eps"]
    close(O.normalize(c["x"], dim=c["dim"], eps=c["eps"]), c["out"])
    for k in ("mp_cat_dim0", "mp_cat_dim2"):
        c = g[k]
        close(O.mp_cat(c["a"], c["b"], dim=c["dim"], t=c["t"]), c["out"])
    for mode in ("down", "up"):
        c = g[f"resample_f1331_{mode}"]
        x = c["x"].clone().requires_grad_(True)
        o = O.resample(x, mode, f=c["f"])
        close(o, c["out"])
        o.backward(c["grad_out"])
        close(x.grad, c["x_grad"])
    c = g["mp_conv_stride2"]
    close(O.mp_conv(c["x"], c["state"]["weights"], c["gain"], stride=2), c["out"], rtol=1e-5, atol=1e-6)
