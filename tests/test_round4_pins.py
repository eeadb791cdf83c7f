"""Round-4 pins (GPU): the advisor's findings of round 3 and the boundary cases of SURVEY.md section 8(c) that had no test yet.

* experts on the dispatch path WITHOUT a routed bank must still be optimised (their routed-row counters are written on every path,
  accumulate over the forwards of a step and are cleared behind the update);
* eval-mode weight images are re-prepared after every kind of parameter write (FusedAdamW.step, load_state_dict, p.data.copy_ +
  hdmoe_hip.invalidate_weights(), and -- through the sampler's content checksum -- p.data.copy_ alone);
* `.half()` tensors (reference tests/test_model/test_Unet_expert.py:106-115), a non-square 64 x 128 Unet_expert
  (test_Unet_expert.py:95-104), Pos_encoding (tests/test_model/test_encoding_scheme.py), and the public-API argument values the shipped
  configs never use (normalize eps, resample filters, MP_Conv stride, mp_cat dim).
"""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import hdmoe_hip
    hdmoe_hip.lib()
    hdmoe_hip.set_compute_dtype(torch.float32)
    yield
    hdmoe_hip.set_compute_dtype(torch.float32)


def _tiny_cfg(tmp_path=None):
    from Utils import configs
    over = dict(img_resolution=16, internal_channels=8, time_emb_dim=16, text_emb_dim=32, VIT_num_blocks=1, VIT_patch_sizes=[2, 4, 4, 8],
                VIT_num_groups=2, VIT_num_heads=2, VIT_emb_size=8, Unet_num_blocks=1, Unet_model_channels=8, log_var_channels=8, top_k=2)
    mcfg = dict(configs.model_configs, **over, total_steps=10)
    if tmp_path is not None:
        mcfg["save_dir"] = str(tmp_path)
    return configs, mcfg


def _tiny_model(seed=0):
    from models import model_config2
    configs, mcfg = _tiny_cfg()
    torch.manual_seed(seed)
    model = model_config2.preconditioned_HDMOEM(**configs.model_kwargs(mcfg)).to(DEV)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
            if n.endswith("alpha_txt"):
                p.fill_(0.3)
    return configs, mcfg, model


def _eval_out(model, gen_seed=5, B=4):
    gen = torch.Generator(device=DEV).manual_seed(gen_seed)
    x = 0.5 * torch.randn(B, 4, 16, 16, device=DEV, generator=gen)
    sigma = torch.rand(B, 1, 1, 1, device=DEV, generator=gen) + 0.3
    text = torch.randn(B, 5, 32, device=DEV, generator=gen)
    ones = torch.ones(B, 4, device=DEV)
    with torch.no_grad():
        return model(x=x, sigma=sigma, text_emb=text, Unet_router_mask=ones, Vit_router_mask=ones, zeta=0.0, transition_point=-1.2,
                     softness=1.6)["denoised"].clone()


# ------------------------------------------------------------------------------------------- ADVICE (high): experts off the routed bank
@pytest.mark.parametrize("vit_bank", [False, True])
def test_trainer_updates_vit_experts_on_every_dispatch_path(vit_bank, monkeypatch):
    """With HDMOE_VIT_BANK=0 (also: more than 8 experts, mixed expert types, shapes outside the bank's limits) the ViT experts run on the
    whole batch without a dispatch plan.  Trainer pre-installs their routed-row counters; a counter that nobody writes stays 0 and
    mt_adamw would skip the expert for ever."""
    import hdmoe_hip
    from Utils import training
    from models import _assembly
    monkeypatch.setattr(_assembly, "VIT_BANK", vit_bank)
    hdmoe_hip.set_compute_dtype(torch.bfloat16)
    try:
        configs, mcfg, model = _tiny_model()
        tr = training.Trainer(model, mcfg, configs.optim_configs, configs.loss_configs, configs.mask_configs, configs.zeta_configs)
        gen = torch.Generator(device=DEV).manual_seed(1)
        batches = [(0.5 * torch.randn(8, 4, 16, 16, device=DEV, generator=gen), torch.randn(8, 5, 32, device=DEV, generator=gen)) for _ in range(3)]
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        training.train_steps(tr, batches)
        moved = {n for n, p in model.named_parameters() if not torch.equal(p, before[n])}
        for e in range(4):
            assert any(n.startswith(f"net.VIT_experts.{e}.") for n in moved), f"ViT expert {e} frozen (vit_bank={vit_bank})"
            assert any(n.startswith(f"net.Unet_experts.{e}.") for n in moved), f"U-Net expert {e} frozen"
        # the counters were cleared behind the last update
        assert float(tr.buckets.usage.abs().sum()) == 0.0
    finally:
        hdmoe_hip.set_compute_dtype(torch.float32)


def test_usage_counters_accumulate_over_forwards():
    """Two forwards before one optimizer step (gradient accumulation): an expert routed to only in the FIRST forward still counts."""
    from hdmoe_hip import ops
    from models import _assembly
    configs, mcfg, model = _tiny_model()
    net = model.net
    B, E = 6, 4
    x = torch.randn(B, 16, 16, 8, device=DEV).requires_grad_(True)
    te = torch.randn(B, 16, device=DEV)
    w1 = torch.zeros(B, E, device=DEV); w1[:, 0] = 0.6; w1[:, 1] = 0.4
    w2 = torch.zeros(B, E, device=DEV); w2[:, 2] = 1.0
    for w in (w1, w2):
        _assembly._dispatch_nhwc(x, net.Unet_experts, w, te, None, kcap=2)
    u = net.Unet_experts._hdmoe_usage
    assert u.tolist() == [6.0, 6.0, 6.0, 0.0]
    _assembly._note_usage_sparse(net.VIT_experts, w1)
    _assembly._note_usage_sparse(net.VIT_experts, w2)
    assert net.VIT_experts._hdmoe_usage.tolist() == [6.0, 6.0, 6.0, 0.0]


# ------------------------------------------------------------------------------------------- ADVICE (medium): stale eval-mode weight images
def test_eval_weight_images_follow_every_kind_of_parameter_write():
    import hdmoe_hip
    from hdmoe_hip.optim import FusedAdamW
    from models import model_config2
    configs, mcfg, model = _tiny_model()
    model.eval()
    _eval_out(model); _eval_out(model)                       # registers the bank, second call prepares + reuses the images

    def fresh(state):
        m = model_config2.preconditioned_HDMOEM(**configs.model_kwargs(mcfg))
        m.load_state_dict(state)
        m = m.to(DEV).eval()
        _eval_out(m)
        return _eval_out(m)

    # (1) fused optimizer step (raw-pointer writes)
    opt = FusedAdamW(model.parameters(), lr=5e-2)
    for p in model.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    a = _eval_out(model)
    assert torch.equal(a, fresh(model.state_dict())), "stale images after FusedAdamW.step"
    # (2) load_state_dict
    st = {k: (v + 0.05 * torch.randn_like(v) if v.dtype.is_floating_point else v) for k, v in model.state_dict().items()}
    model.load_state_dict(st)
    assert torch.equal(_eval_out(model), fresh(st)), "stale images after load_state_dict"
    # (3) EMA idiom through .data: invisible to Tensor._version -> documented hook
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(0.9)
    hdmoe_hip.invalidate_weights()
    assert torch.equal(_eval_out(model), fresh(model.state_dict())), "stale images after p.data.mul_ + invalidate_weights"
    # (4) a train-mode forward re-normalises the stored weights in place
    model.train()
    gen = torch.Generator(device=DEV).manual_seed(9)
    ones = torch.ones(4, 4, device=DEV)
    model(x=torch.randn(4, 4, 16, 16, device=DEV, generator=gen), sigma=torch.ones(4, 1, 1, 1, device=DEV), text_emb=torch.randn(4, 5, 32, device=DEV, generator=gen),
          Unet_router_mask=ones, Vit_router_mask=ones, zeta=0.0, transition_point=-1.2, softness=1.6)
    model.eval()
    assert torch.equal(_eval_out(model), fresh(model.state_dict())), "stale images after a train-mode forward"


def test_sampler_graph_sees_weight_changes_between_samples():
    """EDM_Sampler(use_graph=True) captures a denoiser evaluation that holds no weight-prepare launch; sample() refreshes the images
    first and compares a content checksum, so even a silent p.data write is picked up."""
    sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd", "Utils"))
    from EDM_sampler import EDM_Sampler
    configs, mcfg, model = _tiny_model()
    model.eval()
    gen = torch.Generator(device=DEV).manual_seed(2)
    noise = torch.randn(2, 4, 16, 16, device=DEV, generator=gen)
    text = torch.randn(2, 5, 32, device=DEV, generator=gen)
    graphed = EDM_Sampler(model, model, num_solve_steps=3, use_graph=True)
    first = graphed.sample(noise, text, -1.2, 1.6)
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(0.8)                                 # no _version bump, no epoch bump
    eager = EDM_Sampler(model, model, num_solve_steps=3)
    import hdmoe_hip
    second = graphed.sample(noise, text, -1.2, 1.6)
    hdmoe_hip.invalidate_weights()
    ref = eager.sample(noise, text, -1.2, 1.6)
    assert float((first - ref).abs().max()) > 1e-4          # the write matters
    err = float((second - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-5, f"graph replay used stale weight images: {err:.3e}"


# ------------------------------------------------------------------------------------------- boundary pins of SURVEY 8(c)
@pytest.fixture(scope="module")
def g4():
    return torch.load(os.path.join(ROOT, "tests", "golden", "round4.pt"), weights_only=False)


def _close(a, b, rel, msg):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{msg}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= rel * scale + 1e-6, f"{msg}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


def _unet_expert(res, state):
    from models import model_components as mc
    ue = mc.Unet_expert(img_resolution=res, img_channels=4, time_emb_dim=6, text_emb_dim=5, channel_mult=[1, 2], model_channels=8,
                        channel_mult_emb=2, num_blocks=1, kernel_size=(3, 3))
    ue.load_state_dict(state)
    return ue.to(DEV).eval()


def test_unet_expert_non_square_64x128(g4):
    """reference tests/test_model/test_Unet_expert.py:95-104 (rectangular image through the down / up sampling), against the reference's values."""
    c = g4["unet_expert_64x128"]
    ue = _unet_expert(64, c["state"])
    x = c["x"].to(DEV).requires_grad_(True)
    out = ue(x, c["te"].to(DEV), c["text"].to(DEV))
    assert out.shape == (1, 4, 64, 128)
    _close(out, c["out"], 1e-4, "64x128 output")
    out.backward(c["grad_out"].to(DEV))
    _close(x.grad, c["x_grad"], 3e-4, "64x128 x.grad")
    named = dict(ue.named_parameters())
    for n, gref in c["param_grads"].items():
        _close(named[n].grad, gref, 3e-4, f"64x128 grad {n}")


def test_unet_expert_half_precision(g4):
    """reference tests/test_model/test_Unet_expert.py:106-115: model.half() + fp16 inputs give an fp16 output of the input's shape.  The
    values are checked too: fp32 arithmetic on the fp16-rounded parameters, against the reference's own fp16 run (loose) and its fp32 run."""
    c = g4["unet_expert_half"]
    ue = _unet_expert(8, c["state"]).half()
    assert ue.out_conv.weights.dtype == torch.float16
    x, te, tx = c["x"].to(DEV).half(), c["te"].to(DEV).half(), c["text"].to(DEV).half()
    with torch.no_grad():
        out = ue(x, te, tx)
    assert out.dtype == torch.float16 and out.shape == x.shape
    _close(out, c["out_fp32"], 1e-2, "half vs the fp32 reference run")
    if c["out_half"] is not None:
        _close(out, c["out_half"], 2e-2, "half vs the reference's fp16 run")
    # gradients reach the fp16 parameters and the fp16 input
    xg = x.clone().requires_grad_(True)
    ue(xg, te, tx).float().square().mean().backward()
    assert xg.grad is not None and xg.grad.dtype == torch.float16 and torch.isfinite(xg.grad.float()).all()
    assert ue.out_conv.weights.grad is not None and ue.out_conv.weights.grad.dtype == torch.float16 and ue.out_gain.grad is not None


def test_pos_encoding_matches_reference(g4):
    import models.model_internals as mi
    c = g4["pos_encoding"]
    pe = mi.Pos_encoding(emb_dim=c["emb_dim"], freq_emb_dim=c["freq_emb_dim"])
    assert list(pe.state_dict().keys()) == list(c["state"].keys())
    pe.load_state_dict(c["state"])
    pe = pe.to(DEV).eval()
    out = pe(c["t"].to(DEV))
    _close(out, c["out"], 2e-5, "Pos_encoding")
    out.backward(c["grad_out"].to(DEV))
    named = dict(pe.named_parameters())
    for n, gref in c["param_grads"].items():
        _close(named[n].grad, gref, 1e-4, f"Pos_encoding grad {n}")
    _close(pe(c["t"].to(DEV).reshape(5, 1)), c["out_2d"], 2e-5, "Pos_encoding, 2-D time input")


def test_public_api_argument_values_off_the_shipped_configs(g4):
    import models.model_internals as mi
    c = g4["normalize_eps"]
    _close(mi.normalize(c["x"].to(DEV), eps=c["eps"]), c["out"], 1e-5, "normalize eps")
    c = g4["normalize_dim23"]
    _close(mi.normalize(c["x"].to(DEV), dim=c["dim"]), c["out"], 1e-5, "normalize dim=[2,3]")
    c = g4["normalize_dim1_eps"]
    _close(mi.normalize(c["x"].to(DEV), dim=c["dim"], eps=c["eps"]), c["out"], 1e-5, "normalize dim=[1], eps")
    with pytest.raises(NotImplementedError):
        mi.normalize(g4["normalize_eps"]["x"].to(DEV), eps=0.0)
    for k in ("mp_cat_dim0", "mp_cat_dim2"):
        c = g4[k]
        _close(mi.mp_cat(c["a"].to(DEV), c["b"].to(DEV), dim=c["dim"], t=c["t"]), c["out"], 1e-6, k)
    for mode in ("down", "up"):
        c = g4[f"resample_f1331_{mode}"]
        x = c["x"].to(DEV).requires_grad_(True)
        o = mi.resample(x, f=c["f"], mode=mode)
        _close(o, c["out"], 1e-5, f"resample f=[1,3,3,1] {mode}")
        o.backward(c["grad_out"].to(DEV))
        _close(x.grad, c["x_grad"], 1e-5, f"resample f=[1,3,3,1] {mode} backward")
    with pytest.raises(AssertionError):
        mi.resample(g4["resample_f1331_up"]["x"].to(DEV), f=[1, 2, 1], mode="down")     # odd length: the reference asserts
    with pytest.raises(ValueError):
        mi.resample(g4["resample_f1331_up"]["x"].to(DEV), mode="sideways")
    # strided MP_Conv: forward and weight gradient on the general strided kernels; the input gradient is the one stub left (pinned)
    c = g4["mp_conv_stride2"]
    conv = mi.MP_Conv(5, 8, (3, 3), stride=2)
    conv.load_state_dict(c["state"])
    conv = conv.to(DEV).eval()
    out = conv(c["x"].to(DEV), gain=c["gain"])
    _close(out, c["out"], 1e-4, "MP_Conv stride 2")
    out.square().mean().backward()
    assert conv.weights.grad is not None and torch.isfinite(conv.weights.grad).all()
    xg = c["x"].to(DEV).requires_grad_(True)
    with pytest.raises(NotImplementedError):
        conv(xg, gain=c["gain"]).sum().backward()
