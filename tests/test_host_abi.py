"""CPU-only checks of the C-ABI boundary and the drop-in surface (no kernel is launched)."""
import ctypes
import os
import re

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from hdmoe_hip import _lib
    hdr = open(os.path.join(ROOT, "include", "hdmoe.h")).read()
    declared = set(re.findall(r"\bint\s+(hdmoe_\w+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES) and len(declared) > 50
    lib = _lib.lib()                                   # raises if the .so is missing or a symbol is not exported
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)
    assert lib.hdmoe_version() >= 100


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing somewhere else."""
    import pytest
    from hdmoe_hip import ops
    with pytest.raises(RuntimeError):
        ops.mp_silu(torch.randn(4, 4))


def test_state_dict_layout_matches_reference(golden_full):
    from models import model_config1, model_config2
    g = golden_full
    cls = (model_config1 if g["variant"] == 1 else model_config2).preconditioned_HDMOEM
    model = cls(**g["cfg"])
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["state"].keys())            # same names, same registration order
    assert all(sd[k].shape == g["state"][k].shape and sd[k].dtype == g["state"][k].dtype for k in sd)
    model.load_state_dict(g["state"])                             # reference checkpoints load unchanged
    net = model.net                                               # attribute names read by training.py / logger.py / plotter.py
    for attr in ("Unet_experts", "VIT_experts", "cross_attn", "Unet_router", "vit_router"):
        assert hasattr(net, attr)
    assert hasattr(net, "scaling_net") == (g["variant"] == 1)
    assert model.num_experts == g["cfg"]["num_experts"]
    e0 = net.Unet_experts[0]
    assert float(type(e0)(img_resolution=8, img_channels=4, time_emb_dim=4, text_emb_dim=4, channel_mult=[1]).out_gain) == 0.0


def test_constructor_signatures_match_reference_positional_use():
    """Reference tests construct positionally (tests/test_model/test_Unet_blocks.py:17, test_VIT_attention.py:92)."""
    import inspect
    import models.model_components as mc
    import models.model_internals as mi
    assert list(inspect.signature(mi.MP_Conv.__init__).parameters)[1:] == ["in_channels", "out_channels", "kernel", "stride"]
    assert list(inspect.signature(mi.MP_Attention.__init__).parameters)[1:] == [
        "num_heads", "emb_dim", "seq_ln", "time_dim", "context_dim", "attn_balance", "is_cross_attn"]
    assert list(inspect.signature(mc.Unet_block.__init__).parameters)[1:5] == ["in_channels", "out_channels", "kernel", "emb_size"]
    assert list(inspect.signature(mc.Router.forward).parameters)[1:] == ["x", "time_emb", "mask", "zeta"]
    blk = mc.Unet_block(4, 8, (3, 3), 10, Type="enc")
    assert blk.conv_skip is not None and blk.conv_res1.weights.shape == (8, 8, 3, 3)       # enc: conv_res1 in = out
    blk = mc.Unet_block(12, 8, (3, 3), 10, Type="dec")
    assert blk.conv_res1.weights.shape == (8, 12, 3, 3)                                     # dec: conv_res1 in = in
    vb = mc.Vit_block(2, 2, 8, 9, 8)
    assert vb.skip_proj is None and mc.Vit_block(2, 2, 6, 9, 8).skip_proj is not None
    at = mi.MP_Attention(2, 8, 9, is_cross_attn=True)
    assert at.rel_pos_bias is None and at.q_time is None and at.k_time is None


def test_checkpoint_dictionary_layout(tmp_path):
    """save_checkpoint writes the reference's dictionary (training.py:262-268) and the optimizer groups follow :55-60."""
    import torch
    from Utils import configs, training
    from models import model_config2
    over = dict(img_resolution=16, internal_channels=8, time_emb_dim=16, text_emb_dim=32, VIT_num_blocks=1, VIT_patch_sizes=[2, 4, 4, 8],
                VIT_num_groups=2, VIT_num_heads=2, VIT_emb_size=8, Unet_num_blocks=1, Unet_model_channels=8, log_var_channels=8)
    mcfg = dict(configs.model_configs, **over, save_dir=str(tmp_path))
    model = model_config2.preconditioned_HDMOEM(**configs.model_kwargs(mcfg))
    opt = training.build_optimizer(model, configs.optim_configs)
    assert [g["lr"] for g in opt.param_groups] == [configs.optim_configs[k] for k in ("lr_unet", "lr_vit", "lr_attn", "lr_router")]
    n_opt = sum(p.numel() for g in opt.param_groups for p in g["params"])
    n_exp = sum(p.numel() for m in (model.net.Unet_experts, model.net.VIT_experts, model.net.cross_attn, model.net.Unet_router,
                                    model.net.vit_router) for p in m.parameters())
    assert n_opt == n_exp
    path = training.save_checkpoint(model, opt, 7, 0.25, {"model_configs": mcfg}, "ckpt_7.pt")
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"step", "model_state_dict", "optimizer_state_dict", "mse", "config"} and ck["step"] == 7
    assert list(ck["model_state_dict"]) == list(model.state_dict())
    model2 = model_config2.preconditioned_HDMOEM(**configs.model_kwargs(mcfg))
    training.load_checkpoint(path, model2, training.build_optimizer(model2, configs.optim_configs))
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
