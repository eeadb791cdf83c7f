"""Hyper-parameter dictionaries -- restates the reference's ``Utils/configs.py`` (the one shipped config: 4 experts,
internal_channels 32, R = 32) and adds the builder-defined 8-expert variants that BASELINE.json's configs 3-5 name but
the reference never defines."""
import torch

model_configs = {
    "device": "cuda" if torch.cuda.is_available() else "cpu",
    "img_channels": 4, "internal_channels": 32, "data_img_res": 256, "img_resolution": 32, "time_emb_dim": 64,
    "text_emb_dim": 768, "num_experts": 4, "top_k": 1, "fourier_bandwidth": 1.0, "VIT_num_blocks": 4,
    "VIT_patch_sizes": [4, 8, 8, 16], "VIT_num_groups": 4, "VIT_num_heads": 8, "VIT_emb_size": 32, "Unet_num_blocks": 2,
    "Unet_channel_mult": [1, 2], "Unet_kernel_sizes": [(3, 3), (3, 3), (5, 5), (5, 5)], "Unet_model_channels": 32,
    "Unet_channel_mult_emb": 2, "Unet_label_balance": 0.5, "Unet_concat_balance": 0.5, "sigma_data": 0.5,
    "log_var_channels": 32, "batch_size": 32, "total_steps": 5000, "sigma_min": 0.002, "sigma_max": 80,
}
loss_configs = {"unet_bal": 0.05, "vit_bal": 0.1, "z_bal": 0.005, "prior_bal": 0.0}
optim_configs = {"eta_min": 1e-5, "lr_vit": 2e-3, "lr_unet": 5e-4, "lr_attn": 1e-3, "lr_router": 5e-4, "total_schedule_steps": 5000}
mask_configs = {"unet_attr": [3, 3, 5, 5], "vit_attr": [4, 8, 8, 16], "p_mean": -1.2, "p_std": 1.6, "BW": 0.3, "max_BW": 0.8,
                "min_active": 1, "step_size": 0.1, "strat_band": "step", "unet_noise_range": (0.0, 0.6),
                "vit_noise_range": (0.4, 1.0)}
zeta_configs = {"min_zeta": 0.01, "max_zeta": 2, "warmup_ratio": 0.05, "strategy": "cos", "alpha": 4.0, "total_schedule_steps": 900}


def model_kwargs(cfg: dict = None, **over) -> dict:
    """Constructor kwargs of preconditioned_HDMOEM from a config dict (reference Utils/training.py:32-53)."""
    c = dict(model_configs if cfg is None else cfg)
    c.update(over)
    return dict(IN_in_channels=c["img_channels"], IN_img_resolution=c["img_resolution"], internal_channels=c["internal_channels"],
                time_emb_dim=c["time_emb_dim"], text_emb_dim=c["text_emb_dim"], num_experts=c["num_experts"], top_k=c["top_k"],
                Fourier_bandwidth=c["fourier_bandwidth"], VIT_num_blocks=c["VIT_num_blocks"], VIT_patch_sizes=c["VIT_patch_sizes"],
                VIT_num_groups=c["VIT_num_groups"], VIT_num_heads=c["VIT_num_heads"], VIT_emb_size=c["VIT_emb_size"],
                Unet_num_blocks=c["Unet_num_blocks"], Unet_channel_mult=c["Unet_channel_mult"],
                Unet_channel_mult_emb=c["Unet_channel_mult_emb"], Unet_kernel_sizes=c["Unet_kernel_sizes"],
                Unet_model_channels=c["Unet_model_channels"], sigma_data=c["sigma_data"], log_var_channels=c["log_var_channels"])


# BASELINE.json configs.  "module": 1 -> models.model_config1, 2 -> models.model_config2
BASELINE_CONFIGS = {
    1: dict(module=1, batch=8, dtype="fp32", over=dict(img_channels=3, top_k=1)),
    2: dict(module=1, batch=256, dtype="bf16", over=dict(top_k=2)),
    3: dict(module=2, batch=256, dtype="bf16", over=dict(
        num_experts=8, top_k=2, Unet_kernel_sizes=[(3, 3)] * 3 + [(5, 5)] * 3 + [(7, 7)] * 2,
        VIT_patch_sizes=[4, 4, 8, 8, 8, 16, 16, 16])),
    4: dict(module=2, batch=32, dtype="bf16", over=dict(
        img_resolution=64, num_experts=8, top_k=2, Unet_kernel_sizes=[(3, 3)] * 3 + [(5, 5)] * 3 + [(7, 7)] * 2,
        VIT_patch_sizes=[4, 4, 8, 8, 8, 16, 16, 16])),
}
