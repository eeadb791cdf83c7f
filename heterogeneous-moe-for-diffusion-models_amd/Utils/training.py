"""Training iteration and checkpoint format -- the optimizer side of the hot path.

Mirrors reference ``Utils/training.py``: optimizer groups (:55-60), cosine schedule (:62-65), the per-iteration order
forward -> loss -> zero_grad -> backward -> clip_grad_norm_(1.0) -> step -> scheduler.step (:125-197) and the checkpoint
dictionary (:242-271, read back at :303-304).  What the reference takes from the network (Flowers102, the SD VAE, CLIP;
``Utils/VAE_CLIP.py``) is outside this path: `train_steps` is fed latents and text embeddings by the caller.

Differences that are deliberate and visible:
  * gradient clipping + AdamW run as two multi-tensor HIP launches with no host sync (hdmoe_hip/optim.py);
  * the reference's router group reads ``model.net.routers`` (:59), an attribute that exists nowhere in the reference
    either; the two routers ``[net.Unet_router, net.vit_router]`` are used;
  * under torch.distributed the gradient all-reduce is the flat-bucket RCCL path of hdmoe_hip/dp.py.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, Iterable, Optional

import torch

from hdmoe_hip.dp import GradBuckets
from hdmoe_hip.optim import FusedAdamW
from .utils import EDM_LOSS, MaskGenerator, ZetaScheduler, sample_sigma_hybrid


def build_optimizer(model: torch.nn.Module, optim_config: Dict[str, Any]) -> FusedAdamW:
    """The reference's four AdamW groups (training.py:55-60): U-Net experts, ViT experts (boosted), fusion cross-attention,
    routers.  Like the reference, parameters outside these groups (stem, gates, text cross-attention, output conv,
    scaling_net, log-var head) are not optimised."""
    net = model.net
    routers = list(net.Unet_router.parameters()) + list(net.vit_router.parameters())
    opt = FusedAdamW([
        {"params": list(net.Unet_experts.parameters()), "lr": optim_config["lr_unet"]},
        {"params": list(net.VIT_experts.parameters()), "lr": optim_config["lr_vit"]},
        {"params": list(net.cross_attn.parameters()), "lr": optim_config["lr_attn"]},
        {"params": routers, "lr": optim_config["lr_router"]},
    ])
    # an expert without a sample in a step is left out of that step's update, as in the reference (its .grad stays None there:
    # models/model_config1.py:26-29, and torch.optim.AdamW skips grad-None tensors)
    opt.track_expert_usage([net.Unet_experts, net.VIT_experts])
    return opt


def build_scheduler(optimizer: torch.optim.Optimizer, optim_config: Dict[str, Any]):
    return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=optimizer, T_max=optim_config["total_schedule_steps"],
                                                      eta_min=optim_config["eta_min"])


def save_checkpoint(model, optimizer, step, mse_score, configs, filename) -> str:
    """Same dictionary and path rules as reference training.py:242-271 (keys step / model_state_dict /
    optimizer_state_dict / mse / config); tensors are written from host copies so the file loads on any device."""
    if "save_dir" in configs:
        save_path = configs["save_dir"]
    elif "model_configs" in configs and "save_dir" in configs["model_configs"]:
        save_path = configs["model_configs"]["save_dir"]
    else:
        save_path = "./checkpoints"
    os.makedirs(save_path, exist_ok=True)
    full_path = os.path.join(save_path, filename)
    model_state = model.module.state_dict() if hasattr(model, "module") else model.state_dict()
    checkpoint = {"step": step, "model_state_dict": model_state, "optimizer_state_dict": optimizer.state_dict(),
                  "mse": mse_score, "config": configs}
    torch.save(checkpoint, str(full_path))
    print(f"   [Save] Checkpoint saved: {full_path}")
    return full_path


def load_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None, map_location=None) -> dict:
    """Inverse of `save_checkpoint` (reference training.py:303-304 loads only the model; resuming also needs the
    optimizer moments).  Accepts files written by the reference's torch.optim.AdamW as well."""
    ck = torch.load(f=path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    return ck


class Trainer:
    """One training iteration of reference training.py:110-197 as a reusable object (device-side sigma / mask / noise
    generation, fused loss, flat-bucket gradient all-reduce, fused clip + AdamW)."""

    def __init__(self, model, model_config, optim_config, loss_config, mask_config, zeta_config, max_grad_norm: float = 1.0,
                 fuse_clip_into_step: bool = True, logger=None):
        self.model, self.cfg, self.mask_cfg = model, model_config, mask_config
        self.optimizer = build_optimizer(model, optim_config)
        self.scheduler = build_scheduler(self.optimizer, optim_config)
        self.zeta_sched = ZetaScheduler(total_steps=zeta_config["total_schedule_steps"], max_zeta=zeta_config["max_zeta"],
                                        min_zeta=zeta_config["min_zeta"], strategy=zeta_config["strategy"],
                                        warmup_ratio=zeta_config["warmup_ratio"])
        mk = lambda attr, rng: MaskGenerator(expert_attributes=mask_config[attr], p_mean=mask_config["p_mean"], p_std=mask_config["p_std"],
                                             total_steps=model_config["total_steps"], min_active=mask_config["min_active"],
                                             step_size=mask_config["step_size"], max_bandwidth=mask_config["max_BW"],
                                             bandwidth=mask_config["BW"], strat_band=mask_config["strat_band"], noise_range=mask_config[rng])
        self.unet_mask_gen, self.vit_mask_gen = mk("unet_attr", "unet_noise_range"), mk("vit_attr", "vit_noise_range")
        self.criterion = EDM_LOSS(num_experts=model_config["num_experts"], sigma_data=model_config["sigma_data"],
                                  Unet_bal=loss_config["unet_bal"], vit_bal=loss_config["vit_bal"], z_bal=loss_config["z_bal"],
                                  prior_bal=loss_config["prior_bal"])
        self.buckets = GradBuckets(model)                    # .grad become views of flat fp32 buckets (all-reduced when world > 1)
        self.max_grad_norm, self.fuse = float(max_grad_norm), fuse_clip_into_step
        self.logger = logger                                 # graphs.logger.Logger (sync-free) or None
        self._clip_params = [p for p in model.parameters()]
        self.step_idx = 0

    def train_step(self, latent_images: torch.Tensor, text_emb: torch.Tensor) -> dict:
        cfg, mc, step = self.cfg, self.mask_cfg, self.step_idx
        dev = latent_images.device
        sigma = sample_sigma_hybrid(batch_size=latent_images.shape[0], sigma_max=cfg["sigma_max"], sigma_min=cfg["sigma_min"],
                                    p_mean=mc["p_mean"], p_std=mc["p_std"], extreme_prob=0.5, device=dev)
        images_noised = latent_images + torch.randn_like(latent_images) * sigma
        out_model = self.model(x=images_noised, sigma=sigma, text_emb=text_emb, Unet_router_mask=self.unet_mask_gen(sigma=sigma, step=step),
                               Vit_router_mask=self.vit_mask_gen(sigma=sigma, step=step), zeta=self.zeta_sched.get_zeta(step=step),
                               transition_point=mc["p_mean"], softness=mc["p_std"], return_log_var=True)
        loss = self.criterion(sigma_vec=sigma, x=latent_images, sigma=sigma, out_model=out_model)
        lg = self.logger
        if lg is not None:                                   # same calls, same order as reference training.py:160-188
            lg.log_training_step(step=step, loss_dict=loss, zeta=self.zeta_sched.get_zeta(step=step),
                                 log_var=out_model["log_var"] if out_model["log_var"] is not None else 0.0,
                                 lr=self.optimizer.param_groups[0]["lr"], sigma=sigma, p_mean=mc["p_mean"], p_std=mc["p_std"])
            lg.log_router_statistics(step=step, unet_probs=out_model["Unet_router_loss"], vit_probs=out_model["vit_router_loss"],
                                     sigma=sigma, p_mean=mc["p_mean"], p_std=mc["p_std"])
            lg.log_scaling_gating(scaling_factors=out_model["scaling_net_out"], gate_weights=out_model["out_gate"], sigma=sigma)
        self.buckets.zero_grad()
        loss["loss"].backward()
        self.buckets.finish()
        if lg is not None:
            lg.log_gradients(step=step, model=self.model.net)
            lg.log_weight_statistics(step=step, model=self.model.net)
        from hdmoe_hip.optim import clip_grad_norm_
        if self.fuse:
            self.optimizer.step(clip=(self._clip_params, self.max_grad_norm))
        else:
            clip_grad_norm_(self._clip_params, self.max_grad_norm)
            self.optimizer.step()
        self.scheduler.step()
        self.step_idx += 1
        return {"loss": loss, "out_model": out_model, "sigma": sigma}


def train_steps(trainer: Trainer, batches: Iterable, on_step: Optional[Callable[[int, dict], None]] = None) -> None:
    """Drive `trainer` over (latents, text_emb) batches; `on_step(step, result)` is where logging / checkpoints hook in."""
    trainer.model.train()
    for latents, text in batches:
        res = trainer.train_step(latents, text)
        if on_step is not None:
            on_step(trainer.step_idx - 1, res)
