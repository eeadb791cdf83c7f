"""Training-side neighbours of the hot path -- drop-in for the names the reference's ``Utils/utils.py`` exports to
``Utils/training.py`` (``EDM_LOSS``, ``sample_sigma_hybrid``, ``ZetaScheduler``, ``MaskGenerator``).

``EDM_LOSS`` runs as one fused HIP forward + one fused backward with no ``.item()`` host syncs.  The input generators
produce a few (B,)/(B,E) values per step from torch's RNG before the timed path starts; they are kept as plain
torch device ops (data generation, not the denoiser's arithmetic).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from hdmoe_hip import ops


class EDM_LOSS(nn.Module):
    """log-var-weighted MSE + router load-balance + z-loss, all clamped at 50 (reference Utils/utils.py:105-172)."""

    def __init__(self, num_experts: int, sigma_data: float = 0.5, Unet_bal: float = 0.0005, vit_bal: float = 0.0005,
                 z_bal: float = 0.0001, prior_bal: float = 0.001, transition_sigma: float = 1.0, sharpness: float = 2.0):
        super().__init__()
        self.num_experts = num_experts
        self.sigma_data = sigma_data
        self.Unet_lambda = Unet_bal
        self.vit_lambda = vit_bal
        self.z_bal = z_bal
        self.prior_bal = prior_bal      # the reference computes no prior term (commented out at utils.py:143,145)

    def __call__(self, sigma_vec: torch.Tensor, x: torch.Tensor, sigma: torch.Tensor, out_model: dict) -> dict:
        loss, st = ops.edm_loss(out_model["denoised"], x, out_model["log_var"], out_model["Unet_router_loss"],
                                out_model["vit_router_loss"], out_model["Unet_raw"], out_model["vit_raw"],
                                self.Unet_lambda, self.vit_lambda, self.z_bal)
        return {"loss": loss, "denoising": st[1], "balance": st[2], "z_loss": st[3], "entropy": 0.0, "pure_loss": st[4]}


def sample_sigma_hybrid(batch_size, sigma_min=0.002, sigma_max=80.0, p_mean=-0.4, p_std=1.0, extreme_prob=0.2, device="cuda",
                        generator: Optional[torch.Generator] = None):
    """Log-normal core + log-uniform tail, shuffled (reference Utils/utils.py:26-61)."""
    n_ln = int(batch_size * (1 - extreme_prob))
    ln = (torch.randn([n_ln, 1, 1, 1], device=device, generator=generator) * p_std + p_mean).exp()
    u = torch.rand([batch_size - n_ln, 1, 1, 1], device=device, generator=generator)
    lu = (u * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min)).exp()
    sigma = torch.cat([ln, lu], dim=0).clamp(sigma_min, sigma_max)
    return sigma[torch.randperm(batch_size, device=device, generator=generator)]


class ZetaScheduler:
    """Exploration-noise schedule (reference Utils/utils.py:175-225); pure host arithmetic."""

    def __init__(self, total_steps: int, max_zeta: float, min_zeta: float = 0.0, strategy: str = "cos", alpha: float = 4.0,
                 warmup_ratio: float = 0.05):
        self.total_steps, self.max_zeta, self.min_zeta = total_steps, max_zeta, min_zeta
        self.strategy, self.alpha = strategy, alpha
        self.warmup_steps = int(total_steps * warmup_ratio)

    def get_zeta(self, step: int) -> float:
        if step < self.warmup_steps:
            return self.max_zeta
        if step >= self.total_steps:
            return self.min_zeta
        cur, tot = step - self.warmup_steps, self.total_steps - self.warmup_steps
        if self.strategy == "cos":
            return float(self.min_zeta + (self.max_zeta - self.min_zeta) * 0.5 * (1 + np.cos(np.pi * cur / tot)))
        if self.strategy == "exp":
            term = max(min(-self.alpha * (cur - (self.max_zeta / tot)), 10), -10)
            z = (self.max_zeta - self.min_zeta) * np.exp(term) + self.min_zeta
            return float(max(min(z, self.max_zeta), self.min_zeta))
        raise ValueError(f"Unknown strategy: {self.strategy}")


class MaskGenerator(nn.Module):
    """Rank-based noise-band expert masks (reference Utils/utils.py:228-330)."""

    def __init__(self, expert_attributes: list, p_mean: float = -0.4, p_std: float = 1.0, bandwidth: float = 0.3,
                 max_bandwidth: float = 0.9, min_active: int = 1, total_steps: int = 5000, step_size: float = 0.1,
                 noise_range: tuple = (0.0, 1.0), strat_band: str = "step"):
        super().__init__()
        self.strat_band, self.total_steps, self.max_bw, self.step_size = strat_band, total_steps, max_bandwidth, step_size
        self.p_mean, self.p_std, self.bandwidth, self.min_active = p_mean, p_std, bandwidth, min_active
        attrs = torch.tensor(expert_attributes, dtype=torch.float32)
        order = torch.sort(attrs, stable=True).indices
        centers = torch.zeros_like(attrs)
        centers[order] = torch.linspace(noise_range[0], noise_range[1], steps=len(attrs))
        self.register_buffer("expert_centers", centers)

    @torch.no_grad()
    def __call__(self, sigma: torch.Tensor, step: int) -> torch.Tensor:
        s = sigma.flatten()
        pct = (0.5 * (1 + torch.erf((torch.log(s) - self.p_mean) / (self.p_std * np.sqrt(2))))).clamp(0, 1)
        dist = torch.abs(pct.view(-1, 1) - self.expert_centers.to(s.device).view(1, -1))
        mask = (dist <= self.bandwidth_scheduler(step)).float()
        mask.scatter_(1, torch.topk(-dist, k=self.min_active, dim=-1).indices, 1.0)
        return mask

    def bandwidth_scheduler(self, step: int) -> float:
        if step >= self.total_steps:
            return self.max_bw
        if self.strat_band == "linear":
            return self.bandwidth + (self.max_bw - self.bandwidth) * step / float(self.total_steps)
        if self.strat_band == "step":
            progress = min(int(step / (self.total_steps * self.step_size)) / int(1.0 / self.step_size), 1.0)
            return self.bandwidth + (self.max_bw - self.bandwidth) * progress
        raise ValueError(f"Unknown bandwidth strategy: {self.strat_band}")
