"""HDMOEM with closed-form sigmoid path scaling -- drop-in for the reference's ``models/model_config2.py``
(what Utils/training.py and Utils/EDM_sampler.py import)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from hdmoe_hip import bank as wbank
from hdmoe_hip import ops
from models._assembly import _HDMOEMBase, _PrecondBase, router_to_unet_experts   # noqa: F401  (re-exported helper)

Tensor = torch.Tensor


class HDMOEM(_HDMOEMBase):
    """Reference model_config2.py:42-303: scaling_vit = 2*(sigmoid((4*c_noise - transition_point)/softness) + 0.01),
    scaling_unet = 2*(1 - sigmoid(.) + 0.01); query = U-Net features, context = ViT features."""

    _has_scaling_net = False

    def _scaling(self, time_vec, time_embed, zeta, transition_point, softness):
        return ops.sigmoid_scaling(time_vec, transition_point, softness)

    def _fusion_inputs(self, fu, fv, s_vit, s_unet, **kw):
        return fu, fv

    def forward(self, x: Tensor, time_vec: Tensor, text_emb: Tensor, Unet_router_mask: Tensor, Vit_router_mask: Tensor,
                zeta: float, transition_point: float, softness: float):
        wbank.bank_for(self).begin_step(self.training)
        res = self._fwd(ops.to_nhwc(ops.cast(x, torch.float32)), time_vec, text_emb, Unet_router_mask, Vit_router_mask, zeta,
                        transition_point=transition_point, softness=softness)
        wbank.deactivate()
        return self._public(res)


class preconditioned_HDMOEM(_PrecondBase):
    """Reference model_config2.py:306-468.  Returns the reference's dict of 8 entries."""

    _net_cls = HDMOEM

    def __init__(self, IN_in_channels: int, IN_img_resolution: int, internal_channels: int, time_emb_dim: int,
                 text_emb_dim: int, num_experts: int, top_k: int, Fourier_bandwidth: float, VIT_num_blocks: int,
                 VIT_patch_sizes: List[int], VIT_num_groups: int, VIT_num_heads: int, VIT_emb_size: int, Unet_num_blocks: int,
                 Unet_channel_mult: list, Unet_kernel_sizes: List[Tuple[int, int]], Unet_model_channels: Optional[int] = 192,
                 Unet_channel_mult_emb: Optional[int] = None, Unet_label_balance: Optional[float] = 0.5,
                 Unet_concat_balance: Optional[float] = 0.5, sigma_data: Optional[float] = 0.5,
                 log_var_channels: Optional[int] = 128):
        super().__init__(sigma_data=sigma_data, log_var_channels=log_var_channels, IN_in_channels=IN_in_channels,
                         IN_img_resolution=IN_img_resolution, internal_channels=internal_channels, time_emb_dim=time_emb_dim,
                         text_emb_dim=text_emb_dim, num_experts=num_experts, top_k=top_k, Fourier_bandwidth=Fourier_bandwidth,
                         VIT_num_blocks=VIT_num_blocks, VIT_patch_sizes=VIT_patch_sizes, VIT_num_groups=VIT_num_groups,
                         VIT_num_heads=VIT_num_heads, VIT_emb_size=VIT_emb_size, Unet_num_blocks=Unet_num_blocks,
                         Unet_channel_mult=Unet_channel_mult, Unet_kernel_sizes=Unet_kernel_sizes,
                         Unet_model_channels=Unet_model_channels, Unet_channel_mult_emb=Unet_channel_mult_emb,
                         Unet_label_balance=Unet_label_balance, Unet_concat_balance=Unet_concat_balance)

    def forward(self, x: Tensor, sigma: Tensor, text_emb: Tensor, Unet_router_mask: Tensor, Vit_router_mask: Tensor, zeta: float,
                transition_point: float, softness: float, return_log_var: bool = False):
        return self._forward(x, sigma, text_emb, Unet_router_mask, Vit_router_mask, zeta, return_log_var,
                             transition_point=transition_point, softness=softness)
