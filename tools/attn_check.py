"""Dev tool: error of the bf16 attention kernels (fwd + grads) against an fp64 torch reference on the same bf16 inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"))
import torch
from hdmoe_hip import ops
torch.manual_seed(0)
for (B, Sq, Skv, H, D) in [(4, 1024, 1024, 8, 4), (3, 300, 77, 8, 4), (2, 33, 130, 2, 4)]:
    dt = torch.bfloat16
    q = torch.randn(B, Sq, H * D, device="cuda").to(dt).requires_grad_(True)
    k = torch.randn(B, Skv, H * D, device="cuda").to(dt).requires_grad_(True)
    v = torch.randn(B, Skv, H * D, device="cuda").to(dt).requires_grad_(True)
    go = torch.randn(B, Sq, H * D, device="cuda").to(dt)
    o = ops.attention(q, k, v, None, H); o.backward(go)
    q6, k6, v6 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    qh = q6.view(B, Sq, H, D).transpose(1, 2); kh = k6.view(B, Skv, H, D).transpose(1, 2); vh = v6.view(B, Skv, H, D).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / D ** 0.5, dim=-1)
    ref = (p @ vh).transpose(1, 2).reshape(B, Sq, H * D)
    ref.backward(go.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print(f"B{B} Sq{Sq} Skv{Skv} H{H}: out {rel(o, ref.detach()):.2e} dq {rel(q.grad, q6.grad):.2e} dk {rel(k.grad, k6.grad):.2e} dv {rel(v.grad, v6.grad):.2e}")
