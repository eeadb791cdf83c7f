"""dev tool: is a sample's output independent of its batch (and run-to-run deterministic)?  usage: batch_indep.py [side=1] [dtype=bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd")]
import torch, hdmoe_hip
from hdmoe_hip import ops
from Utils import configs
from models import model_config1
from oracle.recipe import fill_state
args = dict(a.split("=") for a in sys.argv[1:])
ops.SIDE_STREAMS = args.get("side", "1") == "1"
dt = torch.bfloat16 if args.get("dtype", "bf16") == "bf16" else torch.float32
hdmoe_hip.set_compute_dtype(dt)
kw = configs.model_kwargs(**configs.BASELINE_CONFIGS[2]["over"])
model = model_config1.preconditioned_HDMOEM(**kw)
model.load_state_dict(fill_state(model.state_dict(), 77)); model = model.cuda().eval()
B, E, k = int(args.get("B", 256)), 4, 2
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, 4, 32, 32, device="cuda", generator=g)
sigma = torch.exp(torch.randn(B, 1, 1, 1, device="cuda", generator=g) * 1.6 - 1.2)
text = torch.randn(B, 77, 768, device="cuda", generator=g)
um = torch.ones(B, E, device="cuda"); vm = torch.ones(B, E, device="cuda")
def run(sl):
    with torch.no_grad():
        out = model(x=x[sl], sigma=sigma[sl], text_emb=text[sl], Unet_router_mask=um[sl], Vit_router_mask=vm[sl], zeta=0.0, return_log_var=True)
    torch.cuda.synchronize()
    return {k_: v.detach().float().clone() for k_, v in out.items() if v is not None}
run(slice(0, 8))
a = run(slice(0, B)); b = run(slice(0, B)); h1 = run(slice(0, B // 2)); h2 = run(slice(B // 2, B))
for key in a:
    cat = torch.cat([h1[key], h2[key]])
    fin = torch.isfinite(a[key]) & torch.isfinite(cat)
    d_rep = float((a[key] - b[key])[fin].abs().max()); d_half = float((a[key] - cat)[fin].abs().max())
    worst = (a[key] - cat).abs().flatten(1).amax(1) if a[key].ndim > 1 else (a[key] - cat).abs()
    print(f"{key:18s} repeat-diff {d_rep:.3e}  half-diff {d_half:.3e}  scale {float(a[key][fin].abs().max()):.3e}  samples differing: {int((worst > 0).sum())} first {worst.nonzero().flatten()[:6].tolist()}")
idx_a = torch.topk(a["Unet_raw"], k).indices; idx_h = torch.topk(torch.cat([h1["Unet_raw"], h2["Unet_raw"]]), k).indices
print("unet idx equal", bool((idx_a == idx_h).all()))
