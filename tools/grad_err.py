"""Diagnostic: relative error (max|d| / max|ref|) of every stored tensor of the wide fixtures, fp32 compute."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd")]
from conftest import wide_setup
import hdmoe_hip
from Utils.utils import EDM_LOSS
hdmoe_hip.set_compute_dtype(torch.float32)
for cid in (2, 3, 4):
    g = torch.load(os.path.join(ROOT, f"tests/golden/wide_config{cid}.pt"), weights_only=False)
    variant, model, kw, state, inp = wide_setup(g)
    model.load_state_dict(state); model = model.cuda().eval()
    d = lambda t: t.cuda()
    for it in range(2):
        model.zero_grad()
        x = d(inp["x"]).requires_grad_(True)
        out = model(x=x, sigma=d(inp["sigma"]), text_emb=d(inp["text"]), Unet_router_mask=d(inp["unet_mask"]), Vit_router_mask=d(inp["vit_mask"]),
                    zeta=0.0, return_log_var=True, **g["extra"])
        lc = g["loss_cfg"]
        crit = EDM_LOSS(num_experts=kw["num_experts"], sigma_data=0.5, Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=0.0)
        loss = crit(sigma_vec=d(inp["sigma"]), x=d(inp["x0"]), sigma=d(inp["sigma"]), out_model=out)
        loss["loss"].backward()
        rel = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max() / b.double().abs().max())
        print(f"--- config {cid} iter {it} (bank {'on' if it else 'off'})")
        for k, v in g["out"].items():
            m = torch.isfinite(v)
            print(f"  out {k:20s} {float((out[k].detach().cpu()[m]-v[m]).abs().max()/v[m].abs().max()):.2e}")
        print(f"  x_grad {rel(x.grad, g['x_grad']):.2e}")
        pg = dict(model.named_parameters())
        for n, r in g["param_grads"].items():
            print(f"  {n:66s} {rel(pg[n].grad, r):.2e}   scale {float(r.abs().max()):.2e}")
