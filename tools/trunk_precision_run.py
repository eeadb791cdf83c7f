"""Evidence for the router-trunk backward precision default: the same 300-step training run at BASELINE configs[1]
(model_config1, 4x32x32 latents, 4 experts top-2, B = 256, bf16 experts, train mode, reference optimizer groups / LRs) twice from the
same seeds -- trunk backward with bf16 operands (HDMOE_TRUNK_BWD_BF16=1, one MFMA per product) vs the three-product fp32-equivalent
arithmetic -- logging loss, denoising loss, router entropy and expert usage.  Writes a JSON with both curves.

    python tools/trunk_precision_run.py profiles/r03_trunk_bwd_precision.json [steps]
"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd")
sys.path[:0] = [PKG, ROOT]
import torch
import hdmoe_hip
from hdmoe_hip import ops
from hdmoe_hip.dp import GradBuckets
from Utils import configs as C
from Utils import utils as U
from Utils import training as T
from models import model_config1

out_path = sys.argv[1] if len(sys.argv) > 1 else "trunk_bwd_precision.json"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda")
hdmoe_hip.set_compute_dtype(torch.bfloat16)
bc = C.BASELINE_CONFIGS[2]
kw = C.model_kwargs(**bc["over"])
B, E, K = bc["batch"], kw["num_experts"], kw["top_k"]


def run(bf16_bwd: bool):
    ops.TRUNK_BWD_BF16 = bf16_bwd
    torch.manual_seed(1234)
    hdmoe_hip.manual_seed(4321)
    model = model_config1.preconditioned_HDMOEM(**kw)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
            elif n.endswith("alpha_txt"):
                p.fill_(0.3)
    model = model.to(dev).train()
    opt = T.build_optimizer(model, C.optim_configs)
    lc = C.loss_configs
    crit = U.EDM_LOSS(num_experts=E, sigma_data=kw["sigma_data"], Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=lc["prior_bal"])
    buckets = GradBuckets(model)
    params = list(model.parameters())
    g = torch.Generator(device=dev).manual_seed(99)
    # a fixed synthetic "dataset" with structure to learn: 64 smooth prototype latents, each batch draws from them with small jitter
    protos = torch.nn.functional.interpolate(torch.randn(64, 4, 8, 8, device=dev, generator=g), size=32, mode="bilinear") * 0.7
    texts = torch.randn(64, 77, kw["text_emb_dim"], device=dev, generator=g)
    ones = torch.ones(B, E, device=dev)
    rec = []
    for step in range(STEPS):
        idx = torch.randint(0, 64, (B,), device=dev, generator=g)
        x0 = protos[idx] + 0.05 * torch.randn(B, 4, 32, 32, device=dev, generator=g)
        sigma = U.sample_sigma_hybrid(B, 0.002, 80.0, p_mean=-1.2, p_std=1.6, extreme_prob=0.5, device=dev, generator=g)
        x = x0 + sigma * torch.randn(B, 4, 32, 32, device=dev, generator=g)
        ops.advance_seed(dev)
        buckets.zero_grad()
        out = model(x=x, sigma=sigma, text_emb=texts[idx], Unet_router_mask=ones, Vit_router_mask=ones, zeta=0.1, return_log_var=True)
        loss = crit(sigma_vec=sigma, x=x0, sigma=sigma, out_model=out)
        loss["loss"].backward()
        buckets.finish()
        opt.step(clip=(params, 1.0))
        if step % 10 == 0 or step == STEPS - 1:
            with torch.no_grad():
                r = {"step": step, "loss": float(loss["loss"])}
                for k_ in ("denoising", "pure_loss", "balance", "z_loss"):
                    if k_ in loss:
                        r[k_] = float(loss[k_])
                for name, pk, rk in (("unet", "Unet_router_loss", "Unet_raw"), ("vit", "vit_router_loss", "vit_raw")):
                    p = out[pk].float().clamp_min(1e-12)
                    r[f"{name}_entropy"] = float(-(p * p.log()).sum(-1).mean())
                    top = torch.topk(out[rk].float(), K, dim=-1).indices
                    r[f"{name}_usage"] = [round(float((top == e).any(-1).float().mean()), 4) for e in range(E)]
                rec.append(r)
    return rec


res = {"config": "BASELINE configs[1], B=256, bf16 experts, train mode, FusedAdamW with the reference's 4 LR groups, clip 1.0, zeta 0.1, 64 prototype latents + noise",
       "steps": STEPS, "bf16_operands": run(True), "three_products": run(False)}
a, b = res["bf16_operands"], res["three_products"]
dl = [abs(x["loss"] - y["loss"]) / max(abs(y["loss"]), 1e-9) for x, y in zip(a, b)]
de = [abs(x["unet_entropy"] - y["unet_entropy"]) + abs(x["vit_entropy"] - y["vit_entropy"]) for x, y in zip(a, b)]
du = [max(abs(u - v) for u, v in zip(x["unet_usage"] + x["vit_usage"], y["unet_usage"] + y["vit_usage"])) for x, y in zip(a, b)]
res["summary"] = {"max_rel_loss_diff": max(dl), "mean_rel_loss_diff": sum(dl) / len(dl), "max_entropy_diff_sum": max(de), "max_usage_diff": max(du),
                  "final_loss": [a[-1]["loss"], b[-1]["loss"]], "final_unet_entropy": [a[-1]["unet_entropy"], b[-1]["unet_entropy"]],
                  "final_vit_entropy": [a[-1]["vit_entropy"], b[-1]["vit_entropy"]]}
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["summary"], indent=1))
for x, y in list(zip(a, b))[::3]:
    print(x["step"], round(x["loss"], 4), round(y["loss"], 4), round(x["unet_entropy"], 4), round(y["unet_entropy"], 4), x["unet_usage"], y["unet_usage"])
