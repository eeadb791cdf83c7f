"""Race detector for the staged replay: the wide_config1 fixture in train mode with dropout p = 0 and zeta = 0 is deterministic, so every replay
must reproduce the router logits, the output and the gradients of the replay before it (up to the ~1e-6 drift of the forced weight
re-normalisation).  A section that runs beside another one it should have waited for shows up as an outlier.  usage: replay_determinism.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    import hdmoe_hip
    from hdmoe_hip import graph as hgraph
    from hdmoe_hip.dp import GradBuckets
    from Utils.utils import EDM_LOSS
    from conftest import wide_setup
    g = torch.load(os.path.join(ROOT, "tests", "golden", "wide_config1.pt"), map_location="cpu", weights_only=False)
    hdmoe_hip.set_compute_dtype(torch.bfloat16)
    variant, model, kw, state, inp = wide_setup(g)
    model.load_state_dict(state)
    model = model.to("cuda").train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
            mod.dropout = 0.0
    inp = {k: v.to("cuda") for k, v in inp.items()}
    lc = g["loss_cfg"]
    crit = EDM_LOSS(num_experts=kw["num_experts"], sigma_data=0.5, Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=0.0)
    buckets = GradBuckets(model)
    x = inp["x"].clone().requires_grad_(True)
    keep = {}

    def fwd_bwd():
        buckets.zero_grad()
        if x.grad is not None:
            x.grad.zero_()
        out = model(x=x, sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["unet_mask"], Vit_router_mask=inp["vit_mask"],
                    zeta=0.0, return_log_var=True, **g["extra"])
        loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
        hgraph.backward(loss["loss"])
        keep["out"] = {k: (None if v is None else v.detach()) for k, v in out.items()}
        return loss["loss"].detach()

    staged = hgraph.StagedStep(fwd_bwd, "cuda", warmup=2)
    print("graphs:", sorted(staged.graphs), flush=True)
    names = [nm for nm, p in model.named_parameters()]
    pick = [nm for nm in names if ".vit_router." in nm or ".Unet_router." in nm][:12] + names[:4] + names[-4:]
    params = dict(model.named_parameters())

    def snap():
        o = keep["out"]
        d = {k: v.float().clone() for k, v in o.items() if torch.is_tensor(v)}
        d["x.grad"] = x.grad.float().clone()
        for nm in pick:
            if params[nm].grad is not None:
                d["grad:" + nm] = params[nm].grad.float().clone()
        return d

    for _ in range(4):
        staged()
    torch.cuda.synchronize()
    prev = snap()
    worst = {}
    bad = 0
    for it in range(n):
        staged()
        torch.cuda.synchronize()
        cur = snap()
        for k in cur:
            sc = float(prev[k].abs().max()) + 1e-30
            e = float((cur[k] - prev[k]).abs().max()) / sc
            worst[k] = max(worst.get(k, 0.0), e)
            if e > 1e-3:
                bad += 1
                print(f"replay {it}: {k} moved by {e:.3e} (relative to its max)", flush=True)
        prev = cur
    print("worst relative change between consecutive replays:")
    for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {v:.3e}  {k}")
    print("OUTLIERS:", bad)


if __name__ == "__main__":
    main()
