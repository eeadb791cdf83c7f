"""Race detector for the staged replay when the host runs ahead of the device: the setup of tests/test_bench_path_parity.py (wide_config1, bf16, train
mode, p = 0 -- deterministic), `bursts` times `burst` replays back to back, after each burst the error of the router logits against the reference fixture
(8.5e-5 / 4.6e-5 when all is well).  This is how the transient under prioritised streams was found and localised (hdmoe_hip/graph.py StagedStep, with
HDMOE_STREAM_PRIO=1 and HDMOE_SKIP_STAGE=...).  usage: replay_race.py [eager tests first 0|1] [burst 3] [bursts 150]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_bench_path_parity as T  # noqa: E402
from hdmoe_hip import graph as hgraph  # noqa: E402
from hdmoe_hip.dp import GradBuckets  # noqa: E402
from Utils.utils import EDM_LOSS  # noqa: E402

g = torch.load(os.path.join(ROOT, "tests", "golden", "wide_config1.pt"), weights_only=False)
eager_first = len(sys.argv) > 1 and sys.argv[1] == "1"
if eager_first:
    for m in T.MODES:
        T.test_bank_path_third_step_matches_the_reference(g, *m)
model, kw, inp = T._setup(g, torch.bfloat16, train=True)
lc = g["loss_cfg"]
crit = EDM_LOSS(num_experts=kw["num_experts"], sigma_data=0.5, Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=0.0)
buckets = GradBuckets(model)
x = inp["x"].clone().requires_grad_(True)
keep = {}


def fwd_bwd():
    buckets.zero_grad()
    if x.grad is not None:
        x.grad.zero_()
    out = model(x=x, sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["unet_mask"], Vit_router_mask=inp["vit_mask"], zeta=0.0, return_log_var=True, **g["extra"])
    loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
    hgraph.backward(loss["loss"])
    keep["out"] = {k_: (None if v is None else v.detach()) for k_, v in out.items()}
    return loss["loss"].detach()


staged = hgraph.StagedStep(fwd_bwd, "cuda", warmup=2)
errs = []
burst = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
    for _ in range(burst):                                   # `burst` replays back to back (the host runs ahead), then one look at the logits
        staged()
    torch.cuda.synchronize()
    e = []
    for key in ("vit_raw", "Unet_raw"):
        got = keep["out"][key].detach().float().cpu()
        fin = torch.isfinite(g["out"][key])
        e.append(float((got[fin] - g["out"][key][fin]).abs().max()))
    errs.append(e)
flag = "OUTLIER" if max(e[0] for e in errs) > 5e-4 else "ok"
bad = [i for i, e in enumerate(errs) if e[0] > 5e-4 or e[1] > 5e-4]
print(flag, f"{len(bad)} outliers of {len(errs)} bursts at", bad[:20], " first bursts:", " ".join(f"{a:.1e}/{b:.1e}" for a, b in errs[:4]), flush=True)
