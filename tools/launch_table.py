"""Dev tool: per-launch timing of the conv-family kernels for one bench step (shapes + achieved TFLOP/s)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import torch
from hdmoe_hip import ops
import hdmoe_hip, configs as C, utils as U

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
model, kw, bc = bench.build_model(cfg, dev)
B = bc["batch"]
inp = bench.make_inputs(kw, B, dev, 1234, bc["module"])
lc = C.loss_configs
crit = U.EDM_LOSS(num_experts=kw["num_experts"], Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"])

def step():
    model.zero_grad(set_to_none=True)
    out = model(x=inp["x"], sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["um"], Vit_router_mask=inp["vm"],
                zeta=0.1, return_log_var=True, **inp["extra"])
    crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)["loss"].backward()

for _ in range(2):
    step()
torch.cuda.synchronize()
ops.PROFILE = []
step()
torch.cuda.synchronize()
rec, ops.PROFILE = ops.PROFILE, None
rows = []
for kind, info, s, e in rec:
    ms = s.elapsed_time(e)
    fl = bench.conv_flops(info)
    rows.append((ms, kind, info["dtype"], info["N"], info["HW"], info["O"], info["I"], info["taps"], info["seg"] is not None, fl / (ms * 1e-3) / 1e12))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"total conv-family ms: {tot:.2f} over {len(rows)} launches")
agg = {}
for r in rows:
    key = r[1:9]
    a = agg.setdefault(str(key), [0.0, 0, 0.0])
    a[0] += r[0]; a[1] += 1; a[2] = r[9]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0]:8.3f} ms  n={v[1]:3d}  TF={v[2]:7.1f}  {k}")
