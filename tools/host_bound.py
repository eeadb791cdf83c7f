"""Dev tool: is the training step host-bound?  Compares host issue time per step with GPU time per step."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
import hdmoe_hip, configs as C, utils as U
from hdmoe_hip.dp import GradBuckets
dev = torch.device("cuda", 0)
model, kw, bc = bench.build_model(2, dev)
B = bc["batch"]
inp = bench.make_inputs(kw, B, dev, 1234, bc["module"])
lc = C.loss_configs
crit = U.EDM_LOSS(num_experts=kw["num_experts"], Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"])
buckets = GradBuckets(model)
def step():
    buckets.zero_grad()
    out = model(x=inp["x"], sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["um"], Vit_router_mask=inp["vm"], zeta=0.1, return_log_var=True)
    crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)["loss"].backward()
    buckets.finish()
for _ in range(5): step()
torch.cuda.synchronize(); gc.collect(); gc.disable()
n = 10
issue = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); step(); issue += time.perf_counter() - a
t_issue_all = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {1e3*issue/n:.1f} ms/step ; wall {1e3*t_all/n:.1f} ms/step ; GPU drain after last issue {1e3*(t_all-t_issue_all):.1f} ms")
