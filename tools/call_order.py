"""dev tool: the order in which one eager training step issues its kernels, and on which torch stream (run-length compressed).
usage (GPU box): python tools/call_order.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
from hdmoe_hip import _lib, ops
import configs as C, utils as U
dev = torch.device("cuda:0")
model, kw, bc = bench.build_model(2, dev)
inp = bench.make_inputs(kw, 64, dev, 1234, bc["module"])
lc = C.loss_configs
crit = U.EDM_LOSS(num_experts=kw["num_experts"], sigma_data=kw["sigma_data"], Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=lc["prior_bal"])
def step():
    ops.advance_seed(dev)
    model.zero_grad(set_to_none=False)
    out = model(x=inp["x"], sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["um"], Vit_router_mask=inp["vm"], zeta=0.1, return_log_var=True, **inp["extra"])
    loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
    mark = len(_lib.CALL_LOG) if _lib.CALL_LOG is not None else 0
    loss["loss"].backward()
    return mark
step(); step()
_lib.CALL_LOG = []
mark = step()
torch.cuda.synchronize()
log = _lib.CALL_LOG
_lib.CALL_LOG = None
print(f"{len(log)} calls; backward starts at call {mark}")
i = 0
while i < len(log):
    j = i
    while j < len(log) and log[j][1] == log[i][1]:
        j += 1
    names = [n.replace("hdmoe_", "") for n, _ in log[i:j]]
    print(f"[{i:4d}..{j - 1:4d}] stream {log[i][1]:3d} n={j - i:3d} {'BWD' if i >= mark else 'fwd'}  {' '.join(names[:6])} ... {' '.join(names[-2:])}")
    i = j
