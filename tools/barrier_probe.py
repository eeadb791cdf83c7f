"""Diagnostic: does a collective between two replays of the staged step change what the replays compute?  (one-rank RCCL group, eval mode:
every replay is deterministic, so the stage-boundary tensors of two replays can be compared bit for bit)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd")
sys.path[:0] = [PKG, os.path.join(PKG, "Utils"), ROOT]
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "barrier"
if mode != "nodist":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.cuda.memory._record_memory_history(enabled="all", context="alloc", stacks="python", max_entries=3000000)
import hdmoe_hip
from hdmoe_hip import graph as hgraph
import bench as B
hdmoe_hip.lib()
hdmoe_hip.manual_seed(1)
train = "train" in sys.argv
model, kw, bc = B.build_model(2, dev)
if not train:
    model.eval()
Bn = 256
inp = B.make_inputs(kw, Bn, dev, 7, bc["module"])
import configs as C, utils as U
lc = C.loss_configs
crit = U.EDM_LOSS(num_experts=kw["num_experts"], sigma_data=kw["sigma_data"], Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=lc["prior_bal"])

def fwd_bwd():
    model.zero_grad(set_to_none=False)
    out = model(x=inp["x"], sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["um"], Vit_router_mask=inp["vm"], zeta=0.0 if not train else 0.1,
                return_log_var=True, **inp["extra"])
    loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
    hgraph.backward(loss["loss"])
    return loss["loss"].detach()

st = hgraph.StagedStep(fwd_bwd, dev)
cuts = [(prod, i, t, d) for prod, lst in st._keep.cuts.items() for i, (t, d) in enumerate(lst)]
def snap():
    torch.cuda.synchronize()
    s = {"loss": st.out.clone()}
    for prod, i, t, d in cuts:
        s[f"{prod}[{i}] fwd {tuple(t.shape)}"] = t.detach().clone()
        if d.grad is not None:
            s[f"{prod}[{i}] grad"] = d.grad.clone()
    for n, p in model.named_parameters():
        if p.grad is not None:
            s["g:" + n] = p.grad.clone()
    return s
st(); st()
if mode == "alloc2":
    torch.cuda.synchronize()
    print("loss", float(st.out))
    snap_m = torch.cuda.memory._snapshot()
    events = [e for tr in snap_m["device_traces"] for e in tr]
    for k in range(3):
        t = torch.zeros(1, device=dev); torch.cuda.synchronize()
        ptr = t.data_ptr()
        prev = [e for e in events if e.get("action") == "alloc" and e["addr"] <= ptr < e["addr"] + e["size"]]
        print(f"zeros(1) landed at {ptr:#x}; {len(prev)} earlier allocations there; the last two:")
        for e in prev[-2:]:
            fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in e.get("frames", []) if "/torch/" not in f["filename"]][:14]
            print(f"   {e['size']} B stream {e.get('stream')} <- " + " <- ".join(fr))
        st(); torch.cuda.synchronize()
        print("loss after", float(st.out), flush=True)
        keep = t if k == 0 else None
    sys.exit(0)
ref = snap()
print("loss", float(ref["loss"]), flush=True)
st()
again = snap()
bad = [k for k in ref if not torch.equal(ref[k], again[k])]
print("replay vs replay (no collective between): differing tensors:", len(bad), bad[:6], flush=True)
if mode == "barrier":
    dist.barrier()
elif mode == "allreduce":
    t = torch.ones(1000, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
elif mode == "alloc":
    snap_m = torch.cuda.memory._snapshot()
    events = [e for tr in snap_m["device_traces"] for e in tr]
    t = torch.zeros(1, device=dev); torch.cuda.synchronize()
    ptr = t.data_ptr()
    prev = [e for e in events if e.get("action") == "alloc" and e["addr"] <= ptr < e["addr"] + e["size"]]
    print(f"zeros(1) landed at {ptr:#x}; {len(prev)} earlier allocations there; the last three:")
    for e in prev[-3:]:
        fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in e.get("frames", []) if "/torch/" not in f["filename"]][:10]
        print(f"   {e['size']} B stream {e.get('stream')} <- " + " <- ".join(fr))
elif mode == "sidekernel":
    s2 = torch.cuda.Stream(); 
    with torch.cuda.stream(s2):
        t = torch.ones(1 << 20, device=dev) * 3
    torch.cuda.synchronize()
st()
after = snap()
print("loss after", float(after["loss"]), flush=True)
bad = [k for k in ref if not torch.equal(ref[k], after[k])]
print(f"after {mode}: differing tensors: {len(bad)} of {len(ref)}")
for k in bad[:25]:
    a, b = ref[k].float(), after[k].float()
    print("   ", k, "max|d|", float((a - b).abs().max()), "ref max", float(a.abs().max()), "finite", bool(torch.isfinite(b).all()))
if mode != "nodist":
    dist.destroy_process_group()
