"""Dev tool: micro-benchmark of the grouped conv kernels on one layer shape.
usage: python tools/conv_bench.py [dtype=bf16] [rows=512] [hw=32] [cin=64] [cout=64] [ks=3,3,5,5] [iters=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"))
import torch
from hdmoe_hip import ops

args = dict(a.split("=") for a in sys.argv[1:])
dt = torch.bfloat16 if args.get("dtype", "bf16") == "bf16" else torch.float32
R, HW, CI, CO = int(args.get("rows", 512)), int(args.get("hw", 32)), int(args.get("cin", 64)), int(args.get("cout", 64))
ks = [int(k) for k in args.get("ks", "3,3,5,5").split(",")]
iters = int(args.get("iters", 20))
dev = "cuda"
E = len(ks)
x = torch.randn(R, HW, HW, CI, device=dev).to(dt).requires_grad_(True)
ws = [torch.randn(CO, CI, k, k, device=dev, requires_grad=True) for k in ks]
seg = torch.tensor([R * e // E for e in range(E + 1)], dtype=torch.int32, device=dev) if E > 1 else None
gy = torch.randn(R, HW, HW, CO, device=dev).to(dt)
flops = sum((R // E) * 2.0 * HW * HW * CI * CO * k * k for k in ks)

def run(which):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if which == "fwd":
            with torch.no_grad():
                ops.mp_conv(x, ws, 1.0, seg=seg)
        else:
            y = ops.mp_conv(x, ws, 1.0, seg=seg)
            y.backward(gy)
            x.grad = None
            for w in ws:
                w.grad = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

for which in ("fwd", "fwd", "fwdbwd", "fwdbwd"):
    t = run(which)
    mult = 1 if which == "fwd" else 3
    print(f"{which:7s} {t*1e6:9.1f} us  {mult*flops/t/1e12:8.1f} TFLOP/s  (dtype={dt}, rows={R}, {HW}x{HW}, {CI}->{CO}, ks={ks})")
