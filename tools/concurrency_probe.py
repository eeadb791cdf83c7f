"""dev tool: do two independent chains of large conv launches run faster, equal or slower on two streams than back to back?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"))
import torch
from hdmoe_hip import ops
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
C = 128 if dt == torch.float32 else 64
x1 = torch.randn(256, 32, 32, C, device="cuda").to(dt); x2 = torch.randn(256, 32, 32, C, device="cuda").to(dt)
w1 = torch.randn(C, C, 3, 3, device="cuda"); w2 = torch.randn(C, C, 3, 3, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def chain(x, w, n=10):
    for _ in range(n):
        x = ops.mp_conv(x, w, 1.0)
    return x
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def seq():
    with torch.no_grad(): chain(x1, w1); chain(x2, w2)
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1): chain(x1, w1)
        with torch.cuda.stream(s2): chain(x2, w2)
    cur.wait_stream(s1); cur.wait_stream(s2)
def par_graph():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): par()
    return g
print(f"{dt}: sequential {timed(seq):.2f} ms   two streams {timed(par):.2f} ms")
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1): seq()
g2 = par_graph()
print(f"graph: sequential {timed(g1.replay):.2f} ms   two branches {timed(g2.replay):.2f} ms")
