"""Isolated timing (graph replay) of the router-trunk GroupNorm backward with bf16 gradient tensors (hdmoe_gn1t_bwd) and of the bf16 activation
pass (hdmoe_gn1t_act) at the BASELINE configs[1] trunk shapes.   [HDMOE_GN1T_W=4|8] [HDMOE_GNB_GRID=n] python tools/gn1t_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), ROOT]
import torch
from hdmoe_hip._lib import call
dev = "cuda"
for C in (64, 128):
    N, S = 256, 1024
    y = torch.randn(N, S, C, device=dev)
    dz = torch.randn(N, S, C, device=dev).bfloat16()
    dx = torch.empty_like(dz)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    mean, rstd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    dg, db, ws = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.empty(2 * N, device=dev)
    sc, sh = torch.rand(N, C, device=dev), torch.randn(N, C, device=dev)
    a = torch.empty_like(dz)
    def f1(): call("hdmoe_gn1t_bwd", dx, dg, db, ws, dz, None, 1.0, y, gamma, beta, mean, rstd, N, S, C)
    def f2(): call("hdmoe_gn1t_act", a, y, sc, sh, N, S, C)
    for name, f, byts in (("gn1t_bwd (stats + apply)", f1, N * S * C * (4 + 2) * 2 + N * S * C * 2), ("gn1t_act", f2, N * S * C * 6)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side): f()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): f()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / 10
        print(f"C={C} {name}: {us:7.1f} us  {byts / us / 1e6:6.2f} TB/s", flush=True)
