"""dev tool: the MFMA attention kernels (csrc/attention.hip) at the model's two shapes -- max error against an fp64 softmax on the
bf16-rounded operands (fast shift, forced online maximum, and an input that overflows the fast shift), then kernel times under
hipGraph replay.     python tools/attn_bench.py [--time-only]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), ROOT]
import torch
import hdmoe_hip
from hdmoe_hip._lib import call

dev = "cuda"


def ref(q, k, v, go, H):
    B, Sq, E = q.shape
    Skv, D = k.shape[1], E // H
    qr, kr, vr = (t.double().cpu().requires_grad_(True) for t in (q, k, v))
    s = (qr.view(B, Sq, H, D).transpose(1, 2) @ kr.view(B, Skv, H, D).transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)
    out = (s.softmax(-1) @ vr.view(B, Skv, H, D).transpose(1, 2)).transpose(1, 2).reshape(B, Sq, E)
    out.backward(go.double().cpu())
    lse = torch.logsumexp(s, -1)
    return out.detach(), lse.detach(), qr.grad, kr.grad, vr.grad


def run(q, k, v, go, H):
    B, Sq, E = q.shape
    Skv, D = k.shape[1], E // H
    out = torch.empty_like(q); lse = torch.empty(B, H, Sq, device=dev)
    call("hdmoe_attn_fwd", out, lse, q, k, v, None, B, Sq, Skv, H, D, 0, 1)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Sq, device=dev)
    call("hdmoe_attn_bwd", dq, dk, dv, None, delta, go, out, q, k, v, lse, None, B, Sq, Skv, H, D, 0, 1)
    return out, lse, dq, dk, dv


def err(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


def check(B, Sq, Skv, H, scale_q=1.0, scale_k=1.0, first=None, tag=""):
    torch.manual_seed(Sq * 7 + Skv)
    E = H * 4
    q = (scale_q * torch.randn(B, Sq, E)).bfloat16().to(dev)
    k = (scale_k * torch.randn(B, Skv, E)).bfloat16().to(dev)
    if first is not None:
        k[:, :32] *= first
    v = torch.randn(B, Skv, E).bfloat16().to(dev)
    go = torch.randn(B, Sq, E).bfloat16().to(dev)
    r = ref(q, k, v, go, H)
    ok = True
    for mode in ("fast", "slow"):
        os.environ["HDMOE_ATTN_SLOW"] = "1" if mode == "slow" else "0"
        g = run(q, k, v, go, H)
        e = [err(a, b) for a, b in zip(g, r)]
        good = all(x < 2e-2 for x in e) and all(bool(torch.isfinite(t).all()) for t in g)
        ok &= good
        print(f"{'ok ' if good else 'BAD'} {tag} B={B} Sq={Sq} Skv={Skv} H={H} {mode}: out {e[0]:.2e} lse {e[1]:.2e} dq {e[2]:.2e} dk {e[3]:.2e} dv {e[4]:.2e}", flush=True)
    os.environ["HDMOE_ATTN_SLOW"] = "0"
    return ok


def timeit(B, Sq, Skv, H, iters=10):
    E = H * 4
    q, k, v, go = (torch.randn(B, s, E, device=dev).bfloat16() for s in (Sq, Skv, Skv, Sq))
    out = torch.empty_like(q); lse = torch.empty(B, H, Sq, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Sq, device=dev)
    fw = lambda: call("hdmoe_attn_fwd", out, lse, q, k, v, None, B, Sq, Skv, H, 4, 0, 1)
    bw = lambda: call("hdmoe_attn_bwd", dq, dk, dv, None, delta, go, out, q, k, v, lse, None, B, Sq, Skv, H, 4, 0, 1)
    res = []
    for fn in (fw, bw):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(iters):
                fn()
        gr.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
        res.append(1e3 * s.elapsed_time(e) / iters)
    exps = B * H * Sq * Skv
    merged = Sq <= 1024 and os.environ.get("HDMOE_ATTN_BWD_MERGED", "1") != "0"        # one evaluation of the probabilities instead of two
    print(f"time B={B} Sq={Sq} Skv={Skv} H={H}: fwd {res[0]:7.1f} us ({exps / res[0] / 1e6:5.2f} Texp/s)   bwd ({'merged' if merged else 'dq + dk/dv'}) {res[1]:7.1f} us "
          f"({(1 if merged else 2) * exps / res[1] / 1e6:5.2f} Texp/s)", flush=True)


if __name__ == "__main__":
    hdmoe_hip.lib()
    good = True
    if "--time-only" not in sys.argv:
        good &= check(2, 1024, 1024, 8)
        good &= check(3, 300, 77, 8)
        good &= check(2, 96, 40, 3)
        good &= check(1, 33, 20, 8)                                       # fewer keys than one tile
        good &= check(2, 256, 256, 8, 6.0, 6.0, tag="large scores")
        good &= check(2, 128, 256, 8, 8.0, 30.0, first=1e-3, tag="overflows the first-tile shift")
        good &= check(2, 128, 256, 8, 8.0, 0.05, first=600.0, tag="first tile dominates")
        print("ATTN ALL OK" if good else "ATTN FAILURES", flush=True)
    timeit(256, 1024, 1024, 8)
    timeit(256, 1024, 77, 8)
    sys.exit(0 if good else 1)
