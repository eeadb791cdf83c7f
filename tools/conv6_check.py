"""Dev check of the conv6 kernel (csrc/conv6.hip) through the public op: grouped heterogeneous k x k convs, forward + dgrad,
against torch's CPU conv2d on the bf16-rounded operands; then timings of the BASELINE config-2 layer classes.
    python tools/conv6_check.py [--time-only] [--check-only]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), ROOT]
import torch
import torch.nn.functional as F
import hdmoe_hip
from hdmoe_hip import ops

dev = "cuda"


def ref_conv(x, ws, seg, res, alpha, beta):
    """x (N,H,W,C) bf16 on cpu float; per-group conv2d with 'same' padding as MP_Conv (pad (k-1)//2 left)."""
    outs = []
    for g, w in enumerate(ws):
        xs = x[seg[g]:seg[g + 1]].permute(0, 3, 1, 2).float()
        k = w.shape[-1]
        pl = (k - 1) // 2
        xs = F.pad(xs, (pl, k - 1 - pl, pl, k - 1 - pl))
        outs.append(F.conv2d(xs, w.float()).permute(0, 2, 3, 1))
    y = alpha * torch.cat(outs, 0)
    if res is not None:
        y = y + beta * res.float()
    return y


def check(N, R, Cin, Cout, ks, split, with_res, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, R, R, Cin, generator=g).bfloat16()
    ws = [(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16() for k in ks]
    seg = [0] + list(split)
    assert seg[-1] == N and len(seg) == len(ks) + 1
    res = torch.randn(N, R, R, Cout, generator=g).bfloat16() if with_res else None
    alpha, beta = (0.7, 0.6) if with_res else (1.0, 0.0)
    yref = ref_conv(x, ws, seg, res, alpha, beta)
    xd = x.to(dev).requires_grad_(True)
    wd = [torch.nn.Parameter(w.float().to(dev)) for w in ws]
    segd = torch.tensor(seg, dtype=torch.int32, device=dev)
    y = ops.mp_conv(xd, wd, 1.0, seg=segd, res=None if res is None else res.to(dev), alpha=alpha, beta=beta, normalize=False)
    err = float((y.float().cpu() - yref).abs().max()) / float(yref.abs().max())
    # dgrad
    go = torch.randn(N, R, R, Cout, generator=g).bfloat16()
    y.backward(go.to(dev))
    xr = x.float().requires_grad_(True)
    wr = [w.float().requires_grad_(True) for w in ws]
    outs = []
    for gi, w in enumerate(wr):
        xs = xr[seg[gi]:seg[gi + 1]].permute(0, 3, 1, 2)
        k = w.shape[-1]; pl = (k - 1) // 2
        outs.append(F.conv2d(F.pad(xs, (pl, k - 1 - pl, pl, k - 1 - pl)), w).permute(0, 2, 3, 1))
    (alpha * torch.cat(outs, 0) * go.float()).sum().backward()
    gerr = float((xd.grad.float().cpu() - xr.grad).abs().max()) / float(xr.grad.abs().max())
    werr = 0.0
    for gi in range(len(ks)):
        if seg[gi + 1] > seg[gi]:
            werr = max(werr, float((wd[gi].grad.cpu() - wr[gi].grad).abs().max()) / float(wr[gi].grad.abs().max()))
        else:
            werr = max(werr, float(wd[gi].grad.abs().max()))
    ok = err < 2e-2 and gerr < 2e-2 and werr < 2e-2
    print(f"{'ok ' if ok else 'BAD'} N={N} R={R} {Cin}->{Cout} ks={ks} split={split} res={with_res}: fwd {err:.2e} dgrad {gerr:.2e} wgrad {werr:.2e}", flush=True)
    return ok


def check_split(N, R, Cin, Cout, with_res=False, seed=0):
    """fp32 tensors through the split-bf16 kernel (csrc/conv6s.hip) against an fp64 CPU conv: expect ~1e-5 relative."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, R, R, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xd = x.to(dev).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(dev))
    y = ops.mp_conv(xd, wd, 1.0, normalize=False, split=True)
    go = torch.randn(N, R, R, Cout, generator=g)
    y.backward(go.to(dev))
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(F.pad(x64.permute(0, 3, 1, 2), (1, 1, 1, 1)), w64).permute(0, 2, 3, 1)
    (y64 * go.double()).sum().backward()
    e1 = float((y.detach().cpu().double() - y64.detach()).abs().max() / y64.abs().max())
    e2 = float((xd.grad.cpu().double() - x64.grad).abs().max() / x64.grad.abs().max())
    e3 = float((wd.grad.cpu().double() - w64.grad).abs().max() / w64.grad.abs().max())
    ok = e1 < 5e-5 and e2 < 5e-5 and e3 < 5e-5
    print(f"{'ok ' if ok else 'BAD'} split N={N} R={R} {Cin}->{Cout}: fwd {e1:.2e} dgrad {e2:.2e} wgrad {e3:.2e}", flush=True)
    return ok


def time_split(N, R, Cin, Cout, iters=10):
    x = torch.randn(N, R, R, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    from hdmoe_hip._lib import call
    res = {}
    for name, code in (("split", 2), ("fp32", 0)):
        wstride = 9 * Cout * Cin
        wf = torch.empty((2 if code == 2 else 1) * wstride, dtype=torch.bfloat16 if code == 2 else torch.float32, device=dev)
        call("hdmoe_wprep_fwd", [w], None, 1.0, [3], [3], 1, Cout, Cin, Cin, Cout, wf, wstride, None, 0, 1, 0, 1, code)
        y = torch.empty(N, R, R, Cout, device=dev)
        args = (x, wf, y, None, 1.0, 0.0, None, 1, wstride, N, R, R, R, R, Cin, Cin, Cin, Cout, Cout, 1, 0, [3], [3], [1], [1], code)
        for _ in range(2):
            call("hdmoe_conv_fwd", *args)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            call("hdmoe_conv_fwd", *args)
        e.record()
        torch.cuda.synchronize()
        res[name] = 1e3 * s.elapsed_time(e) / iters
    fl = 2.0 * N * R * R * Cout * Cin * 9
    print(f"time split N={N} R={R} {Cin}->{Cout}: split {res['split']:8.1f} us ({fl / res['split'] / 1e6:6.1f} TF/s)   fp32 MFMA {res['fp32']:8.1f} us ({fl / res['fp32'] / 1e6:6.1f} TF/s)", flush=True)


def timeit(N, R, Cin, Cout, ks, iters=20):
    x = torch.randn(N, R, R, Cin, device=dev).bfloat16()
    wd = [torch.randn(Cout, Cin, k, k, device=dev) for k in ks]
    E = len(ks)
    seg = torch.tensor([N * i // E for i in range(E + 1)], dtype=torch.int32, device=dev)
    with torch.no_grad():
        for _ in range(3):
            y = ops.mp_conv(x, wd, 1.0, seg=seg)
        # time the conv launch alone: prepare weights once, call the C entry directly
        from hdmoe_hip._lib import call
        O, I = Cout, Cin
        taps = max(k * k for k in ks)
        wstride = taps * O * I
        wf = torch.empty(E * wstride, dtype=torch.bfloat16, device=dev)
        call("hdmoe_wprep_fwd", wd, None, 1.0, list(ks), list(ks), E, O, I, I, (O + 15) // 16 * 16, wf, wstride, None, 0, 1, 0, 1, 1)
        y = torch.empty(N, R, R, O, dtype=torch.bfloat16, device=dev)
        pts = [(k - 1) // 2 for k in ks]
        args = (x, wf, y, None, 1.0, 0.0, seg, E, wstride, N, R, R, R, R, I, I, I, O, O, 1, 0, list(ks), list(ks), pts, pts, 1)
        for _ in range(3):
            call("hdmoe_conv_fwd", *args)
        torch.cuda.synchronize()
        # kernel time without host launch gaps: replay `iters` captured launches as one hipGraph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call("hdmoe_conv_fwd", *args)
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(iters):
                call("hdmoe_conv_fwd", *args)
        gr.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.replay()
        e.record()
        torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    flops = sum(2.0 * (N // E) * R * R * Cout * Cin * k * k for k in ks)
    byts = 2.0 * N * R * R * (Cin + Cout)
    print(f"time N={N} R={R} {Cin}->{Cout} ks={ks}: {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  {byts / us / 1e3:7.1f} GB/s", flush=True)


def time_wgrad(N, R, Cin, Cout, ks, iters=10):
    """Kernel time of the weight gradient (graph replay): wgrad6 + reduce vs the general kernel (HDMOE_WGRAD6=0 in another process)."""
    from hdmoe_hip._lib import call
    x = torch.randn(N, R, R, Cin, device=dev).bfloat16()
    dy = torch.randn(N, R, R, Cout, device=dev).bfloat16()
    E = len(ks)
    seg = torch.tensor([N * i // E for i in range(E + 1)], dtype=torch.int32, device=dev)
    Gs = [torch.zeros(k * k, Cout, Cin, device=dev) for k in ks]
    pts = [(k - 1) // 2 for k in ks]
    info = dict()
    def run():
        ops._wgrad(info, x, dy, Gs, seg, E, N, R, R, R, R, Cin, Cin, Cout, False, list(ks), list(ks), pts)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            run()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    fl = sum(2.0 * (N // E) * R * R * Cout * Cin * k * k for k in ks)
    print(f"time wgrad N={N} R={R} {Cin}->{Cout} ks={ks}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)


def stamps(N, R, Cin, Cout, ks):
    """One launch with in-kernel stamps of workgroup 0: per wave, (tag, cycles since the first stamp)."""
    import ctypes
    from hdmoe_hip._lib import call, lib
    x = torch.randn(N, R, R, Cin, device=dev).bfloat16()
    wd = [torch.randn(Cout, Cin, k, k, device=dev) for k in ks]
    E = len(ks)
    seg = torch.tensor([N * i // E for i in range(E + 1)], dtype=torch.int32, device=dev)
    O, I = Cout, Cin
    wstride = max(k * k for k in ks) * O * I
    wf = torch.empty(E * wstride, dtype=torch.bfloat16, device=dev)
    call("hdmoe_wprep_fwd", wd, None, 1.0, list(ks), list(ks), E, O, I, I, (O + 15) // 16 * 16, wf, wstride, None, 0, 1, 0, 1, 1)
    y = torch.empty(N, R, R, O, dtype=torch.bfloat16, device=dev)
    pts = [(k - 1) // 2 for k in ks]
    args = (x, wf, y, None, 1.0, 0.0, seg, E, wstride, N, R, R, R, R, I, I, I, O, O, 1, 0, list(ks), list(ks), pts, pts, 1)
    for _ in range(3):
        call("hdmoe_conv_fwd", *args)
    buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
    lib().hdmoe_conv6_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
    call("hdmoe_conv_fwd", *args)
    torch.cuda.synchronize()
    lib().hdmoe_conv6_debug_stamps(None)
    b = buf.cpu().view(8, 64)
    t0 = min(int(b[w, 0]) & ((1 << 56) - 1) for w in range(8))
    names = {1: "start", 2: "table", 3: "decode", 4: "plan", 5: "nxt>", 6: "<nxt", 7: ">bar", 8: "<bar", 9: "dma'd", 10: "mfma'd", 11: "stored"}
    print(f"stamps N={N} R={R} {Cin}->{Cout} ks={ks} (cycles @100MHz memtime ticks x clock ratio; tag:delta)")
    for w in (0, 7):
        prev, out = t0, []
        for i in range(64):
            v = int(b[w, i])
            if v == 0:
                break
            tag, t = (v >> 56) & 0xFF, v & ((1 << 56) - 1)
            out.append(f"{names.get(tag, tag)}:{t - prev}")
            prev = t
        print(f" wave {w}: " + " ".join(out), flush=True)


if __name__ == "__main__":
    hdmoe_hip.lib()
    if "--wgrad" in sys.argv:
        for (N, R, Ci, Co) in [(512, 32, 32, 32), (512, 16, 64, 64), (512, 32, 64, 64), (512, 32, 96, 32), (512, 16, 128, 64)]:
            time_wgrad(N, R, Ci, Co, [3, 3, 5, 5])
        time_wgrad(512, 16, 64, 64, [5]); time_wgrad(512, 16, 64, 64, [3]); time_wgrad(512, 32, 32, 32, [5])
        sys.exit(0)
    if "--split" in sys.argv:
        good = True
        good &= check_split(3, 32, 32, 64)
        good &= check_split(5, 32, 64, 128)
        good &= check_split(4, 16, 128, 128)
        good &= check_split(70, 32, 128, 128, seed=3)
        good &= check_split(9, 64, 64, 32)
        print("SPLIT ALL OK" if good else "SPLIT FAILURES", flush=True)
        for (Ci, Co) in [(32, 64), (64, 128), (128, 128), (128, 64), (64, 32)]:
            time_split(256, 32, Ci, Co)
        sys.exit(0 if good else 1)
    if "--stamps" in sys.argv:
        stamps(512, 32, 32, 32, [3]); stamps(512, 16, 64, 64, [5]); stamps(512, 32, 64, 64, [3, 3, 5, 5])
        sys.exit(0)
    good = True
    if "--time-only" not in sys.argv:
        good &= check(4, 32, 32, 32, [3], [4], False)
        good &= check(6, 32, 32, 32, [3, 5], [2, 6], True)
        good &= check(5, 16, 64, 64, [5, 3], [3, 5], False)
        good &= check(7, 16, 64, 64, [3, 3, 5, 5], [1, 1, 4, 7], True)
        good &= check(9, 32, 96, 32, [3, 5, 7], [2, 5, 9], True)
        good &= check(6, 16, 128, 64, [5, 7], [6, 6], False)       # empty second group
        good &= check(3, 64, 32, 32, [7, 3], [1, 3], False)
        good &= check(3, 32, 64, 128, [3], [3], True)               # two output-channel blocks
        good &= check(300, 16, 32, 64, [3, 5], [100, 300], False)   # many units per workgroup, MT = 1
        good &= check(300, 32, 32, 32, [5, 3], [120, 300], True)    # MT = 2 units
        print("ALL OK" if good else "FAILURES", flush=True)
    if "--check-only" not in sys.argv:
        for (N, R, Ci, Co) in [(512, 32, 32, 32), (512, 16, 64, 64), (512, 32, 64, 64), (512, 32, 96, 32), (512, 32, 64, 32),
                               (512, 16, 128, 64), (512, 16, 96, 64)]:
            timeit(N, R, Ci, Co, [3, 3, 5, 5])
        timeit(512, 32, 32, 32, [3]); timeit(512, 32, 32, 32, [5]); timeit(512, 16, 64, 64, [3]); timeit(512, 16, 64, 64, [5])
    sys.exit(0 if good else 1)
