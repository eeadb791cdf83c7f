"""Dev check of the conv7 kernel (csrc/conv7.hip, whole-image streaming conv for 32 x 32 maps) through the public op, against torch's
CPU conv2d on the bf16-rounded operands (forward, dgrad, wgrad), then graph-replay timings of the BASELINE config-2 layer classes.
    HDMOE_C7_MINN=1 [HDMOE_C7_G=5] [HDMOE_BWD6=0] python tools/conv7_check.py --check
    [HDMOE_CONV7=0] python tools/conv7_check.py --time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), ROOT, os.path.join(ROOT, "tools")]
import conv6_check as c6

tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("HDMOE_"))
print(f"== conv7_check [{tag}]", flush=True)
ok = True
if "--check" in sys.argv:
    cases = [
        (16, 32, 32, (3, 3, 5, 5), (3, 7, 12, 16), True),
        (13, 32, 32, (5, 3), (4, 13), False),
        (12, 64, 64, (3, 5), (5, 12), True),
        (10, 96, 32, (3, 3, 5, 5), (2, 2, 7, 10), False),
        (9, 64, 32, (5, 3, 5), (3, 3, 9), True),
        (8, 128, 64, (3, 5), (4, 8), False),
        (11, 32, 32, (3, 5, 7), (4, 8, 11), True),
        (7, 64, 64, (7, 3), (3, 7), False),
        (6, 32, 32, (5, 7), (6, 6), False),
        (300, 32, 32, (3, 3, 5, 5), (70, 150, 210, 300), True),
        (10, 32, 96, (3, 3, 5, 5), (2, 5, 7, 10), True),          # three output blocks of 32 over a resident one-chunk image
        (7, 64, 128, (3, 5), (3, 7), False),                      # two output blocks of 64, the two chunks streamed again per block
    ]
    for N, Cin, Cout, ks, split, res in cases:
        ok &= c6.check(N, 32, Cin, Cout, ks, split, res, seed=N)
    cases16 = [
        (16, 64, 64, (3, 3, 5, 5), (3, 7, 12, 16), True),       # odd group sizes: pairs with an absent second image
        (13, 32, 32, (5, 3), (4, 13), False),
        (9, 128, 64, (3, 5), (5, 9), True),
        (10, 96, 64, (3, 3, 5, 5), (2, 2, 7, 10), False),
        (11, 64, 64, (3, 5, 7), (4, 8, 11), True),
        (6, 64, 32, (5, 7), (6, 6), False),
        (301, 64, 64, (3, 3, 5, 5), (70, 151, 210, 301), True),
        (9, 64, 128, (3, 5), (4, 9), True),
        (10, 64, 96, (5, 3), (5, 10), False),
        (5, 32, 160, (3, 7), (2, 5), False),
    ]
    for N, Cin, Cout, ks, split, res in cases16:
        ok &= c6.check(N, 16, Cin, Cout, ks, split, res, seed=N + 1)
    print("ALL OK" if ok else "FAILURES", flush=True)
if "--time" in sys.argv:
    for Cin, Cout in ((32, 32), (64, 64), (96, 32), (64, 32)):
        c6.timeit(512, 32, Cin, Cout, (3, 3, 5, 5))
    c6.timeit(512, 32, 32, 32, (3, 3, 3, 3))
    c6.timeit(512, 32, 32, 32, (5, 5, 5, 5))
    c6.timeit(256, 32, 32, 32, (3, 3, 5, 5))
    c6.timeit(1024, 32, 32, 32, (3, 3, 5, 5))
    for Cin, Cout in ((64, 64), (128, 64), (96, 64), (32, 32)):
        c6.timeit(512, 16, Cin, Cout, (3, 3, 5, 5))
    c6.timeit(512, 16, 64, 64, (3, 3, 3, 3))
    c6.timeit(512, 16, 64, 64, (5, 5, 5, 5))
if ("--check" in sys.argv or "--time" in sys.argv) and not any(f in sys.argv for f in ("--check-bwd", "--time-bwd", "--stamps")):
    sys.exit(0 if ok else 1)


# ---- the fused backward launch (hdmoe_conv_bwd6: dgrad program + weight-gradient programs in one grid) called directly --------------------
def _bwd_setup(N, R, Cin, Cout, ks, split, seed):
    import torch
    from hdmoe_hip._lib import call, lib, _int_array
    import ctypes
    dev = "cuda"
    g = torch.Generator().manual_seed(seed)
    E = len(ks)
    x = torch.randn(N, R, R, Cin, generator=g).bfloat16()
    dy = torch.randn(N, R, R, Cout, generator=g).bfloat16()
    ws = [(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16() for k in ks]
    seg = [0] + list(split)
    O, I = Cout, Cin
    Opad = (O + 15) // 16 * 16
    taps = max(k * k for k in ks)
    wstride, wdstride = taps * O * I, taps * I * Opad
    wdv = [w.float().to(dev) for w in ws]
    wf = torch.empty(E * wstride, dtype=torch.bfloat16, device=dev)
    wd = torch.empty(E * wdstride, dtype=torch.bfloat16, device=dev)
    call("hdmoe_wprep_fwd", wdv, None, 1.0, list(ks), list(ks), E, O, I, I, Opad, wf, wstride, wd, wdstride, 0, 0, 1, 1)
    kib = lib().hdmoe_conv_wgrad6_ws_kib(E, N, R, R, I, O, ctypes.cast(_int_array(ks), ctypes.c_void_p), ctypes.cast(_int_array(ks), ctypes.c_void_p), 1)
    assert kib > 0
    wsb = torch.empty(2 * kib * 256, dtype=torch.float32, device=dev)
    Gs = [torch.zeros(k * k, O, I, device=dev) for k in ks]
    segd = torch.tensor(seg, dtype=torch.int32, device=dev)
    xd, dyd = x.to(dev), dy.to(dev)
    dx = torch.empty_like(xd)
    pts = [(k - 1) // 2 for k in ks]
    args = (xd, dyd, wd, dx, Gs, segd, E, wdstride, N, R, R, I, O, list(ks), list(ks), pts, pts, 1.0, wsb, wsb.numel() * 4, 1)
    dims = [E, N, R, R, I, O, 1, 0] + list(ks) + [0] * (8 - E)
    return x, dy, ws, seg, Gs, dx, segd, wsb, args, dims


def check_bwd(N, R, Cin, Cout, ks, split, seed=0):
    import torch
    import torch.nn.functional as F
    from hdmoe_hip._lib import call
    x, dy, ws, seg, Gs, dx, segd, wsb, args, dims = _bwd_setup(N, R, Cin, Cout, ks, split, seed)
    rc = call("hdmoe_conv_bwd6", *args)
    if rc != 0:
        print(f"--  bwd N={N} R={R} {Cin}->{Cout} ks={ks}: outside the fused launch's domain (rc {rc})", flush=True)
        return True
    call("hdmoe_conv_wgrad6_reduce_batch", Gs + [None] * (8 - len(Gs)), [segd], [wsb], dims, 1)
    torch.cuda.synchronize()
    xr = x.float().requires_grad_(True)
    wr = [w.float().requires_grad_(True) for w in ws]
    outs = []
    for gi, w in enumerate(wr):
        xs = xr[seg[gi]:seg[gi + 1]].permute(0, 3, 1, 2)
        k = w.shape[-1]; pl = (k - 1) // 2
        outs.append(F.conv2d(F.pad(xs, (pl, k - 1 - pl, pl, k - 1 - pl)), w).permute(0, 2, 3, 1))
    (torch.cat(outs, 0) * dy.float()).sum().backward()
    gerr = float((dx.float().cpu() - xr.grad).abs().max()) / float(xr.grad.abs().max())
    werr = 0.0
    for gi, k in enumerate(ks):
        ref = wr[gi].grad.permute(2, 3, 0, 1).reshape(k * k, Cout, Cin)
        if seg[gi + 1] > seg[gi]:
            werr = max(werr, float((Gs[gi].cpu() - ref).abs().max()) / float(ref.abs().max()))
        else:
            werr = max(werr, float(Gs[gi].abs().max()))
    ok = gerr < 2e-2 and werr < 1e-3
    print(f"{'ok ' if ok else 'BAD'} bwd N={N} R={R} {Cin}->{Cout} ks={ks} split={split}: dgrad {gerr:.2e} wgrad {werr:.2e}", flush=True)
    return ok


def time_bwd(N, R, Cin, Cout, ks, iters=10):
    import torch
    from hdmoe_hip._lib import call
    E = len(ks)
    split = [N * (i + 1) // E for i in range(E)]
    x, dy, ws, seg, Gs, dx, segd, wsb, args, dims = _bwd_setup(N, R, Cin, Cout, ks, split, 1)
    for _ in range(3):
        call("hdmoe_conv_bwd6", *args)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call("hdmoe_conv_bwd6", *args)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            call("hdmoe_conv_bwd6", *args)
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    fl = 2 * sum(2.0 * (N // E) * R * R * Cout * Cin * k * k for k in ks)
    print(f"time bwd (dgrad + wgrad) N={N} R={R} {Cin}->{Cout} ks={ks}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)


if "--check-bwd" in sys.argv:
    okb = True
    okb &= check_bwd(16, 32, 32, 32, (3, 3, 5, 5), (3, 7, 12, 16), seed=1)
    okb &= check_bwd(9, 32, 64, 64, (3, 5), (5, 9), seed=2)
    okb &= check_bwd(10, 32, 96, 32, (5, 3), (4, 10), seed=3)
    okb &= check_bwd(7, 32, 64, 32, (3, 5, 3), (2, 5, 7), seed=4)
    okb &= check_bwd(6, 32, 32, 32, (5, 3), (6, 6), seed=5)               # empty 3x3 class member
    okb &= check_bwd(300, 32, 32, 32, (3, 3, 5, 5), (70, 150, 210, 300), seed=6)
    okb &= check_bwd(20, 32, 128, 128, (3,), (20,), seed=9)               # 2 x 2 chunk pairs per workgroup (router-trunk shape)
    okb &= check_bwd(12, 32, 64, 128, (3, 3), (5, 12), seed=10)
    okb &= check_bwd(9, 32, 32, 64, (3, 5), (4, 9), seed=11)              # 1 x 2 chunks
    okb &= check_bwd(13, 32, 128, 32, (5, 3), (6, 13), seed=12)           # 2 x 1 chunks
    okb &= check_bwd(210, 32, 64, 64, (3, 3, 5, 5), (50, 110, 160, 210), seed=13)
    okb &= check_bwd(11, 16, 64, 64, (3, 3, 5, 5), (2, 5, 8, 11), seed=7)
    okb &= check_bwd(300, 16, 64, 64, (3, 5), (140, 300), seed=8)
    print("BWD ALL OK" if okb else "BWD FAILURES", flush=True)
    if not okb or not any(f in sys.argv for f in ("--time-bwd", "--stamps")):
        sys.exit(0 if okb else 1)
if "--time-bwd" in sys.argv:
    for Cin, Cout in ((32, 32), (64, 64), (96, 32), (64, 32)):
        time_bwd(512, 32, Cin, Cout, (3, 3, 5, 5))
    for Cin, Cout in ((64, 64), (128, 64), (32, 32)):
        time_bwd(512, 16, Cin, Cout, (3, 3, 5, 5))
    for Cin, Cout in ((128, 128), (64, 128), (32, 64)):                    # router-trunk layers (one 3x3 class)
        time_bwd(256, 32, Cin, Cout, (3,))


def stamps7(N, R, Cin, Cout, ks):
    """One conv7 launch with in-kernel stamps of workgroup 0: per wave (tag: cycles since the previous stamp).  tags: 1 start, 2 at a chunk's first
    stage barrier, 3 past it (tile + weights landed), 4 MFMA stages done, 5 epilogue issued."""
    import ctypes, torch
    from hdmoe_hip._lib import call, lib
    dev = "cuda"
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
    names = {1: "start", 2: ">bar", 3: "<bar", 4: "mfma'd", 5: "stored", 6: "segs", 7: "zeroed", 8: "consts", 9: "issued"}
    print(f"stamps N={N} R={R} {Cin}->{Cout} ks={ks}  (s_memtime ticks; tag:delta)")
    for w in (0, 3, 7):
        prev, out = t0, []
        for i in range(64):
            v = int(b[w, i])
            if v == 0:
                break
            tag, t = (v >> 56) & 0xFF, v & ((1 << 56) - 1)
            out.append(f"{names.get(tag, tag)}:{t - prev}")
            prev = t
        print(f" wave {w}: " + " ".join(out), flush=True)


if "--stamps" in sys.argv:
    stamps7(512, 32, 32, 32, (3, 3, 5, 5))
    stamps7(512, 32, 32, 32, (3, 3, 5, 5))
    stamps7(512, 32, 64, 64, (3, 3, 5, 5))
    stamps7(512, 16, 64, 64, (3, 3, 5, 5))


def stamps_bwd(N, R, Cin, Cout, ks):
    """One fused backward launch with in-kernel stamps: dgrad workgroup 0 and the first weight-gradient workgroup of each class.
    wgrad7 tags: 1 start, 2 slot decoded, 3 first unit issued, 4 / 5 before / after a unit's barrier, 6 MFMAs done, 7 reduced and stored."""
    import ctypes, torch
    from hdmoe_hip._lib import call, lib
    E = len(ks)
    split = [N * (i + 1) // E for i in range(E)]
    x, dy, ws, seg, Gs, dx, segd, wsb, args, dims = _bwd_setup(N, R, Cin, Cout, ks, split, 1)
    for _ in range(3):
        call("hdmoe_conv_bwd6", *args)
    buf = torch.zeros(3 * 512, dtype=torch.int64, device="cuda")
    lib().hdmoe_conv6_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
    call("hdmoe_conv_bwd6", *args)
    torch.cuda.synchronize()
    lib().hdmoe_conv6_debug_stamps(None)
    b = buf.cpu().view(3, 8, 64)
    t0 = min(int(b[0, w, 0]) & ((1 << 56) - 1) for w in range(8))
    print(f"bwd stamps N={N} R={R} {Cin}->{Cout} ks={ks}  (s_memtime ticks since the dgrad workgroup's start; tag:delta)")
    for sect, nm in ((0, "dgrad"), (1, "wgrad 3x3"), (2, "wgrad 5x5")):
        for w in (0, 7):
            prev, out = t0, []
            for i in range(64):
                v = int(b[sect, w, i])
                if v == 0:
                    break
                tag, t = (v >> 56) & 0xFF, v & ((1 << 56) - 1)
                out.append(f"{tag}:{t - prev}")
                prev = t
            print(f" {nm} wave {w}: " + " ".join(out), flush=True)


if "--stamps-bwd" in sys.argv:
    stamps_bwd(512, 32, 32, 32, (3, 3, 5, 5))
    stamps_bwd(512, 32, 64, 64, (3, 3, 5, 5))
