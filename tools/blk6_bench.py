"""Dev aid: the fused Unet_block launch (csrc/blk6.hip) against the three launches it replaces, per layer shape of BASELINE configs[1]
(N = 512 routed rows, experts [3,3,5,5] with 128 rows each), forward only, eager launches timed with HIP events over REPS back-to-back calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), ROOT]
import torch
import hdmoe_hip
from hdmoe_hip import ops, bank as wbank

hdmoe_hip.set_compute_dtype(torch.bfloat16)
DEV = "cuda"
REPS = int(os.environ.get("REPS", "30"))
shapes = [(32, 32, 32), (64, 32, 32), (96, 32, 32), (64, 64, 32), (32, 32, 16), (64, 64, 16), (96, 64, 16), (128, 64, 16)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
ks = [int(v) for v in os.environ.get("KS", "3,3,5,5").split(",")]
N = int(os.environ.get("ROWS", "512"))
p = float(os.environ.get("P", "0.2"))
G = len(ks)


def timed(fn):
    """GPU time per call: REPS calls captured into one hipGraph (eager launches of these kernels are host-bound), replayed three times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * s.elapsed_time(e) / REPS)
    return best


for Cin, C, HW in shapes:
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w1 = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(C, Cin, k, k)) for k in ks])
            self.w2 = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(C, C, k, k)) for k in ks])
    m = M().to(DEV)
    x = torch.randn(N, HW, HW, Cin, device=DEV).bfloat16()
    res = torch.randn(N, HW, HW, C, device=DEV).bfloat16()
    emb = 1.0 + 0.3 * torch.randn(N, C, device=DEV)
    seg = torch.tensor([N * g // G for g in range(G + 1)], dtype=torch.int32, device=DEV)
    bank = wbank.bank_for(m)
    ops.BLK6 = True
    with torch.no_grad():
        for _ in range(2):
            bank.begin_step(False)
            ops.unet_block_fused(x, res, list(m.w1), list(m.w2), 1.0, 1.0, emb, p, True, seg, 0.7, 0.7)
            wbank.deactivate()
        bank.begin_step(False)

        def fused():
            ops.BLK6 = True
            return ops.unet_block_fused(x, res, list(m.w1), list(m.w2), 1.0, 1.0, emb, p, True, seg, 0.7, 0.7)

        def separate():
            ops.BLK6 = False
            hh = ops.mp_conv_film(x, list(m.w1), 1.0, emb, p, True, seg=seg)
            return ops.mp_conv(hh, list(m.w2), 1.0, seg=seg, res=res, alpha=0.7, beta=0.7, training=True)

        ok = fused() is not None
        if os.environ.get("STAMPS") and ok:
            from hdmoe_hip._lib import lib
            import ctypes
            buf = torch.zeros(8 * 64, dtype=torch.int64, device=DEV)
            lib().hdmoe_blk6_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
            fused(); torch.cuda.synchronize()
            lib().hdmoe_blk6_debug_stamps(None)
            st = buf.cpu().view(8, 64)
            # absolute cycle stamps of the third unit of workgroup 0, per wave (tag:cycles since the unit's first stamp of wave 0)
            rows = []
            for w in range(8):
                row = [(int(v) >> 56, int(v) & ((1 << 56) - 1)) for v in st[w].tolist() if v]
                starts = [i for i, (tag, _) in enumerate(row) if tag == 2]
                rows.append(row[starts[2]:starts[3] + 1] if len(starts) > 3 else row[starts[-2]:starts[-1] + 1])
            t0 = min(r[0][1] for r in rows)
            for w, r in enumerate(rows):
                print(f"  wave {w}: " + " ".join(f"{tag}:{t - t0}" for tag, t in r))
        tf = timed(fused) if ok else float("nan")
        ts = timed(separate)
        wbank.deactivate()
    flops = sum(2.0 * (N // G) * HW * HW * C * (Cin + C) * k * k for k in ks)
    byt = N * HW * HW * 2 * (Cin + 4 * C)
    print(f"Cin={Cin:3d} C={C:2d} {HW}x{HW} ks={ks}: fused {tf:7.1f} us ({flops / tf / 1e6:6.0f} TF/s alg, {byt / tf / 1e3:5.0f} GB/s)   separate {ts:7.1f} us", flush=True)
