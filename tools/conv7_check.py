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
sys.exit(0 if ok else 1)
