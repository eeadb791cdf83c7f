"""The whole-image streaming kernels of round 4 -- conv7 (forward / input gradient), the fused backward launch bwd7 with the weight-gradient
programs wgrad7 (5x5) and wgrad8 (3x3, up to 2 x 2 channel chunks per workgroup) -- against torch's conv2d on the bf16-rounded operands
(autograd of MP_Conv, reference models/model_internals.py:253-275 via F.conv2d; the grouped dispatch of model_components.py:232-253).

They take over from conv6 / wgrad6 at >= 192 routed rows on 32 x 32 / 16 x 16 maps only, i.e. at the benchmark's sizes and not in the small
fixtures: these cases call the public entry points at such sizes (ragged segments, an expert without rows, one to four channel chunks, one
kernel-size class alone as in the router trunks).  The checks themselves live in tools/conv6_check.py / tools/conv7_check.py (also used for the
graph-replay timings in profiles/)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checks():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import hdmoe_hip
    hdmoe_hip.lib()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    argv, sys.argv = sys.argv, ["conv7_check.py"]            # (the tool is a script: no mode flags, it only defines its functions)
    try:
        import conv6_check
        import conv7_check
    finally:
        sys.argv = argv
    return conv6_check, conv7_check


FWD = [  # N, R, Cin, Cout, kernel sizes, segment ends, residual
    (300, 32, 32, 32, (3, 3, 5, 5), (70, 150, 210, 300), True),
    (200, 32, 64, 64, (3, 5), (90, 200), True),
    (210, 32, 96, 32, (5, 3), (100, 210), False),
    (200, 32, 32, 96, (3, 3, 5, 5), (40, 40, 130, 200), True),    # an expert without rows; three output blocks over a resident image
    (196, 32, 32, 32, (3, 5, 7), (60, 130, 196), False),
    (301, 16, 64, 64, (3, 3, 5, 5), (70, 151, 210, 301), True),   # odd group sizes: image pairs with an absent second image
    (200, 16, 128, 64, (3, 5), (99, 200), False),
]


@pytest.mark.parametrize("N,R,Cin,Cout,ks,split,res", FWD, ids=[f"{c[0]}x{c[1]}_{c[2]}to{c[3]}_k{''.join(map(str, c[4]))}" for c in FWD])
def test_conv7_forward_dgrad_wgrad_match_conv2d(checks, N, R, Cin, Cout, ks, split, res):
    c6, _ = checks
    assert c6.check(N, R, Cin, Cout, ks, split, res, seed=N + R)


BWD = [  # N, R, Cin, Cout, kernel sizes, segment ends
    (300, 32, 32, 32, (3, 3, 5, 5), (70, 150, 210, 300)),
    (210, 32, 64, 64, (3, 3, 5, 5), (50, 110, 160, 210)),         # 3x3 class: 2 x 2 chunks per workgroup (wgrad8<4, 8>)
    (200, 32, 128, 128, (3,), (200,)),                            # one class alone, four chunk pairs per side (router-trunk shape)
    (200, 32, 64, 128, (3, 3), (80, 200)),
    (200, 32, 32, 64, (3, 5), (100, 200)),                        # 1 x 2 chunks
    (200, 32, 128, 32, (5, 3), (90, 200)),                        # 2 x 1 chunks
    (200, 32, 96, 32, (5, 3), (200, 200)),                        # the 3x3 expert without rows
    (300, 16, 64, 64, (3, 5), (140, 300)),
]


@pytest.mark.parametrize("N,R,Cin,Cout,ks,split", BWD, ids=[f"{c[0]}x{c[1]}_{c[2]}to{c[3]}_k{''.join(map(str, c[4]))}" for c in BWD])
def test_fused_backward_launch_matches_conv2d(checks, N, R, Cin, Cout, ks, split):
    _, c7 = checks
    assert c7.check_bwd(N, R, Cin, Cout, ks, split, seed=N + Cin)
