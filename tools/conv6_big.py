"""dev tool: conv6 / wgrad6 on layers several times larger than the BASELINE model's (what the kernels reach when a launch is not dominated by
its fixed costs).  usage (GPU box): python tools/conv6_big.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools")]
sys.argv = [sys.argv[0], "--none"]
import conv6_check as c
for (N, R, Ci, Co, ks) in [(512, 32, 128, 128, [3, 3, 5, 5]), (512, 32, 128, 128, [5]), (1024, 32, 128, 128, [5]), (512, 64, 64, 64, [5]),
                           (2048, 32, 64, 64, [5]), (512, 32, 256, 256, [3]), (512, 32, 256, 256, [5])]:
    c.timeit(N, R, Ci, Co, ks)
for (N, R, Ci, Co, ks) in [(512, 32, 128, 128, [5]), (1024, 32, 128, 128, [3, 3, 5, 5]), (2048, 32, 64, 64, [5])]:
    c.time_wgrad(N, R, Ci, Co, ks)
