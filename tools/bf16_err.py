"""Measured errors behind the bf16 tolerances: runs tests/test_hip_parity.py::test_full_model_real_widths for the four real-width fixtures in
bf16 mode (and fp32 for comparison) with close_scaled replaced by a recorder; prints max|a - b| / max|b| per compared tensor group."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd", "Utils"), ROOT]
import torch
import test_hip_parity as T

rec = collections.defaultdict(float)
def recorder(a, b, rel, msg="", atol=1e-6, outlier_frac=0.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    fin = torch.isfinite(b)
    if not bool(fin.any()):
        return
    e = float((a[fin] - b[fin]).abs().max()) / max(float(b[fin].abs().max()), 1e-30)
    grp = "router logits" if msg in ("Unet_raw", "vit_raw") else ("x_grad" if msg == "x_grad" else ("outputs" if msg in CUR["outs"] else
          ("router-trunk param grads" if "hard_route" in msg else "other param grads")))
    rec[(CUR["cfg"], CUR["dtype"], grp)] = max(rec[(CUR["cfg"], CUR["dtype"], grp)], e)
T.close_scaled = recorder
T.close = lambda *a, **k: None
CUR = {}
for i in (1, 2, 3, 4):
    g = torch.load(os.path.join(ROOT, "tests", "golden", f"wide_config{i}.pt"), weights_only=False)
    for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        CUR.update(cfg=i, dtype=name, outs=set(g["out"].keys()))
        T.test_full_model_real_widths.__wrapped__(g, dt, 1.0, 1.0) if hasattr(T.test_full_model_real_widths, "__wrapped__") else T.test_full_model_real_widths(g, dt, 1.0, 1.0)
for (cfg, dt, grp), e in sorted(rec.items()):
    print(f"config {cfg} {dt:5s} {grp:26s} max rel err {e:.2e}")
