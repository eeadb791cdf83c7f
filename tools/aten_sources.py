"""Which lines of the package still call plain torch ops inside a training step?  One eager fwd + loss + bwd of the wide_config1 fixture (fourth step: the
banked path) under a TorchFunctionMode that logs every non-trivial torch call with the innermost frame inside this package.  Round 4: four calls are left
(three zero_() of the step's persistent buffers, one zeros); the ~70 ATen kernels that remain in the replayed step (fills and adds, 0.44 ms of kernel
time, 0.25 ms of it on the serial stages) are launched by the autograd engine itself -- zero gradients materialised for unused outputs of multi-output
Functions and the sums of gradients of tensors with several consumers."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_bench_path_parity as T  # noqa: E402
from hdmoe_hip import graph as hgraph  # noqa: E402
from hdmoe_hip.dp import GradBuckets  # noqa: E402
from Utils.utils import EDM_LOSS  # noqa: E402

g = torch.load(os.path.join(ROOT, "tests", "golden", "wide_config1.pt"), weights_only=False)
model, kw, inp = T._setup(g, torch.bfloat16, train=True)
lc = g["loss_cfg"]
crit = EDM_LOSS(num_experts=kw["num_experts"], sigma_data=0.5, Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"], z_bal=lc["z_bal"], prior_bal=0.0)
buckets = GradBuckets(model)
x = inp["x"].clone().requires_grad_(True)


def fwd_bwd():
    buckets.zero_grad()
    out = model(x=x, sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["unet_mask"], Vit_router_mask=inp["vit_mask"], zeta=0.1, return_log_var=True, **g["extra"])
    loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
    hgraph.backward(loss["loss"])
    return loss["loss"].detach()


for _ in range(3):
    fwd_bwd()
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.overrides import TorchFunctionMode, resolve_name  # noqa: E402

cnt = collections.Counter()
SKIP = ("size", "shape", "dim", "view", "reshape", "detach", "requires_grad", "is_", "__get__", "stride", "data_ptr", "numel", "dtype", "device", "__getitem__", "unbind",
        "expand", "permute", "transpose", "squeeze", "unsqueeze", "flatten", "chunk", "split", "narrow", "select", "as_strided", "__len__", "element_size", "grad", "__set__",
        "storage", "untyped_storage", "_base", "apply", "backward", "contiguous", "t", "mT", "_version", "is_contiguous", "new_empty", "empty", "empty_like", "item", "tolist", "__hash__", "__repr__", "record_stream")


class Log(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", None) or str(func)
        if not any(name == s_ or name.startswith(s_) for s_ in SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=14)):
                if "heterogeneous-moe" in fr.filename and "tools/" not in fr.filename:
                    where = f"{fr.filename.split('heterogeneous-moe-for-diffusion-models_amd/')[-1]}:{fr.lineno}"
                    break
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    fwd_bwd()
torch.cuda.synchronize()
for (name, where), n in cnt.most_common(60):
    print(f"{n:4d}  {name:22s} {where}")
print("total logged torch calls:", sum(cnt.values()))
