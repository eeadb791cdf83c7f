"""Dev tool: BASELINE config 5 shape on one GPU -- EDM_Sampler (Heun, N solver steps = 2N-1 denoiser evaluations) on 4x64x64
latents, 8 heterogeneous experts top-2, bf16, eval, hipGraph replay of the denoiser call.  usage: sampler_bench.py [batch=128] [N=40]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd"), os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd", "Utils")]
import torch
import hdmoe_hip
from Utils import configs
from Utils.EDM_sampler import EDM_Sampler
from models import model_config2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
hdmoe_hip.set_compute_dtype(torch.bfloat16)
kw = configs.model_kwargs(**configs.BASELINE_CONFIGS[4]["over"])
torch.manual_seed(0)
model = model_config2.preconditioned_HDMOEM(**kw).cuda().eval()
with torch.no_grad():
    for n, p in model.named_parameters():
        if n.endswith("out_gain"): p.fill_(0.5)
noise = torch.randn(B, 4, 64, 64, device="cuda")
text = torch.randn(B, 77, 768, device="cuda")
for graph in (False, True):
    s = EDM_Sampler(model, Guide_net=model, guidance=1.0, num_solve_steps=N, use_graph=graph)
    with torch.no_grad():
        s.sample(noise=noise, text_emb=text, transition_mean=-1.2, softness=1.2)     # warm-up (+ capture)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = s.sample(noise=noise, text_emb=text, transition_mean=-1.2, softness=1.2)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    print(f"graph={graph}: B={B} N={N} ({2*N-1} evals): {dt:.3f} s -> {B/dt:.1f} imgs/s, {1e3*dt/(2*N-1):.2f} ms per denoiser evaluation")
