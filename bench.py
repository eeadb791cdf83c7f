#!/usr/bin/env python3
"""bench.py -- denoise-steps/sec (fwd + EDM_LOSS + bwd) of the HDMOEM hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): models.model_config1.preconditioned_HDMOEM, reference Utils/configs.py shapes
(internal_channels 32, R = 32, 4 experts [3x3,3x3,5x5,5x5] / ViT patches [4,8,8,16]) with top_k = 2, synthetic 4x32x32
latents, per-GPU batch 256, bf16 experts / fp32 stem + router trunks (the trunks' convs: fp32-equivalent split-bf16 arithmetic in the
forward -- routing indices bit-exact -- and bf16 operands with fp32 accumulation in the backward), train() mode (dropout, logit noise, forced weight
normalisation all run).  One step = forward + fused EDM_LOSS + backward (+ gradient all-reduce when N > 1), weak scaling.
Prints ONE JSON line on rank 0.  Only the `cpu_baseline` leg touches oracle/ (the CPU restatement, timed as a baseline).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "heterogeneous-moe-for-diffusion-models_amd")
for p in (PKG, os.path.join(PKG, "Utils"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

# dense TFLOP/s, MI355X_MICROARCH.md "Chip-level parameters".  split_bf16 = fp32 tensors computed as three bf16 MFMAs per
# product (csrc/conv6s.hip): its peak in ALGORITHMIC (fp32-equivalent) FLOP/s is a third of the bf16 peak.
MFMA_PEAK = {"bfloat16": 2500.0, "float32": 157.3, "split_bf16": 2500.0 / 3}
HBM_PEAK_GBS = 8000.0


def build_model(cfg_id, device):
    import hdmoe_hip
    import configs as C
    from models import model_config1, model_config2
    bc = C.BASELINE_CONFIGS[cfg_id]
    kw = C.model_kwargs(**bc["over"])
    mod = model_config1 if bc["module"] == 1 else model_config2
    torch.manual_seed(1234)                                  # identical replicas on every rank
    model = mod.preconditioned_HDMOEM(**kw)
    with torch.no_grad():                                    # zero-inits would make the experts' output identically 0
        for n, p in model.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
            elif n.endswith("alpha_txt"):
                p.fill_(0.3)
    hdmoe_hip.set_compute_dtype(torch.bfloat16 if bc["dtype"] == "bf16" else torch.float32)
    return model.to(device).train(), kw, bc


def make_inputs(kw, B, device, seed, module):
    """Seeded synthetic step inputs mirroring reference Utils/training.py:125-153."""
    import utils as U
    g = torch.Generator(device=device).manual_seed(seed)
    R, Cl = kw["IN_img_resolution"], kw["IN_in_channels"]
    x0 = 0.5 * torch.randn(B, Cl, R, R, device=device, generator=g)
    sigma = U.sample_sigma_hybrid(B, 0.002, 80.0, p_mean=-1.2, p_std=1.6, extreme_prob=0.5, device=device, generator=g)
    x = x0 + sigma * torch.randn(B, Cl, R, R, device=device, generator=g)
    text = torch.randn(B, 77, kw["text_emb_dim"], device=device, generator=g)
    ones = torch.ones(B, kw["num_experts"], device=device)
    extra = dict(transition_point=-1.2, softness=1.6) if module == 2 else {}
    return dict(x0=x0, sigma=sigma, x=x, text=text, um=ones, vm=ones, extra=extra)


def conv_flops(info):
    """Algorithmic FLOPs of one conv / dgrad / wgrad launch from the REALISED per-expert row counts."""
    per_px = 2.0 * info["HW"] * info["O"] * info["I"]
    if info["seg"] is None:
        return info["N"] * per_px * info["taps"][0]
    seg = info["seg"].tolist()
    return sum((seg[g + 1] - seg[g]) * per_px * t for g, t in enumerate(info["taps"]))


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected
    offline as MI355X_MICROARCH.md prescribes: separate passes, FETCH_SIZE doubled on gfx950), or None."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    pmc_traffic.source = os.path.join("profiles", os.path.basename(files[-1]))
    tbl = json.load(open(files[-1]))
    keys = [kernel_name]
    m = re.match(r"(\w+)<__bf16, (.*)>", kernel_name)
    if m:                                                   # rocprof prints bf16 instantiations mangled: IDF16b + Li<n>E / Lb<0|1>E
        args = "".join(("Lb1E" if a == "true" else "Lb0E" if a == "false" else f"Li{a}E") for a in m.group(2).split(", ") if a.strip("-").isdigit() or a in ("true", "false"))
        keys = [f"{m.group(1)}IDF16b{args}"]
    keys = [k.split(" (")[0].replace("<split>", "") for k in keys]          # "wgrad6_kernel (+ reduce)" -> "wgrad6_kernel"
    if "bwd6_kernel" in keys:                                # the fused dgrad + weight-gradient launch of the expert layers: bwd6_kernel or, on
        keys.append("bwd7_kernel")                           # 32 x 32 / 16 x 16 maps, bwd7_kernel (the streaming programs) -- one family
    tot = n = 0.0
    for k, v in tbl.items():                                # launch-weighted mean over the instantiations of the kernel
        if any(key in k for key in keys):
            tot += (v["hbm_read_bytes_per_launch_corrected"] + v["hbm_write_bytes_per_launch"]) * v["launches"]
            n += v["launches"]
    return round(tot / n) if n else None


def replay_profile(kernel_name):
    """(average duration in us of `kernel_name` INSIDE the replayed step, launches per step of the whole step, summed kernel ms per step, file)
    from the newest committed rocprofv3 kernel trace summary (profiles/rNN_kernel_hist.txt, tools/kernel_hist.py over `rocprofv3 --kernel-trace
    --stats -- python3 bench.py ...`).  The roofline's own clock is HIP events around eager launches; this is the same kernel's duration while
    the four streams of the replayed step share the chip.  (None, ...) when no trace is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_hist.txt")))
    if not files:
        return None, None, None, None
    txt = open(files[-1]).read().split("--- by launch count")[0].splitlines()
    launches = summed = None
    m = re.match(r"\s*(\d+) launches/step, ([\d.]+) ms summed", txt[0]) if txt else None
    if m:
        launches, summed = int(m.group(1)), float(m.group(2))
    base = kernel_name.split("<")[0].split(" ")[0]
    n = t = 0.0
    for ln in txt[1:]:
        mm = re.match(r"\s*([\d.]+) x\s+([\d.]+) us =\s+([\d.]+) ms\s+(\S+)", ln)
        if mm and mm.group(4).startswith(base) or (mm and base == "bwd6_kernel" and mm.group(4).startswith("bwd7_kernel")):
            n += float(mm.group(1)); t += float(mm.group(1)) * float(mm.group(2))
    return (round(t / n, 2) if n else None), launches, summed, os.path.join("profiles", os.path.basename(files[-1]))


# Algorithmic forward MFLOP per sample at R = 32, C = 32 (torch.utils.flop_counter on the reference modules)
FWD_MFLOP = {"router": 490.8, "unet": {3: 802.1, 5: 2178.9, 7: 4244.0}, "vit": {4: 13.3, 8: 6.2, 16: 4.8, 2: 65.2}, "xattn": 142.6, "text": 21.9, "misc": 9.0}


def whole_step_flops(kw, B):
    """fwd + bwd FLOPs of one B-sample step (3 x forward), nominal even routing over the experts; None outside the table's shapes."""
    if kw["IN_img_resolution"] != 32 or kw["internal_channels"] != 32:
        return None
    try:
        fu = sum(FWD_MFLOP["unet"][k[0]] for k in kw["Unet_kernel_sizes"]) / len(kw["Unet_kernel_sizes"])
        fv = sum(FWD_MFLOP["vit"][p] for p in kw["VIT_patch_sizes"]) / len(kw["VIT_patch_sizes"])
    except KeyError:
        return None
    per_sample = 2 * FWD_MFLOP["router"] + kw["top_k"] * (fu + fv) + FWD_MFLOP["xattn"] + FWD_MFLOP["text"] + FWD_MFLOP["misc"]
    return 3.0 * per_sample * 1e6 * B


def roofline_leg(step_fn, n_steps):
    """Time every conv-family launch of a few extra steps with events on the launch stream; report the kernel
    instantiation with the largest total time.  achieved = sum(algorithmic FLOPs) / sum(duration)."""
    from hdmoe_hip import ops
    # price of the measuring stick: an event pair around a trivial launch (markers + launch latency), taken off every sample
    ops.PROFILE = []
    sc = ops.step_counter(torch.device("cuda"))
    for _ in range(64):
        ops._timed("calib", None, "hdmoe_seed_advance", sc)
    torch.cuda.synchronize()
    cal = sorted(s.elapsed_time(e) for _, _, s, e in ops.PROFILE)
    overhead = max(cal[len(cal) // 2] - 0.002, 0.0)          # ms; the trivial kernel itself runs ~2 us
    ops.PROFILE = []
    for _ in range(n_steps):
        step_fn()
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    # An event pair also spans any host-side gap while the stream is idle (tiny launches are host-latency bound), so a
    # launch's duration is taken as min(measured, median of its shape): robust against host hiccups, exact for GPU-bound launches.
    agg = {}
    shapes = {}
    per_shape = {}
    attn = {}
    for kind, info, s, e in rec:
        if kind == "attn":                                   # attention cores: exp-issue bound, not a GEMM roofline (reported separately)
            key = f"{info['dir']} B={info['B']} Sq={info['Sq']} Skv={info['Skv']} H={info['H']} D={info['D']}"
            t = attn.setdefault(key, dict(info=info, ms=[]))
            t["ms"].append(max(s.elapsed_time(e) - overhead, 0.001))
            continue
        name = info["name"] if kind == "fused" else (info["fwd_name"] if kind == "conv_fwd" else info["wgrad_name"])
        sk = f"{name} N={info['N']} HW={info['HW']} O={info['O']} I={info['I']} taps={info['taps']}"
        per_shape.setdefault(sk, []).append((name, info, max(s.elapsed_time(e) - overhead, 0.001)))
    for sk, lst in per_shape.items():
        ds = sorted(d for _, _, d in lst)
        med = ds[len(ds) // 2]
        for name, info, d in lst:
            d = min(d, med)
            sh = shapes.setdefault(sk, [0.0, 0])
            sh[0] += d; sh[1] += 1
            a = agg.setdefault(name, dict(ms=0.0, flops=0.0, n=0, dtype=info["dtype"]))
            a["ms"] += d
            a["flops"] += conv_flops(info) * info.get("mult", 1.0)     # fused launches: dgrad + wgrad (x 2), conv_res1 + conv_res2 (I = Cin + C)
            a["n"] += 1
    # v_exp_f32 issues at 8 cycles per wave-instruction (MI355X_MICROARCH.md): 1024 SIMDs x 64 lanes / 8 cycles x 2.4 GHz exps per second.
    # The forward evaluates one exp per (query, key, head) pair; so does the merged backward kernel (the two-kernel form, > 1024 queries, two).
    # (measured on these kernels with SQ_ACTIVE_INST_VALU at a 2.08 GHz clock: ~16 VALU-active cycles per v_exp_f32 wave instruction --
    #  against THAT rate the forward is at ~75 %, the backward kernels at 80-90 % of their instruction-issue floors)
    exp_peak = 1024 * 64 / 8 * 2.4e9
    # ... and the rate itself, measured: 8 independent chains of dependent v_exp_f32 per thread, 16 waves per SIMD, nothing else in the loop
    exp_meas = None
    try:
        from hdmoe_hip._lib import call as _call
        blocks, iters = 256 * 16, 4096
        buf = torch.empty(blocks * 256, dtype=torch.float32, device="cuda")
        _call("hdmoe_exp_rate", buf, blocks, 64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _call("hdmoe_exp_rate", buf, blocks, iters); e1.record()
        torch.cuda.synchronize()
        exp_meas = blocks * 256 * 8.0 * iters / (e0.elapsed_time(e1) * 1e-3)
    except Exception as ex:                                  # (an old library without the entry point)
        print(f"exp rate microbenchmark skipped: {ex}", file=sys.stderr)
    rep = []
    for key, t in sorted(attn.items(), key=lambda kv: -sum(kv[1]["ms"])):
        i = t["info"]
        ms = sorted(t["ms"])[len(t["ms"]) // 2]
        pairs = float(i["B"]) * i["H"] * i["Sq"] * i["Skv"]
        merged = i["Sq"] <= 1024 and os.environ.get("HDMOE_ATTN_BWD_MERGED", "1") != "0"       # one evaluation for dq, dk and dv (csrc/attention.hip)
        exps = pairs * (1 if (i["dir"] == "fwd" or merged) else 2)
        E = i["H"] * i["D"]
        hbm = i["B"] * E * i["esz"] * ((2 * i["Sq"] + 2 * i["Skv"]) if i["dir"] == "fwd" else (5 * i["Sq"] + 4 * i["Skv"]))
        rep.append(dict(shape=key, median_ms=round(ms, 4), launches_per_step=len(t["ms"]) / n_steps, exps_per_s=round(exps / (ms * 1e-3), 1),
                        exp_issue_bound_per_s=exp_peak, frac_of_exp_bound=round(exps / (ms * 1e-3) / exp_peak, 4),
                        exp_rate_measured_per_s=None if exp_meas is None else round(exp_meas, 1),
                        frac_of_measured_exp_rate=None if exp_meas is None else round(exps / (ms * 1e-3) / exp_meas, 4),
                        hbm_floor_us=round(hbm / 6.3e12 * 1e6, 2), frac_of_hbm_floor=round(hbm / 6.3e12 / (ms * 1e-3), 4)))
    roofline_leg.attention = rep[:4] or None
    if not agg:
        return None, {}
    def roof_of(name, a):
        peak = MFMA_PEAK[a["dtype"]]
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
        tr = pmc_traffic(name)
        return dict(bound="mfma", kernel=name, dtype=a["dtype"], achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4),
                    traffic=tr, traffic_source=getattr(pmc_traffic, "source", None) if tr is not None else None,
                    avg_launch_us=round(1e3 * a["ms"] / a["n"], 2), launches_per_step=a["n"] / n_steps, ms_per_step=round(a["ms"] / n_steps, 3))
    name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    # the expert grouped-GEMM kernels (bf16): the north_star's MFMA-utilisation target is about these, whatever kernel dominates
    bf = {k: v for k, v in agg.items() if v["dtype"] == "bfloat16" and ("conv6_bf16" in k or "conv7" in k or "wgrad6_kernel" in k or "bwd6_kernel" in k or "blk6" in k)}
    expert = None
    if bf:
        en, ea = max(bf.items(), key=lambda kv: kv[1]["ms"])
        expert = roof_of(en, ea)
        tot_ms = sum(v["ms"] for v in bf.values()); tot_fl = sum(v["flops"] for v in bf.values())
        expert["all_expert_kxk_kernels"] = dict(ms_per_step=round(tot_ms / n_steps, 3), achieved=round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                                                frac=round(tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_PEAK["bfloat16"], 4),
                                                note="time-weighted over the bf16 expert k x k launches (conv6, wgrad6, bwd6 = dgrad + wgrad, blk6 = fused block), realised routing")
    roofline_leg.expert = expert
    peak = MFMA_PEAK[a["dtype"]]
    ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
    table = {k: dict(launches_per_step=v["n"] / n_steps, avg_us=1e3 * v["ms"] / v["n"], ms_per_step=v["ms"] / n_steps,
                     tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12) for k, v in agg.items()}
    table["_by_shape_ms_per_step"] = {k: [round(v[0] / n_steps, 3), v[1] / n_steps]
                                      for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:80]}
    tr = pmc_traffic(name)
    return dict(bound="mfma", kernel=name, dtype=a["dtype"], achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4),
                traffic=tr, traffic_source=getattr(pmc_traffic, "source", None) if tr is not None else None,
                avg_launch_us=round(1e3 * a["ms"] / a["n"], 2), launches_per_step=a["n"] / n_steps, ms_per_step=round(a["ms"] / n_steps, 3),
                clock="HIP events around each EAGER launch (launch stream; kernels run one at a time)",
                method="HIP events around each launch on the launch stream, a spacer launch in front keeps host enqueue gaps out and the event-pair overhead measured on a trivial launch is subtracted; see profiles/", event_overhead_us=round(1e3 * overhead, 2)), table


def cpu_baseline(cfg_id, kw, module, seconds=9.0):
    """The CPU oracle (port of the reference algorithm) timed on the host cores on a bounded sample: B = 8 samples per
    step, TRAIN-mode (dropout 0.2, logit noise zeta 0.1) fwd + EDM loss + bwd, as many steps as fit in ~`seconds`."""
    from oracle import hdmoe_oracle as O
    import configs as C
    from models import model_config1, model_config2
    torch.manual_seed(1234)
    mod = model_config1 if module == 1 else model_config2
    model = mod.preconditioned_HDMOEM(**kw)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
            elif n.endswith("alpha_txt"):
                p.fill_(0.3)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k.split(".")[-1] not in ("freqs", "phases"))
         for k, v in model.state_dict().items()}
    cfg = dict(internal_channels=kw["internal_channels"], VIT_num_heads=kw["VIT_num_heads"], VIT_num_groups=kw["VIT_num_groups"],
               top_k=kw["top_k"], sigma_data=kw["sigma_data"])
    B = 8
    inp = make_inputs(kw, B, "cpu", 99, module)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))        # the GPU box gives one GPU a 16-core CPU share; more threads only oversubscribe
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle on {cores} threads ...", file=sys.stderr, flush=True)
    lc = C.loss_configs
    O.TRAIN.update(on=True, p=0.2, zeta=0.1)                 # the GPU step runs in train mode: so does the baseline
    times = []
    t_end = time.time() + seconds
    while time.time() < t_end or len(times) < 2:
        t0 = time.time()
        out = O.preconditioned_hdmoem(P, cfg, module, inp["x"], inp["sigma"], inp["text"], inp["um"], inp["vm"],
                                      return_log_var=True, **inp["extra"])
        loss = O.edm_loss(out, inp["x0"], kw["num_experts"], lc["unet_bal"], lc["vit_bal"], lc["z_bal"])["loss"]
        loss.backward()
        for v in P.values():
            v.grad = None
        times.append(time.time() - t0)
        print(f"[bench] cpu_baseline step {len(times)}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
        if len(times) >= 40:
            break
    med = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    # second point (n = 8 threads as well): a few more steps on 8 threads
    med8 = None
    if cores != 8:
        torch.set_num_threads(8)
        t8 = []
        t_end = time.time() + 5.0
        while time.time() < t_end or len(t8) < 3:
            t0 = time.time()
            out = O.preconditioned_hdmoem(P, cfg, module, inp["x"], inp["sigma"], inp["text"], inp["um"], inp["vm"], return_log_var=True, **inp["extra"])
            O.edm_loss(out, inp["x0"], kw["num_experts"], lc["unet_bal"], lc["vit_bal"], lc["z_bal"])["loss"].backward()
            for v in P.values():
                v.grad = None
            t8.append(time.time() - t0)
        med8 = sorted(t8[1:])[len(t8[1:]) // 2]
        torch.set_num_threads(cores)
    cpu_baseline.med8 = med8
    O.TRAIN.update(on=False)
    return med, B, cores, len(times)


def sampler_leg(device, B=128, N=40, chunk=0):
    """BASELINE configs[4] on this GPU's share of the batch (1024 images over 8 GPUs = 128 per GPU): EDM_Sampler, 2nd-order Heun, N = 40
    solver steps = 79 denoiser evaluations, 8 heterogeneous experts top-2, 4x64x64 latents, bf16, eval, the evaluation replayed as a
    hipGraph.  Independent replicas: no collective.  Returns a record for the bench line."""
    import hdmoe_hip
    import configs as C
    from EDM_sampler import EDM_Sampler
    from models import model_config2
    hdmoe_hip.set_compute_dtype(torch.bfloat16)
    kw = C.model_kwargs(**C.BASELINE_CONFIGS[4]["over"])
    torch.manual_seed(0)
    model = model_config2.preconditioned_HDMOEM(**kw).to(device).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
    g = torch.Generator(device=device).manual_seed(7)
    noise = torch.randn(B, 4, 64, 64, device=device, generator=g)
    text = torch.randn(B, 77, kw["text_emb_dim"], device=device, generator=g)
    smp = EDM_Sampler(model, Guide_net=model, guidance=1.0, num_solve_steps=N, use_graph=True)
    chunk = chunk or B                                          # B images as B / chunk consecutive sample() calls through the SAME captured graph
    with torch.no_grad():
        smp.sample(noise=noise[:chunk], text_emb=text[:chunk], transition_mean=-1.2, softness=1.2)     # warm-up + capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [smp.sample(noise=noise[i:i + chunk], text_emb=text[i:i + chunk], transition_mean=-1.2, softness=1.2) for i in range(0, B, chunk)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(o).all()) for o in outs)
    nev = (2 * N - 1) * (B // chunk)
    return dict(metric="imgs/sec, EDM_Sampler (BASELINE configs[4]" + (" per-GPU share)" if B == 128 else ", whole batch on ONE GPU)"), value=round(B / dt, 2) if ok else None,
                batch=B, chunk=chunk, solver_steps=N, denoiser_evals=nev, ms_per_eval=round(1e3 * dt / nev, 3), latents="4x64x64",
                experts="8 heterogeneous (3x3 / 5x5 / 7x7), top-2", dtype="bf16",
                launch="hipGraph replay: denoiser evaluation + the Heun update kernels of a solver stage, sigma schedule read on the device" if getattr(smp, "fused_heun", False) else "hipGraph replay of the denoiser evaluation",
                finite=ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config id (2 = headline single-GPU workload)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="skip the EDM_Sampler leg (BASELINE configs[4] shape, ~4 s)")
    ap.add_argument("--sampler-batch", type=int, default=0, help="also sample this many images on ONE GPU (BASELINE configs[4]: 1024) as chunks of 128 through one captured graph")
    ap.add_argument("--no-fp32-trunk-leg", action="store_true", help="skip the second timed leg with the three-product (fp32-equivalent) router-trunk backward")
    ap.add_argument("--dump-kernels", default="", help="write the per-kernel table of the roofline leg to this JSON file")
    ap.add_argument("--single-graph", action="store_true", help="capture the step as ONE hipGraph instead of the seven staged graphs (A/B)")
    ap.add_argument("--sync-each-step", action="store_true", help="diagnostic: synchronize after every step (the host never runs ahead of the GPU)")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python instead of replaying the captured hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # HDMOE_BENCH_BACKEND=gloo rehearses the multi-process control flow on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("HDMOE_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # HDMOE_BENCH_FORCE_DIST=1: run the one-rank job through the process group as well (RCCL with world size 1: every collective, stream
    # hand-off and launch point of the multi-GPU path executes on a single-GPU box)
    force_dist = world == 1 and os.environ.get("HDMOE_BENCH_FORCE_DIST", "0") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    multi = world > 1 or force_dist
    if multi:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    import hdmoe_hip
    from hdmoe_hip.dp import GradBuckets
    import configs as C
    import utils as U
    hdmoe_hip.lib()
    hdmoe_hip.manual_seed(4321 + rank)
    model, kw, bc = build_model(args.config, device)
    B = args.batch or bc["batch"]
    inp = make_inputs(kw, B, device, 1234 + rank, bc["module"])
    lc = C.loss_configs
    crit = U.EDM_LOSS(num_experts=kw["num_experts"], sigma_data=kw["sigma_data"], Unet_bal=lc["unet_bal"], vit_bal=lc["vit_bal"],
                      z_bal=lc["z_bal"], prior_bal=lc["prior_bal"])
    buckets = GradBuckets(model, bucket_mb=16.0, force_collectives=force_dist)            # flat fp32 grad buckets; RCCL all-reduce when world > 1
    zeta = 0.1

    from hdmoe_hip import ops
    from hdmoe_hip import graph as hgraph
    from hdmoe_hip.graph import GraphedStep, StagedStep

    def fwd_bwd():
        buckets.zero_grad()
        out = model(x=inp["x"], sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["um"], Vit_router_mask=inp["vm"],
                    zeta=zeta, return_log_var=True, **inp["extra"])
        loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
        hgraph.backward(loss["loss"])                        # loss.backward(), or the staged backward sections (StagedStep)
        return loss["loss"].detach()

    def eager_step():
        ops.advance_seed(device)
        l = fwd_bwd()
        buckets.finish()                                    # RCCL all-reduce of the flat gradient buckets when world > 1
        return {"loss": l}

    def local_step():                                       # one eager step without the gradient exchange (roofline leg, rank 0 only)
        ops.advance_seed(device)
        return {"loss": fwd_bwd()}

    step = eager_step
    graphed = None
    n_staged = 1
    if not args.no_graph:
        # the step is ~2.4k launches: replay it as one hipGraph (fwd + loss + bwd + weight-gradient finish); the gradient
        # all-reduce stays outside the graph and runs right after the replay
        buckets.enabled = False                             # no collectives from autograd hooks while capturing
        try:
            graphed = GraphedStep(fwd_bwd, device) if args.single_graph else StagedStep(fwd_bwd, device)
            n_staged = len(getattr(graphed, "graphs", {})) or 1
        except Exception as exc:                            # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches", file=sys.stderr)
            graphed = None
            args.no_graph = True
        buckets.enabled = True
        if graphed is not None:
            if hasattr(graphed, "after") and multi and os.environ.get("HDMOE_BENCH_NO_AFTER", "0") != "1":          # staged step: a branch's bucket goes to RCCL as soon as its backward is launched
                graphed.after = buckets.staged_hooks(graphed)

            def step():
                l = graphed()
                buckets.finish()
                return {"loss": l}

    trace_loss = os.environ.get("HDMOE_BENCH_TRACE_LOSS", "0") == "1"    # diagnostic: the loss of every warm-up step (one sync each)
    for _ in range(args.warmup):
        l = step()
        if trace_loss:
            print(f"[bench] warm-up loss {float(l['loss']):.6g}", file=sys.stderr)
    torch.cuda.synchronize()
    # the step builds ~4k autograd nodes; CPython's cyclic GC would stall the launch thread for ~10 ms every few steps.
    # Collect now and keep the collector off inside the timed region (tensors are freed by reference counting).
    gc.collect()
    gc.disable()
    mem0 = torch.cuda.memory_allocated()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - h0                   # host time to ENQUEUE a step (no sync inside)
        if args.sync_each_step:
            torch.cuda.synchronize()
        if trace_loss:
            print(f"[bench] timed-step loss {float(loss['loss']):.6g}", file=sys.stderr)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = 1e3 * dt / args.steps
    stage_ms = None
    if not args.no_graph and graphed is not None and hasattr(graphed, "stage_times"):
        graphed.timing = True                                # one extra, untimed step with HIP events between the staged graphs
        step()
        torch.cuda.synchronize()
        stage_ms = graphed.stage_times()
        graphed.timing = False
        if getattr(graphed, "host_us", None):                # development (HDMOE_HOST_TIMES=1): host time of each graph launch, per call
            hu = graphed.host_us
            print("host us per graph launch: " + ", ".join(f"{k}={hu[k] / hu['_calls_' + k]:.0f}" for k in hu if not k.startswith("_")), file=sys.stderr, flush=True)
    # when does each gradient bucket become final (= its backward section ends) relative to the U-Net bank's backward?  (the hooks hand a bucket
    # to the process group right behind the launch of its section: the collective starts when the section's stream gets there)
    grad_buckets = None
    if stage_ms and "unet_bwd0" in stage_ms:
        sec_of = {"unet_s3": "unet_bwd", "unet_s2": "unet_bwd2", "unet_s1": "unet_bwd1", "vit": "vit_bwd", "unet_s0": "unet_bwd0", "rest": "pre_bwd"}
        u0, u1 = stage_ms["unet_bwd"][0], stage_ms["unet_bwd0"][1]
        tot = float(buckets.nbytes())
        end_of = {t: stage_ms[sec_of[t]][1] for t in buckets.tags}
        if "vr_bwd" in stage_ms:                             # the ViT router's backward is its own section (graph.py SPLIT_VROUTER): same bucket
            end_of["vit"] = max(end_of["vit"], stage_ms["vr_bwd"][1])
        grad_buckets = {t: dict(bytes=4 * b.numel(), final_at_ms=end_of[t],
                                frac_of_unet_bwd=round((end_of[t] - u0) / max(u1 - u0, 1e-9), 3)) for t, b in zip(buckets.tags, buckets.buckets)}
        early = sum(v["bytes"] for v in grad_buckets.values() if v["frac_of_unet_bwd"] <= 0.75)
        grad_buckets["_bytes_final_before_75pct_of_unet_bwd"] = round(early / tot, 3)
    mem_growth = torch.cuda.memory_allocated() - mem0
    gc.enable()
    grads_equal = None
    if multi and os.environ.get("HDMOE_BENCH_CHECK_RANKS", "0") == "1":
        # after finish() every rank must hold the same averaged gradients, bit for bit (the replicas stay in sync without a broadcast)
        cs = torch.stack([b.double().sum() for b in buckets.buckets] + [b.double().abs().sum() for b in buckets.buckets])
        got = [torch.empty_like(cs) for _ in range(dist.get_world_size())]
        dist.all_gather(got, cs)
        grads_equal = bool(all(torch.equal(t, got[0]) for t in got)) and bool(torch.isfinite(cs).all()) and float(cs[len(buckets.buckets):].sum()) > 0.0
    loss_val = float(loss["loss"].detach())
    # sanity: EDM_LOSS clamps at 50 and random-init weights give ~4-5; anything else means the step computed garbage (this check caught a
    # hipGraph memset-node hazard in round 2) -- said loudly, and recorded in the JSON line
    import math
    loss_ok = math.isfinite(loss_val) and 0.0 < loss_val < 49.0
    if not loss_ok:
        print(f"[bench] WARNING: implausible loss {loss_val!r} after the timed steps", file=sys.stderr)

    # second leg: the same step with MaskGenerator(step=0, BW=0.3) masks instead of all-ones -- routing restricted to
    # each sample's noise band.  The masks are graph inputs: copied into the captured tensors, no re-capture.
    ms_masked = None
    try:
        mc = C.mask_configs
        sig = inp["sigma"].flatten()
        mg_u = U.MaskGenerator(expert_attributes=[k[0] for k in kw["Unet_kernel_sizes"]], p_mean=mc["p_mean"], p_std=mc["p_std"], bandwidth=mc["BW"],
                               max_bandwidth=mc["max_BW"], min_active=mc["min_active"], step_size=mc["step_size"],
                               noise_range=mc["unet_noise_range"], strat_band=mc["strat_band"])
        mg_v = U.MaskGenerator(expert_attributes=list(kw["VIT_patch_sizes"]), p_mean=mc["p_mean"], p_std=mc["p_std"], bandwidth=mc["BW"],
                               max_bandwidth=mc["max_BW"], min_active=mc["min_active"], step_size=mc["step_size"],
                               noise_range=mc["vit_noise_range"], strat_band=mc["strat_band"])
        um_save, vm_save = inp["um"].clone(), inp["vm"].clone()
        inp["um"].copy_(mg_u(sig, 0).to(device)); inp["vm"].copy_(mg_v(sig, 0).to(device))
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms_masked = 1e3 * (time.perf_counter() - tm) / args.steps
        inp["um"].copy_(um_save); inp["vm"].copy_(vm_save)
    except Exception as exc:                                  # diagnostic leg only
        print(f"[bench] masked leg failed: {type(exc).__name__}: {exc}", file=sys.stderr)
    # third leg: the to-the-letter "fp32 router" configuration -- the router trunks' BACKWARD with the three-product
    # (fp32-equivalent) split arithmetic as well (default: bf16 operands + fp32 accumulation; a 300-step A/B is in profiles/r03_trunk_bwd_precision.json)
    ms_fp32_trunk = None
    if world == 1 and not args.no_fp32_trunk_leg and not args.no_graph and graphed is not None and not args.single_graph and bc["dtype"] == "bf16":
        prev_flag = ops.TRUNK_BWD_BF16
        try:
            # (the first staged step goes first: its four prioritised streams next to a second set slow whole stages down 1.5-2x -- DESIGN.md section 5)
            graphed = None
            gc.collect()
            torch.cuda.synchronize()
            ops.TRUNK_BWD_BF16 = False
            buckets.enabled = False
            g2 = StagedStep(fwd_bwd, device)
            buckets.enabled = True
            for _ in range(3):
                g2(); buckets.finish()
            torch.cuda.synchronize()
            tm = time.perf_counter()
            for _ in range(args.steps):
                g2(); buckets.finish()
            torch.cuda.synchronize()
            ms_fp32_trunk = 1e3 * (time.perf_counter() - tm) / args.steps
            g2 = None
        except Exception as exc:                              # diagnostic leg only
            print(f"[bench] fp32-equivalent trunk-backward leg failed: {type(exc).__name__}: {exc}", file=sys.stderr)
        finally:
            ops.TRUNK_BWD_BF16 = prev_flag
            buckets.enabled = True
    roof, table = (None, {})
    cpu = None
    launch_desc = "eager" if args.no_graph else ("hipGraph replay (one graph)" if args.single_graph else f"hipGraph replay ({n_staged} staged graphs, expert branches on their own streams)")
    roof_unfused = None
    attention_rep = None
    sampler_rec = None
    if rank == 0:
        if not args.no_roofline:
            # per-launch events need eager launches; rank-local (no collective).  Leg 1: the launches of the REPLAYED step (weight-bank
            # path: dgrad + wgrad in one launch, fused Unet_block, fused router trunks) -> `roofline` = its dominant GEMM-shaped kernel.
            # Leg 2 (reference point of rounds 1-2): every layer on its own unfused kernels.
            ops.PROFILE_FUSED = True
            roof, table = roofline_leg(local_step, 3)
            expert_in_step = getattr(roofline_leg, "expert", None)
            attention_rep = getattr(roofline_leg, "attention", None)
            ops.PROFILE_FUSED = False
            roof_unfused, table_u = roofline_leg(local_step, 3)
            expert_unfused = getattr(roofline_leg, "expert", None)
            roofline_leg.expert = expert_in_step
            if roofline_leg.expert is not None and expert_unfused is not None:
                roofline_leg.expert["unfused_eager_leg"] = {k: expert_unfused[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "ms_per_step", "all_expert_kxk_kernels") if k in expert_unfused}
            if roof is not None:
                # the same kernel's clock inside the replayed step (committed rocprofv3 trace), the whole step against the bf16 MFMA peak,
                # and how many launches a replayed step is
                rep_us, launches, summed_ms, src = replay_profile(roof["kernel"])
                roof["in_replay_avg_us"] = rep_us
                roof["in_replay_frac"] = round(roof["frac"] * roof["avg_launch_us"] / rep_us, 4) if rep_us else None
                roof["in_replay_source"] = src
                wf = whole_step_flops(kw, B)
                roof["whole_step"] = None if wf is None else dict(flops=wf, achieved=round(wf / (ms * 1e-3) / 1e12, 1), unit="TFLOP/s",
                                                                  frac=round(wf / (ms * 1e-3) / 1e12 / MFMA_PEAK["bfloat16"], 4),
                                                                  note="3 x forward FLOPs (SURVEY 8(d) table, nominal even routing) / ms_per_step, against the bf16 MFMA peak")
                from hdmoe_hip import _lib as hlib
                hlib.CALL_LOG = []
                local_step()
                torch.cuda.synchronize()
                n_calls, hlib.CALL_LOG = len(hlib.CALL_LOG), None
                roof["launches_per_step"] = dict(profiled=launches, summed_kernel_ms_per_step=summed_ms, source=src, c_abi_calls_per_eager_step=n_calls)
            if args.dump_kernels:
                with open(args.dump_kernels, "w") as f:
                    json.dump({"in_step_launches": table, "unfused_launches": table_u}, f, indent=1)
        if world == 1 and not args.no_sampler and args.config == 2:
            try:
                graphed = None                                # (release the training step's graphs / pools first)
                gc.collect()
                torch.cuda.empty_cache()
                sampler_rec = sampler_leg(device)
                if args.sampler_batch:
                    sampler_rec["whole_batch_one_gpu"] = sampler_leg(device, B=args.sampler_batch, chunk=128)
            except Exception as exc:                          # a reported extra, never a reason to lose the headline line
                print(f"[bench] sampler leg failed: {type(exc).__name__}: {exc}", file=sys.stderr)
                sampler_rec = None
            hdmoe_hip.set_compute_dtype(torch.bfloat16 if bc["dtype"] == "bf16" else torch.float32)
        if world == 1 and not args.no_cpu_baseline:
            med, cb, cores, nst = cpu_baseline(args.config, kw, bc["module"])
            # metric unit: steps of B samples per second -> a CPU step of cb samples counts as cb/B of a bench step
            cpu = dict(value=round((cb / med) / B, 5), unit="denoise-steps/sec", cores=cores, kind="port",
                       sample=f"CPU oracle, fp32, train-mode (dropout, logit noise) fwd+loss+bwd on B={cb} samples/step, {cores} threads, median of {nst - 1} steps "
                              f"({med:.2f} s/step = {cb / med:.2f} samples/s), scaled to the bench's {B}-sample step"
                              + (f"; on 8 threads {cpu_baseline.med8:.2f} s/step" if getattr(cpu_baseline, "med8", None) else ""))
            if getattr(cpu_baseline, "med8", None):
                cpu["value_8_threads"] = round((cb / cpu_baseline.med8) / B, 5)
    if multi:
        dist.barrier()
    if rank == 0:
        line = {
            # (an implausible loss means the step computed garbage: no headline number then, and a non-zero exit code below)
            "metric": "denoise-steps/sec (fwd+bwd) on 4x32x32 latents", "value": round(world * 1e3 / ms, 4) if loss_ok else None,
            "unit": "denoise-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "ms_per_step_fp32_equiv_trunk_bwd": None if ms_fp32_trunk is None else round(ms_fp32_trunk, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bc["dtype"] == "bf16" else "f32",
            "data": "synthetic", "samples_per_sec": round(world * B * 1e3 / ms, 1),
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: model_config{bc['module']} preconditioned_HDMOEM, "
                                   f"{kw['IN_in_channels']}x{kw['IN_img_resolution']}x{kw['IN_img_resolution']} latents, "
                                   f"{kw['num_experts']} experts top-{kw['top_k']}, per-GPU batch {B}, train mode",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist.is_initialized() else "none",
                       "step": "fwd + EDM_LOSS + bwd"
                       + (" + RCCL grad all-reduce" if multi else ""), "launch": launch_desc, "stage_ms": stage_ms, "optimizer": "excluded (metric is fwd+bwd)",
                       "router_dtype": "f32 tensors; forward split-bf16 = fp32-equivalent (routing indices bit-exact), backward bf16 operands + fp32 accumulation", "loss": round(loss_val, 5), "loss_ok": loss_ok, "grads_equal_across_ranks": grads_equal, "hbm_growth_bytes_over_timed_region": mem_growth, "host_enqueue_ms_per_step": round(1e3 * host_s / args.steps, 3),
                       "masks": "all-ones (timed value); MaskGenerator(step=0, BW=0.3) leg: "
                                + (f"{ms_masked:.3f} ms/step" if ms_masked is not None else "n/a"), "grad_bytes": buckets.nbytes(),
                       "grad_buckets": grad_buckets},
            "roofline": roof, "roofline_expert": getattr(roofline_leg, "expert", None), "attention": attention_rep, "cpu_baseline": cpu,
            "sampler": sampler_rec,
        }
        # value = whole-job throughput: every rank runs one B-sample step per step time (weak scaling) => world / t steps/s
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()
    if not loss_ok:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
