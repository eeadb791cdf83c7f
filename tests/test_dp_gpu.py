"""Two ranks on ONE MI355X (gloo over CUDA tensors: the box has a single GPU, RCCL needs one device per rank): the default training
configuration -- staged hipGraph step, a branch's gradient bucket handed to the process group right after that branch's backward
graph is launched, the rest from finish() -- against the single-process average of the two ranks' gradients."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, backend="gloo", bf16=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "heterogeneous-moe-for-diffusion-models_amd")
    for p in (pkg, os.path.join(pkg, "Utils"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    if backend == "nccl":                                          # RCCL; one rank per device, so world == 1 on this box
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import hdmoe_hip
    from hdmoe_hip import ops, graph as hgraph
    from hdmoe_hip.dp import GradBuckets
    from Utils import configs
    from Utils.utils import EDM_LOSS
    from models import model_config1
    from oracle.recipe import fill_state, make_inputs
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    hdmoe_hip.set_compute_dtype(torch.bfloat16 if bf16 else torch.float32)
    tol = 2e-2 if bf16 else 3e-4               # bf16: the reference side accumulates two batches through one replica (other bf16 roundings)
    kw = configs.model_kwargs(**configs.BASELINE_CONFIGS[2]["over"])
    crit = EDM_LOSS(num_experts=4, sigma_data=0.5, Unet_bal=0.05, vit_bal=0.1, z_bal=0.005, prior_bal=0.0)

    def build():
        m = model_config1.preconditioned_HDMOEM(**kw)
        m.load_state_dict(fill_state(m.state_dict(), 5))
        return m.to(dev).eval()

    def inputs(r):
        return {k: v.to(dev) for k, v in make_inputs(6, 4, 32, 4, 77, 768, 20 + r).items()}

    def step_fn(model, inp, zero):
        def f():
            zero()
            out = model(x=inp["x"], sigma=inp["sigma"], text_emb=inp["text"], Unet_router_mask=inp["unet_mask"],
                        Vit_router_mask=inp["vit_mask"], zeta=0.0, return_log_var=True)
            loss = crit(sigma_vec=inp["sigma"], x=inp["x0"], sigma=inp["sigma"], out_model=out)
            hgraph.backward(loss["loss"])
            return loss["loss"].detach()
        return f

    # single-process reference: both ranks' batches through one replica, gradients averaged (plain eager, one stream)
    saved = ops.SIDE_STREAMS
    ops.SIDE_STREAMS = False
    ref_model = build()
    acc = {}
    for r in range(world):
        f = step_fn(ref_model, inputs(r), lambda: ref_model.zero_grad(set_to_none=False))
        f(); f()                                                   # second pass runs through the weight bank
        torch.cuda.synchronize()
        for n, p in ref_model.named_parameters():
            if p.grad is not None:
                acc[n] = acc.get(n, 0) + p.grad.detach().clone() / world
    ops.SIDE_STREAMS = saved
    del ref_model

    model = build()
    buckets = GradBuckets(model, force_collectives=world == 1)
    assert buckets.tags == ["unet_s3", "unet_s2", "unet_s1", "vit", "unet_s0", "rest"]
    staged = hgraph.StagedStep(step_fn(model, inputs(rank), buckets.zero_grad), dev, warmup=2)
    assert staged.unet_sub == ["unet_bwd2", "unet_bwd1", "unet_bwd0"]     # the U-Net bank's backward runs as four sections
    staged.after = buckets.staged_hooks(staged)
    for _ in range(2):
        staged()
        assert sum(w is not None for w in buckets._works) == 5     # every branch bucket went out before finish()
        buckets.finish()
    torch.cuda.synchronize()
    bad = []
    for n, p in model.named_parameters():
        if n in acc:
            scale = float(acc[n].abs().max())
            err = float((p.grad - acc[n]).abs().max())
            if err > tol * scale + 1e-7:
                bad.append((n, err, scale))
    q.put((rank, bad[:5], len(acc)))
    dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_staged_step_with_bucket_overlap_two_ranks_on_one_gpu(bf16):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "gloo", bf16)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    for rank, bad, n in got:
        assert not bad, (rank, bad)
        assert n > 400


def test_staged_step_through_rccl_with_one_rank():
    """The RCCL branch of the gradient exchange on hardware: a one-rank "nccl" group (device_id init, ReduceOp.AVG, collectives handed
    off from the staged step's branch streams between graph replays); the averaged gradients of one rank are that rank's gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), q, "nccl"))
    p.start()
    p.join(420)
    assert p.exitcode == 0
    rank, bad, n = q.get(timeout=10)
    assert not bad, bad
    assert n > 400


def _sync_worker(rank, world, port, q):
    """Three train()-mode iterations of the Trainer (per-rank batches, per-rank dropout / noise seeds, flat-bucket gradient exchange,
    fused clip + AdamW with the expert-usage skip): the replicas must stay bit-identical without any parameter broadcast -- the
    in-forward weight re-normalisation (reference model_internals.py:254-256) is a deterministic function of identical weights, the
    averaged gradients are identical on every rank, and the usage flags are summed over the ranks."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "heterogeneous-moe-for-diffusion-models_amd")
    for p in (pkg, os.path.join(pkg, "Utils"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hdmoe_hip
    from Utils import configs, training
    from models import model_config2 as model_config1           # (the reference's training loop drives the config2 signature)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    hdmoe_hip.set_compute_dtype(torch.bfloat16)
    hdmoe_hip.manual_seed(4321 + rank)                              # dropout masks / logit noise differ per rank, as in bench.py
    kw = configs.model_kwargs(**configs.BASELINE_CONFIGS[2]["over"])
    torch.manual_seed(1234)                                         # identical replicas
    model = model_config1.preconditioned_HDMOEM(**kw)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
    model = model.to(dev).train()
    mcfg = dict(configs.model_configs, **configs.BASELINE_CONFIGS[2]["over"], total_steps=10)
    tr = training.Trainer(model, mcfg, configs.optim_configs, configs.loss_configs, configs.mask_configs, configs.zeta_configs)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)       # per-rank data
    torch.manual_seed(7 + rank)                                     # per-rank sigma / noise draws inside train_step
    R, C = kw["IN_img_resolution"], kw["IN_in_channels"]
    for _ in range(3):
        tr.train_step(0.5 * torch.randn(6, C, R, R, device=dev, generator=gen), torch.randn(6, 77, kw["text_emb_dim"], device=dev, generator=gen))
    torch.cuda.synchronize()
    cs = torch.stack([p.detach().double().sum() for p in model.parameters()] + [p.detach().double().abs().sum() for p in model.parameters()])
    got = [torch.empty_like(cs) for _ in range(world)]
    dist.all_gather(got, cs)
    same = all(torch.equal(t, got[0]) for t in got)
    moved = float((cs - q_init(model_config1, kw, dev)).abs().sum()) > 0.0
    q.put((rank, bool(same), bool(moved), bool(torch.isfinite(cs).all())))
    dist.destroy_process_group()


def q_init(mod, kw, dev):
    torch.manual_seed(1234)
    m0 = mod.preconditioned_HDMOEM(**kw)
    with torch.no_grad():
        for n, p in m0.named_parameters():
            if n.endswith("out_gain"):
                p.fill_(0.5)
    m0 = m0.to(dev)
    return torch.stack([p.detach().double().sum() for p in m0.parameters()] + [p.detach().double().abs().sum() for p in m0.parameters()])


def test_replicas_stay_bit_identical_over_three_optimizer_steps():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0
    for rank, same, moved, finite in sorted(q.get(timeout=10) for _ in range(2)):
        assert same, f"rank {rank}: parameters differ between the replicas"
        assert moved and finite


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """`python -m torch.distributed.run ... bench.py --gpus 2`, the command the driver uses for the scaling runs, rehearsed with two
    ranks sharing the one GPU of this box (HDMOE_BENCH_BACKEND=gloo; RCCL needs a device per rank): rendezvous, per-rank seeds, staged
    step with the bucket hand-off, barrier + max-over-ranks timing, ONE JSON line from rank 0."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HDMOE_BENCH_BACKEND="gloo", HDMOE_BENCH_CHECK_RANKS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "32",
           "--no-roofline", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["world_size"] == 2 and rec["config"]["global_batch"] == 64 and rec["scaling"] == "weak"
    assert rec["config"]["loss_ok"] and rec["value"] is not None and rec["value"] > 0
    assert rec["config"]["grads_equal_across_ranks"] is True
    assert "staged graphs" in rec["config"]["launch"]
