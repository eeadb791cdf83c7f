"""Multi-process CPU test (gloo, world_size 2) of the data-parallel gradient exchange (hdmoe_hip/dp.py).
The HDMOEM replicas share nothing but this all-reduce; the test uses a small CPU module with an
'expert' that only one rank routes to, which is the case the flat-bucket design has to get right."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.stem = torch.nn.Linear(6, 8)
        self.experts = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(3)])
        self.head = torch.nn.Linear(8, 2)

    def forward(self, x, route):
        h = torch.tanh(self.stem(x))
        out = torch.zeros_like(h)
        for e, ex in enumerate(self.experts):
            m = route == e
            if m.any():
                out = out.index_add(0, m.nonzero().flatten(), ex(h[m]))
        return self.head(out)


def _worker(rank, world, port, q, side_streams):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hdmoe_hip import ops
    from hdmoe_hip.dp import GradBuckets
    ops.SIDE_STREAMS = side_streams
    model = _Toy()
    buckets = GradBuckets(model, bucket_mb=0.0005)          # tiny buckets -> several, exercised out of order
    if side_streams:                                        # multi-stream steps: one flat bucket, reduced from finish()
        assert len(buckets.buckets) == 1 and not buckets.eager
    else:                                                   # single-stream steps: bucketed, launched from autograd hooks in order
        assert len(buckets.buckets) > 2 and buckets.eager
    assert all(p.grad is not None and p.grad.data_ptr() >= b.data_ptr() for b in buckets.buckets[:1] for p in buckets._members[0])
    g = torch.Generator().manual_seed(10 + rank)
    res = []
    for step in range(2):
        x = torch.randn(5, 6, generator=g)
        route = torch.tensor([0, 0, 1, 0, 1]) if rank == 0 else torch.tensor([0, 1, 1, 1, 0])   # expert 2: nobody; all use 0/1
        if step == 1 and rank == 1:
            route = torch.zeros(5, dtype=torch.long)                                             # expert 1 unused on rank 1 only
        buckets.zero_grad()
        model(x, route).square().sum().backward()
        buckets.finish()
        res.append({n: p.grad.clone() for n, p in model.named_parameters()})
        # single-process reference: same module, grads of both ranks' batches averaged
        ref = _Toy()
        acc = {n: torch.zeros_like(p) for n, p in ref.named_parameters()}
        for r in range(world):
            gr = torch.Generator().manual_seed(10 + r)
            for s in range(step + 1):
                xr = torch.randn(5, 6, generator=gr)
            rr = torch.tensor([0, 0, 1, 0, 1]) if r == 0 else torch.tensor([0, 1, 1, 1, 0])
            if step == 1 and r == 1:
                rr = torch.zeros(5, dtype=torch.long)
            ref.zero_grad()
            ref(xr, rr).square().sum().backward()
            for n, p in ref.named_parameters():
                if p.grad is not None:
                    acc[n] += p.grad / world
        for n in acc:
            torch.testing.assert_close(res[-1][n], acc[n], rtol=1e-5, atol=1e-6)
        assert float(res[-1]["experts.2.weight"].abs().max()) == 0.0      # never-routed expert: exact zeros, same layout on all ranks
    q.put((rank, "ok"))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("side_streams", [False, True])
def test_grad_buckets_world2_gloo(side_streams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, side_streams)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


# ---------------------------------------------------------------------------------------------------------------------------------
# The REAL parameter layout: GradBuckets over preconditioned_HDMOEM (CPU-constructed; the HIP kernels are not involved -- gradients
# are synthetic, written straight into the bucket views the way autograd / the weight bank write them on the GPU).
def _hdmoem_worker(rank, world, port, q, side_streams):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "heterogeneous-moe-for-diffusion-models_amd"))
    from hdmoe_hip import ops
    from hdmoe_hip.dp import GradBuckets
    from models import model_config2
    ops.SIDE_STREAMS = side_streams
    torch.manual_seed(1)
    cfg = dict(IN_in_channels=4, IN_img_resolution=16, internal_channels=8, time_emb_dim=16, text_emb_dim=32, num_experts=4, top_k=2,
               Fourier_bandwidth=1.0, VIT_num_blocks=1, VIT_patch_sizes=[2, 4, 4, 8], VIT_num_groups=2, VIT_num_heads=2, VIT_emb_size=8,
               Unet_num_blocks=1, Unet_channel_mult=[1, 2], Unet_kernel_sizes=[(3, 3), (3, 3), (5, 5), (5, 5)], Unet_model_channels=8,
               Unet_channel_mult_emb=2, sigma_data=0.5, log_var_channels=8)
    model = model_config2.preconditioned_HDMOEM(**cfg)
    names = [n for n, _ in model.named_parameters()]
    buckets = GradBuckets(model, bucket_mb=0.25)
    assert buckets.nbytes() == 4 * sum(p.numel() for p in model.parameters())
    if side_streams:                                                 # one bucket per section of the staged step, in completion order
        assert buckets.tags == ["unet_s3", "unet_s2", "unet_s1", "vit", "unet_s0", "rest"] and not buckets.eager and buckets.top == "16x16"
        for tag, members in zip(buckets.tags, buckets._members):
            owned = {id(p) for p in members}
            assert all((id(p) in owned) == (GradBuckets.tag_of(n, buckets.top) == tag) for n, p in model.named_parameters())
        T = lambda n: GradBuckets.tag_of(n, "16x16")
        assert T("net.VIT_experts.0.patch.weight") == "vit" and T("net.Unet_router.linear.weights") == "unet_s0"
        assert T("net.input_proj.weights") == "rest" and T("log_var_linear.weights") == "rest"
        assert T("net.Unet_experts.1.decoders.16x16_block0.conv_res1.weights") == "unet_s3" and T("net.Unet_experts.1.out_conv.weights") == "unet_s3"
        assert T("net.Unet_experts.1.out_gain") == "unet_s3" and T("net.Unet_experts.0.decoders.8x8_in0.conv_res1.weights") == "unet_s2"
        assert T("net.Unet_experts.3.encoders.8x8_block0.conv_res2.weights") == "unet_s1" and T("net.Unet_experts.3.encoders.16x16_conv.weights") == "unet_s0"
        assert T("net.Unet_experts.2.decoders.16x16_block0.emb_layer.weights") == "unet_s0" and T("net.Unet_experts.2.map_noise.weights") == "unet_s0"
        # the decoder sections -- handed to the process group while the encoder sections still run -- hold most of the U-Net experts' bytes
        nb = {t: sum(p.numel() for p in m) for t, m in zip(buckets.tags, buckets._members)}
        unet_all = sum(v for t, v in nb.items() if t.startswith("unet"))
        assert (nb["unet_s3"] + nb["unet_s2"]) > 0.55 * unet_all, nb
    else:
        assert len(buckets.buckets) >= 3 and buckets.eager
    # every parameter's .grad is a view into exactly one bucket, in reverse registration order
    spans = sorted((p.grad.data_ptr(), p.grad.numel()) for p in model.parameters())
    for (a0, n0), (a1, _) in zip(spans, spans[1:]):
        assert a0 + 4 * n0 <= a1
    unused = "net.Unet_experts.2." if rank == 1 else None            # rank 1 routes nothing to U-Net expert 2
    for step in range(2):
        buckets.zero_grad()
        g = torch.Generator().manual_seed(100 * step + rank)
        mine = {}
        # gradients land in reverse registration order, like backward; the unused expert's slice is never touched
        for n, p in reversed(list(model.named_parameters())):
            if unused and n.startswith(unused):
                mine[n] = torch.zeros_like(p)
                continue
            mine[n] = torch.randn(p.shape, generator=g)
            p.grad.add_(mine[n])                                     # AccumulateGrad into the bucket view
            for h in (p._post_accumulate_grad_hooks or {}).values():  # what autograd calls after accumulating (hooks exist in the one-stream mode only)
                h(p)
        if side_streams:                                             # what StagedStep's hooks do behind the backward sections, in its order
            for tag in ("unet_s3", "unet_s2", "unet_s1", "vit", "unet_s0"):
                buckets.launch_tag(tag)
            assert sum(w is not None for w in buckets._works) == 5
        buckets.finish()
        # reference: average of both ranks' synthetic gradients
        for n, p in model.named_parameters():
            ref = torch.zeros_like(p)
            for r in range(world):
                gr = torch.Generator().manual_seed(100 * step + r)
                vals = {}
                for n2, p2 in reversed(list(model.named_parameters())):
                    if r == 1 and n2.startswith("net.Unet_experts.2."):
                        vals[n2] = torch.zeros_like(p2)
                    else:
                        vals[n2] = torch.randn(p2.shape, generator=gr)
                ref += vals[n] / world
            torch.testing.assert_close(p.grad, ref, rtol=1e-6, atol=1e-7)
    q.put((rank, "ok", len(names)))
    dist.destroy_process_group()


@pytest.mark.parametrize("side_streams", [False, True])
def test_grad_buckets_over_the_real_hdmoem_parameter_list(side_streams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hdmoem_worker, args=(r, 2, port, q, side_streams)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(2)]
    assert sorted(r for r, _, _ in got) == [0, 1] and got[0][2] == got[1][2] > 150
