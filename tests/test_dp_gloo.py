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
