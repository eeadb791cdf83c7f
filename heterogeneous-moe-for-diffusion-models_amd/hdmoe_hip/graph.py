"""hipGraph capture of a whole training step (forward + loss + backward + weight-gradient finish).

The step issues ~2.4k kernel launches from Python; once the kernels are fast the host becomes the bottleneck.  Every op of the
path is sync-free and allocation goes through torch's caching allocator, so the step can be captured once with
``torch.cuda.graph`` and replayed; randomness stays fresh across replays through the device-side step counter
(`ops.advance_seed`), and parameters / gradient buckets keep their addresses.
"""
from __future__ import annotations

import torch

from . import bank, ops


class GraphedStep:
    def __init__(self, step_fn, device, warmup: int = 3):
        """``step_fn()`` must be re-runnable with static inputs and write its results into persistent tensors."""
        self.device = torch.device(device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # registers the weight bank, sizes the allocator pool
                ops.advance_seed(self.device)
                step_fn()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            ops.advance_seed(self.device)
            self.out = step_fn()
            bank.join_forked_streams()                  # every forked side stream is back on the capture stream

    def __call__(self):
        self.graph.replay()
        return self.out
