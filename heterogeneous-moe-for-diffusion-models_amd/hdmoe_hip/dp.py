"""Data-parallel gradient exchange for the HDMOEM replicas: one process per GPU, flat fp32 gradient buckets whose
views ARE the parameters' ``.grad`` tensors, one RCCL all-reduce(avg) per bucket over xGMI.

Sample routing is per-sample and every expert is replicated, so the path has no data-path collective: the only exchange
is this gradient all-reduce.  An expert that received no sample on a rank contributes exact
zeros (its bucket slice was memset and never written) so every rank reduces identical layouts.

When do the collectives go out?  Almost every gradient of this path is written straight into its bucket view by a kernel (the
weight bank's finish launch, the norm / bias / table backward kernels) and never passes through an AccumulateGrad node, so
autograd hooks cannot tell when a bucket is complete.  Completion is known structurally instead:
  * default (multi-stream step): the parameters are bucketed by the SECTION of the staged step (hdmoe_hip/graph.py) whose backward
    finishes them -- "vit" (ViT router + experts), "unet" (U-Net router + experts), "rest" (stem, fusion, head, preconditioning).
    ``launch_tag("vit")`` / ``launch_tag("unet")`` are called by StagedStep right after the section's backward graph has been
    launched, on that section's stream: the all-reduce then runs beside the remaining sections.  ``finish()`` sends what is left
    and waits.  Without a StagedStep everything goes out from ``finish()`` (correct, not overlapped).
  * HDMOE_SIDE_STREAMS=0 (one stream): 16 MB buckets in reverse registration order; hooks launch the complete prefix for the few
    gradients that do come through autograd, the rest again from ``finish()``.
The order of collectives is the same on every rank by construction (tags in a fixed order / bucket index order).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradBuckets:
    TAGS = ("vit", "unet", "rest")

    @staticmethod
    def tag_of(name: str) -> str:
        """Section of the staged step that completes the parameter's gradient (HDMOEM attribute names, models/_assembly.py)."""
        if ".vit_router." in "." + name or ".VIT_experts." in "." + name:
            return "vit"
        if ".Unet_router." in "." + name or ".Unet_experts." in "." + name:
            return "unet"
        return "rest"

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 16.0, process_group=None, force_collectives: bool = False):
        """``force_collectives``: issue the all-reduces even in a one-rank group (a single-GPU box then exercises the RCCL path:
        stream hand-off, ReduceOp.AVG, the staged step's launch points)."""
        self.group = process_group
        self.force = bool(force_collectives)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        named.reverse()                                            # backward produces gradients roughly in this order
        from . import ops as _ops
        self.buckets: List[torch.Tensor] = []
        self._members: List[List[torch.nn.Parameter]] = []
        self.tags: List[str] = []
        if _ops.SIDE_STREAMS:
            # one flat bucket per section of the staged step (sections complete as a whole: see the module docstring)
            for tag in self.TAGS:
                members = [p for n, p in named if self.tag_of(n) == tag]
                if members:
                    self._seal(members, sum(p.numel() for p in members))
                    self.tags.append(tag)
        else:
            cap = int(bucket_mb * (1 << 20) / 4)
            cur, n = [], 0
            for _, p in named:
                if cur and n + p.numel() > cap:
                    self._seal(cur, n)
                    cur, n = [], 0
                cur.append(p)
                n += p.numel()
            if cur:
                self._seal(cur, n)
            self.tags = ["-"] * len(self.buckets)
        # Per-expert routed-row counts of the step (written by the model's forward, read by FusedAdamW to skip experts without samples):
        # one small tensor for all expert lists, summed over the ranks behind the last gradient bucket -- an expert is skipped only when no
        # rank routed a sample to it.
        self.usage = None
        lists = [mod for name, mod in module.named_modules()
                 if isinstance(mod, torch.nn.ModuleList) and name.rsplit(".", 1)[-1] in ("Unet_experts", "VIT_experts") and len(mod)]
        if lists and self.buckets:
            self.usage = torch.zeros(sum(len(l) for l in lists), dtype=torch.float32, device=self.buckets[0].device)
            off = 0
            for l in lists:
                object.__setattr__(l, "_hdmoe_usage", self.usage[off:off + len(l)])
                off += len(l)
        self._usage_work = None
        self._pending = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._next = 0                                             # collectives are issued strictly in bucket order
        self.enabled = True                                        # False: hooks only count (e.g. while capturing a hipGraph)
        # Hooks fire on whatever stream the gradient was produced on.  With the ViT experts forked onto side streams a bucket can
        # be completed on one stream while another still writes a neighbouring slice, so eager launches are only safe when the
        # whole step runs on one stream; otherwise every bucket goes out in finish(), after autograd has joined the streams.
        self.eager = not _ops.SIDE_STREAMS
        self._backend = dist.get_backend(process_group) if dist.is_initialized() else None
        if self.eager:                                             # (a hook keeps the parameter's AccumulateGrad node -- and the stream it was
            for bi, members in enumerate(self._members):           #  created on -- alive across steps; the section mode needs none)
                for p in members:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _seal(self, members, n):
        p0 = members[0]
        flat = torch.zeros(n, dtype=torch.float32, device=p0.device)
        off = 0
        for p in members:
            p.grad = flat[off:off + p.numel()].view_as(p)           # gradient_as_bucket_view
            off += p.numel()
        self.buckets.append(flat)
        self._members.append(list(members))

    def _make_hook(self, bi):
        def hook(_p):
            if not self.enabled:
                return
            self._pending[bi] += 1
            if not self.eager:
                return
            # every rank must issue the same collectives in the same order, but which hooks fire (and when) depends on
            # the local routing: launch only the contiguous prefix of complete buckets, the rest waits for finish()
            while self._next < len(self.buckets) and self._pending[self._next] == len(self._members[self._next]):
                self._launch(self._next)
                self._next += 1
        return hook

    def _launch(self, bi):
        if (self.world == 1 and not self.force) or self._works[bi] is not None:
            return
        if self._backend == "nccl":                               # "nccl" is RCCL on ROCm
            self._works[bi] = dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:                                                       # gloo (CPU tests): no AVG
            self._works[bi] = dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def launch_tag(self, tag: str):
        """The section ``tag`` of the staged step has been launched on the current stream: send its bucket.  Collectives are issued
        strictly in bucket order on every rank, so an earlier bucket that is still waiting goes first (it is complete as well:
        sections are launched in bucket order)."""
        if not self.enabled or tag not in self.tags:
            return
        bi = self.tags.index(tag)
        while self._next <= bi:
            self._launch(self._next)
            self._next += 1

    def finish(self):
        """Call after backward: reduce every bucket that has not gone out yet, wait for all."""
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        self._next = 0
        if self.usage is not None and (self.world > 1 or self.force):
            dist.all_reduce(self.usage, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
        for bi, w in enumerate(self._works):
            if w is not None:
                w.wait()
                if self._backend != "nccl":
                    self.buckets[bi].div_(self.world)
            self._works[bi] = None
            self._pending[bi] = 0

    def zero_grad(self):
        for flat in self.buckets:
            flat.zero_()

    def nbytes(self) -> int:
        return sum(b.numel() for b in self.buckets) * 4
