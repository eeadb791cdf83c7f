"""Data-parallel gradient exchange for the HDMOEM replicas: one process per GPU, flat fp32 gradient buckets whose
views ARE the parameters' ``.grad`` tensors, one RCCL all-reduce(avg) per bucket over xGMI.

Sample routing is per-sample and every expert is replicated, so the path has no data-path collective: the only exchange
is this gradient all-reduce.  An expert that received no sample on a rank contributes exact
zeros (its bucket slice was memset and never written) so every rank reduces identical layouts.

When do the collectives go out?  Almost every gradient of this path is written straight into its bucket view by a kernel (the
weight bank's finish launch, the norm / bias / table backward kernels) and never passes through an AccumulateGrad node, so
autograd hooks cannot tell when a bucket is complete.  Completion is known structurally instead:
  * default (multi-stream step): the parameters are bucketed by the SECTION of the staged step (hdmoe_hip/graph.py) whose backward
    finishes them, in the order the sections complete: the U-Net bank's backward runs as four sections (graph.Stager.SPLIT_UNET_BWD) --
    "unet_s3" (full-resolution decoder entries + output conv), "unet_s2" (decoder below), "unet_s1" (encoder below), then "vit" (ViT
    router + experts), "unet_s0" (full-resolution encoder entries, the embedding layers of every block, the U-Net router) and "rest" (stem,
    fusion, head, preconditioning).  ``staged_hooks(staged)`` gives StagedStep one hook per section; each runs right after that section's
    backward graph has been launched, on that section's stream, and hands ITS bucket to the process group: the all-reduce of the decoder
    gradients (two thirds of the U-Net experts' parameters) runs beside the encoder sections.  RCCL executes a group's collectives in
    issue order, hence buckets ordered by completion.  ``finish()`` sends what is left and waits.  Without a StagedStep everything goes out
    from ``finish()`` (correct, not overlapped).
  * HDMOE_SIDE_STREAMS=0 (one stream): 16 MB buckets in reverse registration order; hooks launch the complete prefix for the few
    gradients that do come through autograd, the rest again from ``finish()``.
The order of collectives is the same on every rank by construction (tags in a fixed order / bucket index order).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradBuckets:
    TAGS = ("unet_s3", "unet_s2", "unet_s1", "vit", "unet_s0", "rest")     # completion order of the staged backward

    @staticmethod
    def tag_of(name: str, top: str = "") -> str:
        """Section of the staged step that completes the parameter's gradient (HDMOEM attribute names, models/_assembly.py; ``top`` = the
        full-resolution level prefix of the U-Net experts, e.g. "32x32")."""
        n = "." + name
        if ".vit_router." in n or ".VIT_experts." in n:
            return "vit"
        if ".Unet_router." in n:
            return "unet_s0"                                       # (its backward runs beside the bank's, on a third stream: last U-Net bucket)
        if ".Unet_experts." in n:
            if ".emb_layer." in n or ".map_noise." in n or ".map_text." in n:
                return "unet_s0"                                   # one multi-tensor backward for all blocks' embedding layers, in the last section
            if ".decoders." in n:
                return "unet_s3" if top and f".decoders.{top}_" in n else "unet_s2"
            if ".out_conv." in n or n.endswith(".out_gain"):
                return "unet_s3"
            if ".encoders." in n:
                return "unet_s0" if (not top or f".encoders.{top}_" in n) else "unet_s1"
            return "unet_s0"
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
        top = ""
        for name, mod in module.named_modules():
            if name.endswith("Unet_experts.0") and hasattr(mod, "encoders") and len(mod.encoders):
                top = next(iter(mod.encoders.keys())).split("_")[0]
                break
        self.top = top
        # ONE allocation behind all buckets (16-byte aligned slices): zero_grad is a single fill, whatever the number of buckets
        total = sum((p.numel() + 3) // 4 * 4 for _, p in named) + 4 * (len(self.TAGS) + len(named) // 1 * 0)
        self._master = torch.zeros(total, dtype=torch.float32, device=named[0][1].device) if named else None
        self._moff = 0
        if _ops.SIDE_STREAMS:
            # one flat bucket per section of the staged step (sections complete as a whole: see the module docstring)
            for tag in self.TAGS:
                members = [p for n, p in named if self.tag_of(n, top) == tag]
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
        start = (self._moff + 3) // 4 * 4
        if self._master is not None and start + n <= self._master.numel():
            flat = self._master[start:start + n]
            self._moff = start + n
        else:                                                      # (cannot happen with the sizing above; keeps the class usable if it ever does)
            flat = torch.zeros(n, dtype=torch.float32, device=members[0].device)
            self._master = None
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
        """The section ``tag`` of the staged step has been launched on the current stream: send its bucket (and only its bucket -- an earlier
        bucket may belong to a section on another stream).  Every rank runs the same staged step, so the collectives are issued in the
        same order everywhere."""
        if not self.enabled or tag not in self.tags:
            return
        self._launch(self.tags.index(tag))

    def staged_hooks(self, staged) -> dict:
        """{backward section of ``staged`` (hdmoe_hip.graph.StagedStep): hook}: each hook runs behind the launch of its section, on the
        section's stream, and hands that section's gradient bucket to the process group."""
        L = self.launch_tag
        if len(getattr(staged, "unet_sub", [])) == 3:
            return {"unet_bwd": lambda: L("unet_s3"), "unet_bwd2": lambda: L("unet_s2"), "unet_bwd1": lambda: L("unet_s1"),
                    "vit_bwd": lambda: L("vit"), "unet_bwd0": lambda: L("unet_s0")}
        # the U-Net bank's backward as ONE section: all of its buckets behind it
        return {"vit_bwd": lambda: L("vit"), "unet_bwd": lambda: [L(t) for t in ("unet_s3", "unet_s2", "unet_s1", "unet_s0")]}

    def finish(self):
        """Call after backward: reduce every bucket that has not gone out yet, wait for all."""
        for bi in range(len(self.buckets)):                        # (_launch skips what a hook has already sent)
            self._launch(bi)
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
        if self._master is not None:
            self._master.zero_()
            return
        for flat in self.buckets:
            flat.zero_()

    def nbytes(self) -> int:
        return sum(b.numel() for b in self.buckets) * 4
