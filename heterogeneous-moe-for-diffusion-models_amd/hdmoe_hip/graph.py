"""hipGraph capture of a whole training step (forward + loss + backward + weight-gradient finish).

The step issues ~2.4k kernel launches from Python; once the kernels are fast the host becomes the bottleneck.  Every op of the
path is sync-free and allocation goes through torch's caching allocator, so the step can be captured once with
``torch.cuda.graph`` and replayed; randomness stays fresh across replays through the device-side step counter
(`ops.advance_seed`), and parameters / gradient buckets keep their addresses.
"""
from __future__ import annotations

import os
import time

import torch

from . import bank, ops


def _capture_mode() -> str:
    """Error mode of the stream captures.  With a process group alive (RCCL's watchdog thread polls events from its own thread)
    a 'global' capture would turn that polling into a capture error; 'thread_local' restricts the check to the capturing thread.
    The step allocates nothing new after its warm-up runs, so the laxer mode hides nothing."""
    import os
    import torch.distributed as dist
    if os.environ.get("HDMOE_CAPTURE_MODE"):                      # (A/B aid)
        return os.environ["HDMOE_CAPTURE_MODE"]
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


import contextlib


@contextlib.contextmanager
def no_gc():
    """Python's cyclic collector must not run inside a stream capture: collecting a dead graph, stream or event of an EARLIER capture (a
    test's StagedStep, a sampler's stage graphs -- all cycles) calls hipGraphDestroy / hipEventDestroy from the capturing thread, the
    runtime refuses that inside a global-mode capture and the destructor's exception ends the process ("Fatal Python error: Aborted ...
    Garbage-collecting" in the middle of a capture, once in a few hundred runs).  Collect before, keep the collector off until the end."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


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
        with no_gc(), torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            ops.advance_seed(self.device)
            self.out = step_fn()
            bank.join_forked_streams()                  # every forked side stream is back on the capture stream

    def __call__(self):
        self.graph.replay()
        return self.out


# =====================================================================================================================================
# Staged capture: one hipGraph per independent section of the step, the two expert branches replayed on separate streams.
#
# The two expert branches of HDMOEM (router + U-Net bank / router + ViT bank) are independent between the stem and the fusion, and
# most of their kernels are far too small to fill 256 CUs.  The step is cut at those two points -- in the forward AND in the
# backward -- into seven graphs:
#
#     pre (stem, scaling, time embedding)  ->  unet | vit  ->  post (fusion, head, loss, and their backward)
#                                          ->  unet_bwd | vit_bwd  ->  pre_bwd (stem backward, weight-bank finish)
#
# and the host launches `unet` / `vit` (and the two backward graphs) on two streams, ordered by events.  Autograd is cut with
# detached leaves at the stage boundaries; each backward section is an explicit torch.autograd.backward call over the boundary
# tensors of its stage, so every section is captured on the stream its forward ran on.
#
# Measured on MI355X (BASELINE configs[1], same box, un-profiled): single stream 21.4 ms, ONE graph with the ViT bank forked onto a
# side stream 19.4 ms, staged 19.2 ms.  So a fork inside one graph already overlaps; what the staged form adds is (a) ~1 % of step
# time, (b) HIP events BETWEEN the graphs, i.e. per-stage GPU times without a profiler (`stage_times()`, bench.py "stage_ms"), and
# (c) host-visible points between the backward sections where finished gradient buckets can be handed to RCCL while the rest of the
# backward still runs.  A warning that cost hours: a rocprofv3 --kernel-trace timeline of the replay shows the branches strictly one
# after the other -- the tracer adds ~17-20 us of host time per dispatch, the replay becomes host-bound and the queues drain in
# issue order.  Branch overlap has to be read from events (stage_ms), not from the trace.
# =====================================================================================================================================
class Stager:
    """Stage bookkeeping shared by the eager warm-up runs and the capture run of a StagedStep.  Model code talks to it through
    ``graph.current()``: ``cut(stage, **tensors_by_producer_stage)`` at a boundary, ``backward(loss)`` instead of loss.backward()."""

    KIND = {"pre": "main", "ur": "r", "unet": "u", "vit": "v", "post": "main", "ucomb_bwd": "u", "ur_bwd": "r", "unet_bwd": "u", "vit_bwd": "v",
            "pre_bwd": "main", "unet_bwd2": "u", "unet_bwd1": "u", "unet_bwd0": "u", "vcomb_bwd": "v", "vr_bwd": "main"}
    # SPLIT_VROUTER (round 4): with the U-Net bank's backward down to ~6 ms the ViT section became the LAST one to finish (stage_ms: vit_bwd
    # 6.2 -> 12.8 ms, unet_bwd0 ends at 12.4): its stream runs the ViT bank's backward (~170 small launches) and then the ViT router's trunk
    # backward (the heavy part) one after the other.  The router's backward needs only the gradient of the routing weights, which the
    # combine backward -- the section's first kernel -- produces: same cut as for the U-Net router (`vcomb_bwd` on the ViT stream, then
    # `vr_bwd` beside `vit_bwd`).  The forward stays one graph.  HDMOE_VR_STREAM: the stream of `vr_bwd`.  Same box, ms/step, with the stream
    # priorities of the time: no split 13.39; "r" (behind the U-Net router's backward) 13.13; "main" (idle between `post` and `pre_bwd`) 13.32;
    # "r2" (a fifth stream of our own) 17.9 -- see the note on hardware queues below.  Without priorities (the default now, see StagedStep):
    # "main" 13.02, "r" 13.62.
    SPLIT_VROUTER = __import__("os").environ.get("HDMOE_SPLIT_VROUTER", "1") != "0"
    KIND["vr_bwd"] = __import__("os").environ.get("HDMOE_VR_STREAM", "main")
    # SPLIT_UNET_BWD (round 4): the U-Net bank's backward as up to FOUR sections on its stream -- decoder at full resolution (+ output conv),
    # decoder below, encoder below, encoder at full resolution (+ embeddings) -- cut with detached leaves inside the forward graph
    # (models/model_components.py unet_expert_bank_forward).  Each section finishes its own weight gradients (bank.finish_stage), so its
    # gradient bucket can go to RCCL while the later sections still run (hdmoe_hip/dp.py); single-process jobs just replay four graphs.
    SPLIT_UNET_BWD = __import__("os").environ.get("HDMOE_SPLIT_UNET_BWD", "1") != "0"
    # SPLIT_ROUTER: the U-Net branch is the longer one, and its router's backward (2.2 ms of kernels) needs nothing from the bank's
    # backward but the gradient of the routing weights, which the bank's FIRST backward kernel (the combine) produces.  The router
    # therefore gets its own stream and graphs (`ur`, `ur_bwd`), and the combine backward its own small graph (`ucomb_bwd`) so that
    # `ur_bwd` can start right after it, beside the bank backward.
    SPLIT_ROUTER = __import__("os").environ.get("HDMOE_SPLIT_ROUTER", "1") != "0"
    # The router FORWARD of the U-Net branch is on the critical path (the bank cannot start before the routing weights exist): it runs
    # on the bank's own (prioritised) stream; only its backward, which has slack, uses the third stream.
    if __import__("os").environ.get("HDMOE_UR_ON_U", "1") != "0":
        KIND["ur"] = "u"

    def __init__(self, device, streams, pools):
        self.device, self.streams, self.pools = device, streams, pools
        self.capture = False
        self.graphs = {}
        self.stage = None
        self._ctx = None
        import collections
        self.cuts = collections.defaultdict(list)                # producer segment -> [(tensor, detached leaf)]
        self.order = []

    # -- stage switching ------------------------------------------------------------------------------------------------------
    def begin(self, name: str):
        assert self.stage is None
        st = self.streams[self.KIND[name]]
        if self.capture:
            g = torch.cuda.CUDAGraph()
            self.graphs[name] = g
            self._ctx = torch.cuda.graph(g, pool=self.pools[self.KIND[name]], stream=st, capture_error_mode=_capture_mode())
        else:
            for other in self.streams.values():                  # warm-up: plain streams, fully ordered
                st.wait_stream(other)
            self._ctx = torch.cuda.stream(st)
        self._ctx.__enter__()
        self.stage = name
        self.order.append(name)

    def end(self):
        if self.stage is not None:
            self._ctx.__exit__(None, None, None)
            self.stage, self._ctx = None, None

    def cut(self, stage: str, **by_producer):
        """End the current stage, begin ``stage``; tensors that need a gradient are replaced by detached leaves (their gradients
        are fed to the producer stage's backward section later).  Returns the tensors in keyword order, flattened."""
        self.end()
        self.begin(stage)
        return self.cut_local(**by_producer)

    def cut_local(self, **by_producer):
        """Detached leaves without a stage switch: a boundary between two backward sections inside one forward graph."""
        out = []
        for producer, tensors in by_producer.items():
            for t in tensors:
                if torch.is_tensor(t) and t.requires_grad and not (t.is_leaf and producer.startswith("unet_c")):
                    # (inside the U-Net bank a tensor that already IS a boundary leaf stays one: no chains of leaves)
                    d = t.detach().requires_grad_(True)
                    self.cuts[producer].append((t, d))
                    out.append(d)
                else:
                    out.append(t)
        return out

    def _section(self, stage: str, producer: str):
        self.end()
        self.begin(stage)
        bank.STAGE = stage
        pairs = [(t, d.grad) for t, d in self.cuts[producer] if d.grad is not None]
        if pairs:
            torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
        if stage != "pre_bwd":
            bank.finish_stage(stage)                              # this section's weight gradients are final when its graph ends

    def backward(self, loss):
        if "post" not in self.order:
            raise RuntimeError("staged step: the model did not reach its stage boundaries (not the banked HDMOEM path)")
        bank.DEFER_FINISH = True
        try:
            bank.STAGE = "post"
            loss.backward()                                       # fusion + head + loss section; stops at the detached leaves
            if "ur" in self.order:
                self._section("ucomb_bwd", "ucomb")               # combine backward: gradients of the bank output rows and of the routing weights
                self._section("unet_bwd", "unet")
                for k in (2, 1, 0):                               # the bank's own sections, in backward order (SPLIT_UNET_BWD)
                    if self.cuts.get(f"unet_c{k}"):
                        self._section(f"unet_bwd{k}", f"unet_c{k}")
                self._section("ur_bwd", "ur")
            else:
                self._section("unet_bwd", "unet")
                for k in (2, 1, 0):
                    if self.cuts.get(f"unet_c{k}"):
                        self._section(f"unet_bwd{k}", f"unet_c{k}")
            if self.cuts.get("vcomb"):                            # SPLIT_VROUTER: combine backward, then the bank and the router side by side
                self._section("vcomb_bwd", "vcomb")
                self._section("vit_bwd", "vit")
                self._section("vr_bwd", "vr")
            else:
                self._section("vit_bwd", "vit")
            self._section("pre_bwd", "pre")
        finally:
            bank.DEFER_FINISH = False
            bank.STAGE = None
        bank.finish_pending()                                     # the remaining weight gradients (stem, fusion, head): last section


_ACTIVE = None


def current():
    """The active Stager (inside a StagedStep run) or None."""
    return _ACTIVE


def backward(loss):
    """loss.backward(), or the staged backward sections when a StagedStep is running."""
    if _ACTIVE is not None:
        _ACTIVE.backward(loss)
    else:
        loss.backward()


# Timing experiments only (tools/): sections listed in HDMOE_SKIP_STAGE are dropped from the replay -- the step's results
# are then WRONG.  Read once at import; a left-over setting is announced loudly instead of silently corrupting a training run.
SKIP_STAGES = tuple(s_ for s_ in __import__("os").environ.get("HDMOE_SKIP_STAGE", "").split(",") if s_)
if SKIP_STAGES:
    import warnings as _warnings
    _warnings.warn(f"hdmoe_hip.graph: HDMOE_SKIP_STAGE={','.join(SKIP_STAGES)} drops these sections from every StagedStep replay -- gradients and "
                   "loss are WRONG; this switch is for timing experiments only", RuntimeWarning)


class StagedStep:
    """Drop-in for GraphedStep when ``step_fn`` runs the banked HDMOEM path and calls ``graph.backward(loss)``."""

    ORDER = ["pre", "unet", "vit", "post", "unet_bwd", "vit_bwd", "pre_bwd"]
    ORDER_R = ["pre", "ur", "unet", "vit", "post", "ucomb_bwd", "unet_bwd", "ur_bwd", "vit_bwd", "pre_bwd"]

    def __init__(self, step_fn, device, warmup: int = 3):
        self.device = torch.device(device)
        # the U-Net branch is the longer one (the ViT branch has ~2.5 ms of slack in the backward): its stream gets the higher priority
        # ... but only in a single-process job.  Measured with a one-rank RCCL group (bench.py HDMOE_BENCH_FORCE_DIST=1): as soon as the
        # process owns one more stream (RCCL's) next to prioritised ones, whole stages run 1.5-2x longer (17.3 -> 22.5 ms/step; a fifth
        # stream of our own did the same); with equal priorities the extra stream costs ~0.2 ms (17.2 vs 17.0 ms/step including the
        # all-reduces).  So: priorities only when no process group exists.  How ROCclr maps streams onto its hardware queues
        # (GPU_MAX_HW_QUEUES, default 4 per priority level) decides whether two of our streams end up sharing a queue and whole
        # stages serialise: measured under the one-rank RCCL group 4 queues 17.2, 8 queues 17.4, 6 queues 20.4 ms/step; without a group
        # and without priorities 4 queues 18.0, 8 queues 17.6; a ONE-graph replay with an internal fork (the sampler) is 20 % slower
        # with 8 queues than with 4.  The default (4) is the best setting for every configuration the code itself selects.
        # ROUND 4: stream priorities are OFF by default.  With prioritised streams, replays launched back to back (the host a replay ahead of the
        # device -- what a training loop and bench.py do) showed a transient: in 2-6 % of the replays the router logits of the LOW-priority branch
        # came out with a bf16-sized error (5e-3 .. 3e-2 against 8e-5), gone in the next replay (tools/replay_race.py: 51 outliers in 900 bursts of
        # three replays; 0 in 1800 bursts without priorities, 0 with a synchronize between the replays; it survives dropping every backward section
        # from the replay, so it sits between `pre` / the forward branches of neighbouring replays).  The event dependencies are the same either
        # way; which wait the prioritised queues do not honour was not isolated.  Without priorities the schedule needs `vr_bwd` on the `main`
        # stream to keep the step time (12.98 prioritised, 13.02 equal priorities + vr_bwd on main, 13.62 equal priorities + vr_bwd behind ur_bwd).
        pmode = os.environ.get("HDMOE_STREAM_PRIO", "0")
        use_prio = pmode == "1"
        prio = {"main": -1, "u": -1, "v": 0, "r": 0} if use_prio else {"main": 0, "u": 0, "v": 0, "r": 0}
        names = ("main", "u", "v", "r") + (("r2",) if Stager.KIND["vr_bwd"] == "r2" else ())
        prio["r2"] = 0
        self.streams = {k: torch.cuda.Stream(device=self.device, priority=prio[k]) for k in names}
        self.pools = {k: torch.cuda.graph_pool_handle() for k in names}
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams.values():
            s.wait_stream(cur)
        for _ in range(warmup):
            self._run(step_fn, capture=False)
        for s in self.streams.values():
            cur.wait_stream(s)
        torch.cuda.synchronize(self.device)
        with no_gc():
            st = self._run(step_fn, capture=True)
        core = [n for n in st.order if n not in ("unet_bwd2", "unet_bwd1", "unet_bwd0", "vcomb_bwd", "vr_bwd")]
        self.split_vr = "vr_bwd" in st.order                      # the ViT router's backward as its own section (Stager.SPLIT_VROUTER)
        if core not in (self.ORDER, self.ORDER_R):
            raise RuntimeError(f"staged step: unexpected stage sequence {st.order}")
        self.split_router = core == self.ORDER_R
        self.unet_sub = [n for n in ("unet_bwd2", "unet_bwd1", "unet_bwd0") if n in st.order]    # further sections of the U-Net bank's backward
        self.graphs = st.graphs
        self._keep = st                                           # boundary tensors live in the graphs' pools
        self.after = {}                                           # {"unet_bwd" | "vit_bwd": callable}: run on that section's stream right after its launch
        self.host_us = {} if os.environ.get("HDMOE_HOST_TIMES") == "1" else None

    def _run(self, step_fn, capture: bool):
        global _ACTIVE
        st = Stager(self.device, self.streams, self.pools)
        st.capture = capture
        _ACTIVE = st
        try:
            st.begin("pre")
            ops.advance_seed(self.device)
            self.out = step_fn()
        finally:
            try:
                st.end()
            finally:
                _ACTIVE = None
        return st

    def __call__(self):
        S, g = self.streams, self.graphs
        main, u, v = S["main"], S["u"], S["v"]
        cur = torch.cuda.current_stream(self.device)
        ev = self._events = {} if self.timing else None

        skip = SKIP_STAGES

        def run(name, stream):
            if name in skip:
                if ev is not None:
                    ev[name] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[name][0].record(stream); ev[name][1].record(stream)
                return
            with torch.cuda.stream(stream):
                if ev is not None:
                    ev[name] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[name][0].record(stream)
                if self.host_us is None:
                    g[name].replay()
                else:                                             # development: host time of the graph launch itself (HDMOE_HOST_TIMES=1)
                    t0 = time.perf_counter()
                    g[name].replay()
                    self.host_us[name] = self.host_us.get(name, 0.0) + (time.perf_counter() - t0) * 1e6
                    self.host_us["_calls_" + name] = self.host_us.get("_calls_" + name, 0) + 1
                if ev is not None:
                    ev[name][1].record(stream)

        def done(name, stream):                                   # e.g. hand the section's gradient bucket to RCCL (dp.GradBuckets.launch_tag)
            fn = self.after.get(name)
            if fn is not None:
                with torch.cuda.stream(stream):
                    fn()

        if self.split_router:
            r = S["r"]
            main.wait_stream(cur)
            run("pre", main)
            ur_s = S[self._keep.KIND["ur"]]
            ur_s.wait_stream(main); v.wait_stream(main)
            run("ur", ur_s)
            run("vit", v)
            u.wait_stream(main); u.wait_stream(ur_s)
            run("unet", u)
            main.wait_stream(u); main.wait_stream(v)
            run("post", main)
            u.wait_stream(main); v.wait_stream(main)
            run("ucomb_bwd", u)
            vr_s = S[self._keep.KIND["vr_bwd"]] if self.split_vr else None
            if self.split_vr:
                run("vcomb_bwd", v)
            r.wait_stream(u)
            run("unet_bwd", u)
            run("ur_bwd", r)
            if self.split_vr:
                vr_s.wait_stream(v)                               # (behind the combine backward; on "r" also behind ur_bwd: stream order)
                run("vr_bwd", vr_s)
            run("vit_bwd", v)
            if self.split_vr:
                v.wait_stream(vr_s)                               # the "vit" gradient bucket holds the ViT router's parameters too
            if self.unet_sub:
                # the bank's sections one after the other on its stream; behind each the hook that hands its gradient bucket on
                # (buckets are ordered by completion: decoder sections first, then the ViT branch, then what ends with the backward)
                done("unet_bwd", u)
                for name in self.unet_sub[:-1]:
                    run(name, u)
                    done(name, u)
                run(self.unet_sub[-1], u)
                done("vit_bwd", v)
                u.wait_stream(r)                                  # the last U-Net bucket holds the router's parameters too
                done(self.unet_sub[-1], u)
            else:
                done("vit_bwd", v)
                u.wait_stream(r)                                  # the "unet" gradient bucket holds the router's parameters too
                done("unet_bwd", u)
            main.wait_stream(u); main.wait_stream(v)
            run("pre_bwd", main)
            cur.wait_stream(main)
            return self.out
        main.wait_stream(cur)
        run("pre", main)
        u.wait_stream(main); v.wait_stream(main)
        run("unet", u)
        run("vit", v)
        main.wait_stream(u); main.wait_stream(v)
        run("post", main)
        u.wait_stream(main); v.wait_stream(main)
        run("unet_bwd", u)
        run("vit_bwd", v)
        if self.unet_sub:
            done("unet_bwd", u)
            for name in self.unet_sub[:-1]:
                run(name, u)
                done(name, u)
            run(self.unet_sub[-1], u)
            done("vit_bwd", v)
            done(self.unet_sub[-1], u)
        else:
            done("vit_bwd", v)
            done("unet_bwd", u)
        main.wait_stream(u); main.wait_stream(v)
        run("pre_bwd", main)
        cur.wait_stream(main)
        return self.out

    timing = False

    def stage_times(self):
        """(after a call with ``timing = True`` and a synchronize) {stage: (start ms, end ms)} relative to the start of `pre`."""
        t0 = self._events["pre"][0]
        return {k: (round(t0.elapsed_time(a), 3), round(t0.elapsed_time(b), 3)) for k, (a, b) in self._events.items()}
