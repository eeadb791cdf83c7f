"""Heun 2nd-order EDM sampler -- drop-in for the reference's ``Utils/EDM_sampler.py`` (row N1 of SURVEY.md section 8(f)).

Same constructor / ``denoise`` / ``sample`` signatures and semantics (Karras rho-schedule, optional churn, CFG lerp).  The
2N-1 model evaluations run through the HIP path; the per-step latent updates are fused launches (no torch arithmetic on the latents), and
the denoiser call is sync-free (device-side dispatch plan, no ``mask.any()``).  With ``use_graph=True`` (and no churn) a WHOLE Heun stage --
both evaluations, the Euler step and the 2nd-order correction, sigma taken from a device-side copy of the schedule -- is one captured
hipGraph replayed N - 1 times (plus one for the last, Euler-only stage): no host arithmetic between evaluations.  With churn the
evaluation alone is captured, as before.
"""
import numpy as np
import torch
import torch.nn as nn

from hdmoe_hip import graph as hgraph
from hdmoe_hip import ops


class EDM_Sampler:
    def __init__(self, model: nn.Module, Guide_net: nn.Module, num_solve_steps: int = 32, sigma_min: float = 0.002,
                 sigma_max: float = 80, rho: int = 7, S_churn: float = 0.0, S_min: float = 0.0, S_max: float = float("inf"),
                 S_noise: float = 1.0, guidance: float = 1.0, dtype=torch.float32, use_graph: bool = False):
        self.model = model
        self.gnet = Guide_net
        self.num_steps = num_solve_steps
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.rho = rho
        self.s_churn = S_churn
        self.s_min = S_min
        self.s_max = S_max
        self.s_noise = S_noise
        self.guide = guidance
        self.dtype = dtype
        self.use_graph = use_graph          # extension over the reference: hipGraph replay of the denoiser evaluation
        self._graph = None
        self._gkey = None
        # no churn, fp32 latents: a solver stage = both denoiser evaluations + the fused Euler / Heun-correction kernels with sigma read from a
        # device-side schedule; with use_graph it is ONE captured graph, replayed N - 1 times, plus one graph for the last (Euler-only) stage
        self.fused_heun = False
        self._stage = None
        self._skey = None

    # reference Utils/EDM_sampler.py:35-70
    def denoise(self, x, sigma, text_emb, transition_mean, softness, uncond_text_emb=None):
        bs = x.shape[0]
        num_experts = self.model.num_experts
        Unet_router_mask = torch.ones((bs, num_experts), device=x.device)
        vit_router_mask = torch.ones((bs, num_experts), device=x.device)
        kw = dict(x=x, sigma=sigma, Unet_router_mask=Unet_router_mask, Vit_router_mask=vit_router_mask, zeta=0,
                  transition_point=transition_mean, softness=softness)
        D_x = self.model(text_emb=text_emb, **kw)["denoised"].to(self.dtype)
        if self.guide == 1.0:
            return D_x
        emb_for_guide = uncond_text_emb if uncond_text_emb is not None else text_emb
        ref_D_x = self.gnet(text_emb=emb_for_guide, **kw)["denoised"].to(self.dtype)
        # ref.lerp(D, g) = (1-g)*ref + g*D
        return ops.axpby(ref_D_x, D_x, 1.0 - self.guide, self.guide)

    # ---- hipGraph path ---------------------------------------------------------------------------------------------
    def _denoise_graphed(self, x, t, text_emb, transition_mean, softness, uncond_text_emb):
        key = (tuple(x.shape), x.dtype, tuple(text_emb.shape), float(transition_mean), float(softness),
               None if uncond_text_emb is None else tuple(uncond_text_emb.shape))
        if self._graph is None or self._gkey != key:
            self._sx = torch.empty_like(x)
            self._ssig = torch.zeros((), dtype=self.dtype, device=x.device)
            self._stext = text_emb.clone()
            self._sunc = None if uncond_text_emb is None else uncond_text_emb.clone()
            self._sx.copy_(x)
            self._ssig.fill_(max(float(t), 1e-3))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                     # warm-up: registers the weight bank, sizes the allocator pool
                for _ in range(3):
                    self.denoise(self._sx, self._ssig, self._stext, transition_mean, softness, self._sunc)
            torch.cuda.current_stream().wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with hgraph.no_gc(), torch.cuda.graph(self._graph):
                self._sout = self.denoise(self._sx, self._ssig, self._stext, transition_mean, softness, self._sunc)
            self._gkey = key
        # every replay input is refreshed: a later sample() with another prompt of the same shape must not see the captured one
        self._sx.copy_(x)
        self._ssig.fill_(float(t))
        self._stext.copy_(text_emb)
        if self._sunc is not None:
            self._sunc.copy_(uncond_text_emb)
        self._graph.replay()
        return self._sout.clone()

    # ---- fused solver stage (reference :90-107 without churn) ----------------------------------------------------------
    def _stage_state(self, x, text_emb, transition_mean, softness, uncond_text_emb):
        """Static buffers of the device-side solver: latents, sigma schedule (float64, as the host computes it), stage index; with
        use_graph also the two captured graphs (a full Heun stage, and the last Euler-only stage).  The SAME stage function runs
        eagerly (use_graph=False) and under capture, so the two trajectories are bit-identical."""
        key = (tuple(x.shape), tuple(text_emb.shape), float(transition_mean), float(softness), self.num_steps, bool(self.use_graph),
               None if uncond_text_emb is None else tuple(uncond_text_emb.shape))
        if self._stage is not None and self._skey == key:
            return self._stage
        dev = x.device
        st = dict(x=torch.empty_like(x), xn=torch.empty_like(x), sig=torch.ones((), dtype=torch.float32, device=dev),
                  t=torch.ones(self.num_steps + 1, dtype=torch.float64, device=dev), idx=torch.zeros(1, dtype=torch.int32, device=dev),
                  text=text_emb.clone(), unc=None if uncond_text_emb is None else uncond_text_emb.clone())
        st["x"].copy_(x)
        n = x.numel()

        def stage(last: bool):
            ops.call("hdmoe_sched_pick", st["sig"], st["t"], st["idx"], 0)
            den = self.denoise(st["x"], st["sig"], st["text"], transition_mean, softness, st["unc"])
            if last:
                ops.call("hdmoe_heun_euler", st["x"], st["x"], den, st["t"], st["idx"], n)
            else:
                ops.call("hdmoe_heun_euler", st["xn"], st["x"], den, st["t"], st["idx"], n)
                ops.call("hdmoe_sched_pick", st["sig"], st["t"], st["idx"], 1)
                den2 = self.denoise(st["xn"], st["sig"], st["text"], transition_mean, softness, st["unc"])
                ops.call("hdmoe_heun_correct", st["x"], st["x"], den, st["xn"], den2, st["t"], st["idx"], n)
            ops.call("hdmoe_idx_advance", st["idx"])

        st["stage"] = stage
        if self.use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                     # warm-up: registers the weight bank, sizes the allocator pool
                for _ in range(2):
                    st["idx"].zero_()
                    stage(False)
            torch.cuda.current_stream().wait_stream(side)
            st["g_heun"], st["g_last"] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            st["idx"].zero_()
            with hgraph.no_gc():
                with torch.cuda.graph(st["g_heun"]):
                    stage(False)
                with torch.cuda.graph(st["g_last"], pool=st["g_heun"].pool()):
                    stage(True)
        self._stage, self._skey = st, key
        return st

    def _eval(self, x, t, text_emb, transition_mean, softness, uncond_text_emb):
        if self.use_graph:
            return self._denoise_graphed(x, t, text_emb, transition_mean, softness, uncond_text_emb)
        sig = torch.tensor(t, dtype=self.dtype, device=x.device)
        return self.denoise(x, sig, text_emb, transition_mean, softness, uncond_text_emb)

    def t_schedule(self, device):
        """Karras rho schedule with the appended 0 (reference :80-87), computed on the host in float64."""
        i = np.arange(self.num_steps, dtype=np.float64)
        t = (self.sigma_max ** (1 / self.rho) + i / (self.num_steps - 1) *
             (self.sigma_min ** (1 / self.rho) - self.sigma_max ** (1 / self.rho))) ** self.rho
        return np.concatenate([t, [0.0]])

    @torch.no_grad()
    def sample(self, noise: torch.Tensor, text_emb: torch.Tensor, transition_mean: float, softness: float,
               uncond_text_emb: torch.Tensor = None) -> torch.Tensor:
        device = noise.device
        if self.use_graph:
            # the captured evaluation holds no weight-prepare launch when the images were current at capture time: refresh them here,
            # eagerly, in case the parameters changed since (new checkpoint, an optimizer step)
            from hdmoe_hip import bank as wbank
            for m_ in {id(self.model): self.model, id(self.gnet): self.gnet}.values():
                if isinstance(m_, nn.Module) and getattr(m_, "_hdmoe_bank", None) is not None:
                    m_._hdmoe_bank.refresh_eval()
        t_steps = self.t_schedule(device)
        x_next = ops.axpby(noise.to(self.dtype), None, float(t_steps[0]), 0.0)
        self.fused_heun = bool(self.s_churn <= 0 and self.dtype == torch.float32 and noise.is_cuda and self.num_steps >= 2)
        if self.fused_heun:
            # no churn: the whole solver runs from a device-side schedule (fused Euler / Heun-correction kernels, no host arithmetic between
            # the evaluations); with use_graph each stage is one hipGraph replay
            st = self._stage_state(x_next, text_emb, transition_mean, softness, uncond_text_emb)
            st["x"].copy_(x_next)
            st["text"].copy_(text_emb)
            if st["unc"] is not None:
                st["unc"].copy_(uncond_text_emb)
            st["t"].copy_(torch.from_numpy(t_steps))
            st["idx"].zero_()
            for i in range(self.num_steps):
                last = i == self.num_steps - 1
                if self.use_graph:
                    (st["g_last"] if last else st["g_heun"]).replay()
                else:
                    st["stage"](last)
            return st["x"].clone()
        for i in range(self.num_steps):
            t_cur, t_next = float(t_steps[i]), float(t_steps[i + 1])
            x_cur = x_next
            gamma = min(self.s_churn / self.num_steps, np.sqrt(2) - 1) if (self.s_churn > 0 and self.s_min <= t_cur <= self.s_max) else 0
            t_hat = t_cur + gamma * t_cur
            x_hat = x_cur
            if gamma > 0:
                x_hat = ops.axpby(x_cur, ops.randn_like(x_cur, 1.0), 1.0, float(np.sqrt(t_hat ** 2 - t_cur ** 2) * self.s_noise))
            denoised = self._eval(x_hat, t_hat, text_emb, transition_mean, softness, uncond_text_emb)
            # d_cur = (x_hat - denoised)/t_hat ; x_next = x_hat + (t_next - t_hat) * d_cur
            h = (t_next - t_hat)
            x_next = ops.axpby(x_hat, denoised, 1.0 + h / t_hat, -h / t_hat)
            if i < self.num_steps - 1:
                den2 = self._eval(x_next, t_next, text_emb, transition_mean, softness, uncond_text_emb)
                # x_next = x_hat + h * (0.5*d_cur + 0.5*d_prime),  d_prime = (x_next - den2)/t_next
                d_cur_term = ops.axpby(x_hat, denoised, 1.0 + 0.5 * h / t_hat, -0.5 * h / t_hat)     # x_hat + 0.5 h d_cur
                d_pr = ops.axpby(x_next, den2, 0.5 * h / t_next, -0.5 * h / t_next)                  # 0.5 h d_prime
                x_next = ops.axpby(d_cur_term, d_pr, 1.0, 1.0)
        return x_next
