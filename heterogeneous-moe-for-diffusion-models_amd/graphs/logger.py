"""Training logger without per-step host syncs.

Drop-in for the reference's ``graphs/logger.py`` ``Logger`` (same constructor, method signatures, file names and JSONL
record keys, so ``graphs/plotter.py`` reads the files unchanged).  The reference pulls ~25 scalars per step to the host with
``.item()`` (logger.py:96-121, 195-222) -- each one a device synchronisation in the middle of the training loop.  Here every
tensor-valued metric stays a 0-dim device tensor in the accumulators and is fetched with ONE copy when a record is written
(every ``log_interval`` steps); router / gradient / weight statistics are reduced on the device and fetched with one copy
per record.  Averaging, rounding and record layout follow the reference line by line.
"""
from __future__ import annotations

import json
import math
from collections import defaultdict
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch


def _scalar(v):
    """0-dim float32 tensor on v's device (no sync) or a python float."""
    if isinstance(v, torch.Tensor):
        return v.detach().to(torch.float32).reshape(-1)[0] if v.numel() == 1 else v.detach().to(torch.float32).mean()
    return float(v)


def _sigma_percentile_mean(sigma: torch.Tensor, p_mean: float, p_std: float) -> torch.Tensor:
    log_sigma = torch.log(sigma.detach().to(torch.float32))
    return (0.5 * (1 + torch.erf((log_sigma - p_mean) / (p_std * math.sqrt(2))))).mean()


def _fetch(values: List[torch.Tensor]) -> List[float]:
    """One device->host copy for a list of 0-dim / 1-dim tensors (concatenated)."""
    if not values:
        return []
    flat = torch.cat([v.reshape(-1).to(torch.float64) for v in values])
    return flat.cpu().tolist()


class Logger:
    def __init__(self, log_dir: str = "./training_logs", run_name: str = "experiment", log_interval: int = 10):
        self.log_dir = Path(log_dir)
        self.log_dir.mkdir(parents=True, exist_ok=True)
        self.run_name = run_name
        self.log_interval = log_interval
        self.main_log_file = self.log_dir / f"{run_name}_training.jsonl"
        self.router_log_file = self.log_dir / f"{run_name}_router_stats.jsonl"
        self.gradient_log_file = self.log_dir / f"{run_name}_gradients.jsonl"
        self.weight_log_file = self.log_dir / f"{run_name}_weights.jsonl"
        self.accumulators = defaultdict(list)                # key -> python floats and/or 0-dim device tensors
        print(f"Initialized Logger: {run_name}")
        print(f"  Main log: {self.main_log_file}")
        print(f"  Router stats: {self.router_log_file}")
        print(f"  Gradients: {self.gradient_log_file}")
        print(f"  Weights: {self.weight_log_file}")

    # ------------------------------------------------------------------ reference logger.py:70-121
    def log_training_step(self, step: int, loss_dict: Dict[str, torch.Tensor], zeta: float, log_var, lr: float, p_mean: float,
                          p_std: float, sigma: Optional[torch.Tensor] = None):
        self.accumulators["step"].append(step)
        for key, value in loss_dict.items():
            self.accumulators[key].append(_scalar(value))
        self.accumulators["zeta"].append(float(zeta))
        self.accumulators["log_var"].append(_scalar(log_var))          # the caller may pass the tensor itself (no .item())
        self.accumulators["lr"].append(float(lr))
        if sigma is not None:
            self.accumulators["avg_sigma_percentile"].append(_sigma_percentile_mean(sigma, p_mean, p_std))
        if step % self.log_interval == 0 and len(self.accumulators["step"]) > 0:
            self._flush_training_log()

    # ------------------------------------------------------------------ reference logger.py:123-192
    def log_router_statistics(self, step: int, unet_probs: torch.Tensor, vit_probs: torch.Tensor, p_mean: float, p_std: float,
                              sigma: torch.Tensor):
        if step % self.log_interval != 0:
            return
        with torch.no_grad():
            def stats(probs):
                probs = probs.detach().to(torch.float32)
                usage = probs.mean(dim=0)
                avg = usage / (usage.sum() + 1e-10)
                entropy = -torch.sum(avg * torch.log(avg + 1e-10))
                srt, _ = torch.sort(usage)
                n = srt.numel()
                idx = torch.arange(1, n + 1, device=srt.device, dtype=srt.dtype)
                gini = (2 * torch.sum(idx * srt)) / (n * torch.cumsum(srt, 0)[-1]) - (n + 1) / n
                return [entropy, gini, usage.max(), usage.min(), (usage < 0.01).sum().to(torch.float32), usage.std(), usage]
            su, sv = stats(unet_probs), stats(vit_probs)
            vals = _fetch([_sigma_percentile_mean(sigma, p_mean, p_std)] + su + sv)
        E = unet_probs.shape[1]
        u, v = vals[1:7 + E], vals[7 + E:]
        record = {"step": step, "avg_sigma_percentile": vals[0]}
        for tag, s in (("unet", u), ("vit", v)):
            record[f"{tag}_entropy"] = s[0]
            record[f"{tag}_gini"] = s[1]
            record[f"{tag}_max_usage"] = s[2]
            record[f"{tag}_min_usage"] = s[3]
            record[f"{tag}_dead_experts"] = int(s[4])
            record[f"{tag}_usage_std"] = s[5]
        record["unet_expert_usage"] = u[6:]
        record["vit_expert_usage"] = v[6:]
        self._write_jsonl(self.router_log_file, record)

    # ------------------------------------------------------------------ reference logger.py:194-235
    def log_scaling_gating(self, scaling_factors: torch.Tensor, gate_weights: torch.Tensor, sigma: torch.Tensor):
        sf = scaling_factors.detach().to(torch.float32)
        gw = gate_weights.detach().to(torch.float32)
        sg = sigma.detach().to(torch.float32)
        a = self.accumulators
        a["scaling_vit_mean"].append(sf[:, 0].mean())
        a["scaling_unet_mean"].append(sf[:, 1].mean())
        a["scaling_vit_max"].append(sf[:, 0].max())
        a["scaling_vit_min"].append(sf[:, 0].min())
        a["scaling_unet_min"].append(sf[:, 1].min())
        a["scaling_unet_max"].append(sf[:, 1].max())
        a["gate_wx"].append(gw[:, 0].mean())
        a["gate_wa"].append(gw[:, 1].mean())
        a["noise_level_min"].append(sg.min())
        a["noise_level_max"].append(sg.max())
        a["noise_level_std"].append(sg.std())
        a["noise_level"].append(sg.mean())

    # ------------------------------------------------------------------ reference logger.py:237-266
    def log_gradients(self, step: int, model, component_names: Optional[List[str]] = None):
        if step % self.log_interval != 0:
            return
        if component_names is None:
            component_names = ["Unet_experts", "VIT_experts", "Unet_router", "vit_router", "scaling_net", "cross_attn"]
        names, norms = [], []
        with torch.no_grad():
            for name in component_names:
                if hasattr(model, name):
                    norms.append(self._grad_norm_tensor(getattr(model, name).parameters()))
                    names.append(name)
            vals = _fetch(norms)
        record = {"step": step}
        for name, v in zip(names, vals):
            record[f"{name}_grad_norm"] = v
        self._write_jsonl(self.gradient_log_file, record)

    # ------------------------------------------------------------------ reference logger.py:268-326
    def log_weight_statistics(self, step: int, model: torch.nn.Module):
        if step % (self.log_interval * 50) != 0:
            return
        record = {"step": step}
        with torch.no_grad():
            for name in ["Unet_experts", "VIT_experts"]:
                if not hasattr(model, name):
                    continue
                ws = [p.detach() for p in getattr(model, name).parameters() if p.requires_grad and p.ndim > 1]
                if not ws:
                    record[f"{name}_weight_mean"] = None
                    continue
                per = torch.stack([torch.stack([w.sum().to(torch.float32), w.pow(2).sum().to(torch.float32),
                                                w.min().to(torch.float32), w.max().to(torch.float32)]) for w in ws])
                host = per.to(torch.float64).cpu().numpy()            # one copy per component
                count = sum(w.numel() for w in ws)
                mean = float(host[:, 0].sum()) / count
                var = float(host[:, 1].sum()) / count - mean ** 2
                record[f"{name}_weight_mean"] = round(mean, 6)
                record[f"{name}_weight_std"] = round(float(np.sqrt(max(0, var))), 6)
                record[f"{name}_weight_max"] = round(float(host[:, 3].max()), 6)
                record[f"{name}_weight_min"] = round(float(host[:, 2].min()), 6)
        self._write_jsonl(self.weight_log_file, record)

    # ------------------------------------------------------------------ reference logger.py:328-345
    def _flush_training_log(self):
        if len(self.accumulators["step"]) == 0:
            return
        record = {"step": int(self.accumulators["step"][-1])}
        dev_vals, where = [], []
        for key, values in self.accumulators.items():
            for i, v in enumerate(values):
                if isinstance(v, torch.Tensor):
                    dev_vals.append(v)
                    where.append((key, i))
        fetched = _fetch(dev_vals)                                    # the interval's only synchronisation
        for (key, i), f in zip(where, fetched):
            self.accumulators[key][i] = f
        for key, values in self.accumulators.items():
            if key == "step":
                continue
            if len(values) > 0:
                record[key] = round(float(np.mean(values)), 6)
        self._write_jsonl(self.main_log_file, record)
        self.accumulators.clear()

    @staticmethod
    def _write_jsonl(filepath: Path, record: Dict[str, Any]):
        with open(filepath, "a") as f:
            f.write(json.dumps(record) + "\n")

    @staticmethod
    def _grad_norm_tensor(parameters) -> torch.Tensor:
        grads = [p.grad.detach() for p in parameters if p.grad is not None]
        if not grads:
            return torch.zeros(())
        per = torch.stack([g.to(torch.float32).norm(2) for g in grads])
        return per.to(torch.float64).pow(2).sum().sqrt()

    @staticmethod
    def _compute_grad_norm(parameters) -> float:
        return float(Logger._grad_norm_tensor(parameters))
