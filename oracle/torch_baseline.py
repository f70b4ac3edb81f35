"""Plain-PyTorch CPU restatement of the patched self-attention segment -- the TIMING baseline of bench.py
(`cpu_baseline.kind = "torch"`).  Test infrastructure like the rest of oracle/: only tests/ and bench.py's
cpu_baseline leg import it; the product never does.

What it restates (reference file:line), with the same tensor operations the reference issues on the CPU -- a
materialised score matrix from `bmm`, `max`, `argsort`, `gather`, zero-fill + scatter for the unmerge, SDPA for the
attention -- so that its run time is what the reference's own PyTorch CPU path costs on the same cores:

* partition of the joined chunk into src / dst frames            vidtome/merge.py:41-74
* normalise, split, a @ b^T, row max, argsort, index split        vidtome/merge.py:76-117 (non-aligned branch)
* merge = cat(gather(src, unm), dst)                              vidtome/merge.py:119-133 ("replace" mode)
* unmerge = zeros + three scatters                                vidtome/merge.py:135-155
* global level: src = first src_len tokens, dst = the rest        vidtome/merge.py:343-463
* compute_merge: level loop, coin, anchors update                 vidtome/patch.py:14-91
* block segment: norm1 -> merge -> attn1 -> unmerge -> + residual vidtome/patch.py:139-169; attention arithmetic of
  utils/pnp_utils.py:47-95 through F.scaled_dot_product_attention (what Diffusers' default processor calls)

It is NOT the parity oracle (torch's bmm / norm summation order is unspecified, its argsort unstable): parity is
pinned by oracle.py / vtm_oracle.c against the reference-generated fixtures.  tests/test_oracle_golden.py checks this
file against oracle.py on a small case so that the baseline times the right algorithm.
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


def _match(tokens: torch.Tensor, a_idx: torch.Tensor, b_idx: torch.Tensor, ratio: float):
    """merge.py:84-117 / 389-421: cosine scores of every src against every dst token, top-1, rank by similarity."""
    unit = tokens / tokens.norm(dim=-1, keepdim=True)
    a, b = unit[:, a_idx], unit[:, b_idx]
    scores = torch.bmm(a, b.transpose(1, 2))                       # (B, Ns, Nd) materialised, like the reference
    r = min(a.shape[1], int(a.shape[1] * ratio))
    node_max, node_idx = scores.max(dim=-1)
    order = node_max.argsort(dim=-1, descending=True)
    src_idx, unm_idx = order[:, :r], order[:, r:]
    dst_idx = node_idx.gather(1, src_idx)
    return unm_idx, src_idx, dst_idx


class _Level:
    """One matching level: index sets + the two closures the reference returns."""

    def __init__(self, a_idx, b_idx, unm_idx, src_idx, dst_idx, n_in: int, out_slice: Optional[Tuple[int, int]] = None):
        self.a_idx, self.b_idx, self.unm_idx, self.src_idx, self.dst_idx = a_idx, b_idx, unm_idx, src_idx, dst_idx
        self.n_in, self.out_slice = n_in, out_slice

    def merge(self, x: torch.Tensor) -> torch.Tensor:                # merge.py:119-133
        C = x.shape[-1]
        src, dst = x[:, self.a_idx], x[:, self.b_idx]
        unm = src.gather(1, self.unm_idx.unsqueeze(-1).expand(-1, -1, C))
        return torch.cat([unm, dst], dim=1)

    def unmerge(self, x: torch.Tensor) -> torch.Tensor:              # merge.py:135-155, 439-460
        B, _, C = x.shape
        U = self.unm_idx.shape[1]
        unm, dst = x[:, :U], x[:, U:]
        src = dst.gather(1, self.dst_idx.unsqueeze(-1).expand(-1, -1, C))
        out = torch.zeros(B, self.n_in, C, dtype=x.dtype)
        out[:, self.b_idx] = dst
        a_pos = self.a_idx.unsqueeze(0).expand(B, -1)
        out.scatter_(1, a_pos.gather(1, self.unm_idx).unsqueeze(-1).expand(-1, -1, C), unm)
        out.scatter_(1, a_pos.gather(1, self.src_idx).unsqueeze(-1).expand(-1, -1, C), src)
        if self.out_slice is not None:
            out = out[:, self.out_slice[0]:self.out_slice[1]]
        return out

    @property
    def unm_num(self) -> int:
        return self.unm_idx.shape[1]


def local_level(tokens: torch.Tensor, F_: int, ratio: float, unm_pre: int, randf: int, target_stride: int) -> _Level:
    """merge.py:41-74: frames with (frame % stride == randf) are dst; earlier levels' unmerged tokens ride along as dst."""
    N = tokens.shape[1]
    tnum = (N - unm_pre) // F_
    ts = min(target_stride, F_)
    pos = torch.arange(N - unm_pre)
    is_dst = (pos // tnum) % ts == randf
    a_idx = pos[~is_dst] + unm_pre
    b_idx = torch.cat([pos[is_dst] + unm_pre, torch.arange(unm_pre)])
    unm, src, dst = _match(tokens, a_idx, b_idx, ratio)
    return _Level(a_idx, b_idx, unm, src, dst, N)


def global_level(tokens: torch.Tensor, src_len: int, ratio: float, unmerge_chunk: int) -> _Level:
    """merge.py:343-463: src = the first src_len tokens, dst = the rest; unmerge returns one of the two parts."""
    N = tokens.shape[1]
    a_idx, b_idx = torch.arange(src_len), torch.arange(src_len, N)
    unm, src, dst = _match(tokens, a_idx, b_idx, ratio)
    return _Level(a_idx, b_idx, unm, src, dst, N, (0, src_len) if unmerge_chunk == 0 else (src_len, N))


def compute_merge(x: torch.Tensor, batch_size: int, args: Dict, state: Dict, gen: torch.Generator):
    """patch.py:14-91 for a block that merges.  x: (B*F, N, C) -> (levels, merged tokens (B, M, C))."""
    fsize, tsize = x.shape[0] // batch_size, x.shape[1]
    tokens = x.reshape(batch_size, fsize * tsize, x.shape[2])
    levels: List[_Level] = []
    unm, curF = 0, fsize
    while curF > 1:
        randf = int(torch.randint(0, min(args["target_stride"], curF), (1,), generator=gen))
        lv = local_level(tokens, curF, args["local_merge_ratio"], unm, randf, args["target_stride"])
        unm += lv.unm_num
        tokens = lv.merge(tokens)
        levels.append(lv)
        curF = (tokens.shape[1] - unm) // tsize
    if args["merge_global"]:
        anchors = state.get("global_tokens")
        if anchors is None:
            state["global_tokens"] = tokens.clone()
        else:
            if float(torch.rand(1, generator=gen)) > args["global_rand"]:
                both, src_len, part = torch.cat([tokens, anchors], dim=1), tokens.shape[1], 0
            else:
                both, src_len, part = torch.cat([anchors, tokens], dim=1), anchors.shape[1], 1
            lv = global_level(both, src_len, args["global_merge_ratio"], part)
            tokens = lv.merge(both)
            levels.append(lv)
            state["global_tokens"] = lv.unmerge(tokens).clone()
    return levels, tokens


def segment(hidden: torch.Tensor, batch_size: int, args: Dict, state: Dict, gen: torch.Generator, w: Dict,
            heads: int, merges: bool = True) -> torch.Tensor:
    """patch.py:139-169: norm1 -> compute_merge -> attn1(merged) -> unmerge -> + residual, fp32 on the CPU."""
    C = hidden.shape[-1]
    nh = F.layer_norm(hidden, (C,), w["ln_w"], w["ln_b"])
    if merges:
        levels, merged = compute_merge(nh, batch_size, args, state, gen)
    else:
        levels, merged = [], nh
    B, M, _ = merged.shape
    d = C // heads
    q = F.linear(merged, w["wq"]).view(B, M, heads, d).transpose(1, 2)
    k = F.linear(merged, w["wk"]).view(B, M, heads, d).transpose(1, 2)
    v = F.linear(merged, w["wv"]).view(B, M, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, M, C)
    o = F.linear(o, w["wo"], w["bo"])
    for lv in reversed(levels):
        o = lv.unmerge(o)
    return o.reshape(hidden.shape) + hidden


def random_weights(C: int, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w = {n: torch.randn(C, C, generator=g) * C ** -0.5 for n in ("wq", "wk", "wv", "wo")}
    w.update(ln_w=torch.ones(C), ln_b=torch.zeros(C), bo=torch.zeros(C))
    return w


def time_step(batch: int, frames: int, latent_hw: Tuple[int, int], site_list, args: Dict, budget_s: float,
              seed: int = 0) -> Dict:
    """Time the segment on ONE site of every kind (same shape => same cost) in steady state (anchors populated by a
    preceding chunk, as in bench.py) and add the kinds up to a whole step.  A merged site whose full batch would
    blow the time budget is timed on ONE batch sample (the reference loops nothing over the batch, but every operation
    of the path is independent per sample) and doubled; `sampled` says which.  Every kind gets one untimed warm-up pass
    and the median of two timed ones (a single cold pass spread 108-123 s across boxes)."""
    torch.set_grad_enabled(False)
    kinds: Dict[Tuple[int, int, int], int] = {}
    for s in site_list:
        kinds[(s.downsample, s.channels, s.heads)] = kinds.get((s.downsample, s.channels, s.heads), 0) + 1
    total, detail, sampled, spent = 0.0, {}, [], 0.0
    # cheapest kinds first: their cost calibrates the estimate for the big one
    order = sorted(kinds, key=lambda kd: -kd[0])
    per_flop = None
    for (ds, C, heads) in order:
        n_sites = kinds[(ds, C, heads)]
        N = (latent_hw[0] // ds) * (latent_hw[1] // ds)
        merges = ds <= args["max_downsample"]
        b_run = batch
        if merges:
            L = frames * N
            flops = batch * (2.0 * 1.2 * L * L * C * 0.6 + 4.0 * (0.8 * L) ** 2 * C)        # rough: matching + attention
            # (three passes per kind: warm-up + two timed)
            if per_flop is not None and 3.0 * per_flop * flops > 0.6 * max(budget_s - spent, 1.0) and batch > 1:
                b_run = 1
                sampled.append(f"ds{ds}")
        else:
            flops = batch * frames * 4.0 * N * N * C
        g = torch.Generator().manual_seed(seed + ds)
        base = torch.randn(b_run, 1, N, C, generator=g)
        make = lambda: (base + 0.5 * torch.randn(b_run, frames, N, C, generator=g)).reshape(b_run * frames, N, C)
        w = random_weights(C, seed + C)
        gen = torch.Generator().manual_seed(123)
        state: Dict = {}
        if merges:
            compute_merge(F.layer_norm(make(), (C,)), b_run, args, state, gen)               # preceding chunk: anchors
        # one untimed warm-up pass (first-touch page faults, oneDNN / OpenMP start-up, the allocator's high-water mark), then
        # `reps` timed passes over fresh chunks of the same clip; the site's time is their median (reps = 2: the mean of the
        # two, both listed).  A site kind whose warm-up alone shows that three passes would blow the budget keeps ONE timed
        # pass (`reps` says so).
        t0 = time.perf_counter()
        segment(make(), b_run, args, state, gen, w, heads, merges)
        warm = time.perf_counter() - t0
        spent += warm
        reps = 2 if spent + 2.0 * warm <= budget_s else 1
        times = []
        for _ in range(reps):
            x = make()
            t0 = time.perf_counter()
            segment(x, b_run, args, state, gen, w, heads, merges)
            times.append(time.perf_counter() - t0)
        spent += sum(times)
        dt = sorted(times)[len(times) // 2] if len(times) % 2 else sum(sorted(times)[len(times) // 2 - 1:len(times) // 2 + 1]) / 2.0
        dt_full = dt * (batch / b_run)
        if merges and b_run == batch:
            per_flop = dt / flops if per_flop is None else min(per_flop, dt / flops)
        detail[f"ds{ds}_C{C}"] = {"seconds_per_site": round(dt_full, 3), "sites": n_sites, "timed_batch": b_run,
                                  "warmup_s": round(warm, 3), "timed_s": [round(t, 3) for t in times]}
        total += dt_full * n_sites
    return {"seconds_per_step": total, "detail": detail, "sampled": sampled, "spent": spent}
