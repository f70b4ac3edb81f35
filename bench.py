#!/usr/bin/env python3
"""Benchmark of the VidToMe token-merging hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): denoising steps/sec for a 16-frame 512x512 SD-1.5 chunk, merge ratio 0.5.
A *step* is one pass of the hot path over one chunk: the patched self-attention segment
``norm1 -> compute_merge -> attn1 -> unmerge -> + residual`` (vidtome/patch.py:139-169) at all 16 SD-1.5
transformer-block sites (10 merged: 5x N=4096/C=320/d=40 and 5x N=1024/C=640/d=80; 6 un-merged at C=1280),
batch 2 (CFG [uncond | cond]) x 16 frames, local merge 0.5 + global merge 0.5 in steady state (the
block's anchor tokens were populated by a preceding chunk, as for every chunk but the first of a step).
Synthetic fp16 hidden states (frame-correlated), random-init weights; inputs are resident in HBM before
the timed region.  N > 1: one process per GPU, each rank runs its own chunk (weak scaling); local merging needs
no collective, the global level takes its anchor tokens from an RCCL all-gather of every rank's local merged tokens
per merging block (chunk_parallel.AllGatherExchange); value = chunk-steps per second over all ranks.

The JSON line also carries
  roofline:     the dominant kernel (attention_kernel: flash attention over the merged tokens, fp16 MFMA), its
                algorithmic FLOPs / HIP-event time over the timed region vs the 2.5 PFLOP/s dense fp16 peak;
  matching:     the fused cosine score + row top-1 step (fp32-exact), algorithmic FLOPs / HIP-event time;
  cpu_baseline: the CPU oracle (a port of the reference's algorithm) timed on this host's cores on a
                bounded sample of the same workload and extrapolated to a whole step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails in hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_PEAK_TFLOPS = 157.3          # MI355X fp32 vector / fp32-MFMA peak (MI355X_MICROARCH.md)
FP16_PEAK_TFLOPS = 2500.0         # dense fp16/bf16 MFMA peak (not the 2:1-sparse marketing figure)
BATCH, FRAMES, LATENT = 2, 16, (64, 64)
LOCAL_RATIO, GLOBAL_RATIO = 0.5, 0.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    ap.add_argument("--local-only", action="store_true", help="merge_global=False variant (not the headline)")
    return ap.parse_args()


class KernelTimer:
    """HIP events around the hot kernels' launches, recorded on the launch stream (torch's current stream):
    `attention` = vtm_attention (one kernel), `matching` = vtm_match_filtered / vtm_match (the fused
    score + top-1 step; the filtered variant is filter + refine kernels)."""

    def __init__(self, lib_mod):
        self.lib_mod = lib_mod
        self.orig = {}
        self.records = {"attention": [], "matching": []}
        self.enabled = False

    def _wrap(self, name, kind, flops_of):
        orig = getattr(self.lib_mod, name)
        self.orig[name] = orig

        def timed(*a, **k):
            if not self.enabled:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            self.records[kind].append((flops_of(*a, **k), e0, e1))
            return out
        setattr(self.lib_mod, name, timed)

    def __enter__(self):
        # attention(q, k, vt, heads, M, scale, share): 4 * B * M^2 * C flops (QK^T + PV, all heads)
        self._wrap("attention", "attention", lambda q, k, vt, heads, M, scale, share=1: 4.0 * q.shape[0] * M * M * q.shape[2])
        # attention_kv(q, k, vt, heads, Mq, Mk, scale): 4 * B * Mq * Mk * C executed flops
        self._wrap("attention_kv", "attention", lambda q, k, vt, heads, Mq, Mk, scale: 4.0 * q.shape[0] * Mq * Mk * q.shape[2])
        # match_filtered(x0, x1, a_rows, b_rows, align): 2 * B * Ns * Nd * C algorithmic flops
        self._wrap("match_filtered", "matching",
                   lambda x0, x1, ar, br, align, want_flag=False: 2.0 * x0.shape[0] * ar.shape[1] * br.shape[1] * x0.shape[2])
        self._wrap("match", "matching", lambda a, b, Ns, Nd, align: 2.0 * a.shape[0] * Ns * Nd * a.shape[1] * 8)
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(self.lib_mod, name, fn)

    def summary(self, kind):
        rec = self.records[kind]
        flops = sum(r[0] for r in rec)
        ms = sum(r[1].elapsed_time(r[2]) for r in rec)
        return flops, ms, len(rec)

    def largest(self, kind):
        """(flops, average ms, count) of the launches with the most work -- the top-block launches, whose average
        duration is what the rocprofv3 summary under profiles/ lists for the same kernel instantiation."""
        rec = self.records[kind]
        if not rec:
            return 0.0, 0.0, 0
        top = max(r[0] for r in rec)
        sel = [r for r in rec if r[0] == top]
        return top, sum(r[1].elapsed_time(r[2]) for r in sel) / len(sel), len(sel)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel's largest configuration (top-block attention: 34 816 live queries
    x 52 224 keys),
    from the rocprofv3 PMC passes recorded in profiles/r01_pmc_traffic.json (FETCH_SIZE doubled per the gfx950
    note + WRITE_SIZE); None if the profile file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            d = json.load(f)
        for key in ("attention_kernel<half,40> B=2 h=8 Mq=34816 Mk=52224 (r01_k)",
                    "attention_kernel<half,40> B=2 h=8 Mq=34816 Mk=52224 (r01_f)",
                    "attention_kernel<half,40> B=2 h=8 M=52224 (r01_e)", "attention_kernel<half,40> B=2 h=8 M=52224"):
            if key in d:
                return int(d[key]["hbm_bytes_per_launch"])
        return None
    except Exception:
        return None


def cpu_baseline(target_seconds: float):
    """Time the CPU oracle on a bounded sample of the cfg-2 step and extrapolate to the whole step.
    Sample: for one top site (N=4096, C=320) and one mid site (N=1024, C=640): the three matching levels on a
    slice of src rows, the attention on a slice of query rows, and the projections on a slice of rows; the
    un-merged sites are cheap and measured on one site each.  Everything is scaled by (full rows / sampled
    rows) x (sites of that kind)."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    cores = oracle.num_threads()
    rng = np.random.default_rng(0)
    total = 0.0
    spent = 0.0
    detail = {}
    scale_rows = max(0.25, target_seconds / 20.0)

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def two_point(fn, r, full):
        """t(rows) = fixed + slope * rows measured at r and 2r (the fixed part -- operand transposes, thread
        start-up -- must not be multiplied by the extrapolation factor)."""
        nonlocal spent
        t1, t2 = timed(lambda: fn(r)), timed(lambda: fn(2 * r))
        spent += t1 + t2
        slope = max(t2 - t1, 0.0) / r
        fixed = max(t1 - slope * r, 0.0)
        return fixed + slope * full

    for kind, N, C, heads, nsites in (("top", 4096, 320, 8, 5), ("mid", 1024, 640, 8, 5)):
        L = FRAMES * N
        levels = [(3 * L // 4, L // 4)]                                     # level 1: 12 src / 4 dst frames
        U1 = levels[0][0] - int(levels[0][0] * LOCAL_RATIO)
        levels.append((3 * N, N + U1))                                       # level 2
        Ml = (levels[1][0] - int(levels[1][0] * LOCAL_RATIO)) + levels[1][1]
        levels.append((Ml, Ml))                                              # global (square)
        M = (Ml - int(Ml * GLOBAL_RATIO)) + Ml
        t_kind = 0.0
        for (Ns, Nd) in levels:
            a = rng.standard_normal((BATCH, Ns, C)).astype(np.float32)
            b = rng.standard_normal((BATCH, Nd, C)).astype(np.float32)
            r = int(min(Ns // 2, max(256, 32 * cores * scale_rows)))
            t_kind += two_point(lambda rows: oracle.match(a, b, rows=(0, rows)), r, Ns)
        q = rng.standard_normal((BATCH, M, C)).astype(np.float32)
        r = int(min(M // 2, max(64, 16 * cores * scale_rows)))
        t_kind += two_point(lambda rows: oracle.attention(q, q, q, heads, rows=(0, rows)), r, M)
        w = rng.standard_normal((C, C)).astype(np.float32)
        r = min(M // 2, 4096)
        t_kind += two_point(lambda rows: [q[:, :rows] @ w for _ in range(4)], r, M)
        detail[kind] = round(t_kind, 2)
        total += t_kind * nsites
    # un-merged sites: per-frame attention, N=256 (5 sites) and N=64 (1 site), C=1280
    for N, nsites in ((256, 5), (64, 1)):
        C, heads = 1280, 8
        x = rng.standard_normal((4, N, C)).astype(np.float32)
        w = rng.standard_normal((C, C)).astype(np.float32)
        t = timed(lambda: (oracle.attention(x, x, x, heads), [x.reshape(-1, C) @ w for _ in range(4)]))
        spent += t
        total += t * (BATCH * FRAMES / 4) * nsites
    return {"value": 1.0 / total, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (C/OpenMP fp32) on {cores} host threads: matching on row slices of the 3 levels, "
                      f"attention on query-row slices, projections on row slices of one top and one mid site "
                      f"(+1 un-merged site each), {spent:.1f} s measured, extrapolated to the full 16-site step "
                      f"({total:.0f} s/step)",
            "seconds_per_step_estimate": round(total, 1)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # VIDTOME_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 code path run on a box with fewer GPUs than
    # ranks (the ranks then share devices); the driver's runs use RCCL with one GPU per rank
    backend = os.environ.get("VIDTOME_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)    # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import vidtome_amd
    from vidtome_amd import _lib, sites

    unet = sites.SiteUNet(sites.sd15_sites(), seed=0).to(device=dev, dtype=torch.float16)
    vidtome_amd.apply_patch(unet, local_merge_ratio=LOCAL_RATIO, merge_global=not args.local_only,
                            global_merge_ratio=GLOBAL_RATIO, batch_size=BATCH, target_stride=4, global_rand=0.5)
    unet.set_size(LATENT)
    if world > 1 and not args.local_only:
        # north-star multi-GPU mode: every rank owns one chunk; per merging block the ranks all-gather their
        # local merged tokens over RCCL/xGMI and merge against the previous rank's (chunk_parallel.py)
        from vidtome_amd import chunk_parallel as cp
        cp.enable(unet, cp.AllGatherExchange())
    torch.manual_seed(123)           # the block generators fork this state (default.yaml seed)
    # each rank works on its own chunk of the video: different synthetic frames per rank
    hiddens = [sites.synthetic_hidden(s, BATCH, FRAMES, LATENT, torch.float16, dev, seed=1234 + 97 * rank + i)
               for i, s in enumerate(sites.sd15_sites())]

    def step():
        with torch.no_grad():
            return sites.run_segment_pass(unet, hiddens)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()                            # preceding chunk: populates the anchor tokens (steady state)
    for _ in range(args.warmup):
        step()
    with KernelTimer(_lib) as mt:
        mt.enabled = True
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    aflops, ams, an = mt.summary("attention")
    top_flops, top_ms, top_n = mt.largest("attention")
    mflops, mms, mn = mt.summary("matching")

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * args.steps / dt
        att_tf = aflops / (ams * 1e-3) / 1e12 if ams > 0 else 0.0
        mat_tf = mflops / (mms * 1e-3) / 1e12 if mms > 0 else 0.0
        from vidtome_amd import merge as _merge
        line = {
            "metric": "denoising steps/sec, 16-frame 512x512 SD-1.5 chunk, ratio=0.5",
            "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 tokens / f32-exact matching / f16 MFMA attention",
            "data": "synthetic",
            "config": {"workload": "SD-1.5 16 frames 512x512 (cfg-2): hot-path pass over the 16 transformer-block "
                                   "sites, batch 2 (CFG), local merge 0.5" +
                                   ("" if args.local_only else " + global merge 0.5 (steady state)"),
                       "sites": 16, "merged_sites": 10, "chunk_frames": FRAMES, "batch": BATCH,
                       "matcher": _merge.MATCH_MODE,
                       "parallelism": f"chunk-parallel x{world}" + (", RCCL all-gather of the anchor tokens per merging "
                                                                     "block" if world > 1 else "")},
            # dominant single kernel of the step: the merged-token self-attention (MFMA-bound)
            # `achieved` counts EXECUTED flops (4 B Mq Mk C per launch): with a global level the block only computes the
            # attention rows unmerge() reads, so the reference-algorithmic 4 B M^2 C would overstate the kernel
            "roofline": {"kernel": "attention_kernel<half,d> (flash attention over merged tokens, "
                                   "v_mfma_f32_32x32x16_f16; executed flops)",
                         "bound": "mfma", "achieved": round(att_tf, 1), "peak": FP16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(att_tf / FP16_PEAK_TFLOPS, 4), "traffic": pmc_traffic(),
                         "launches": an, "avg_launch_ms": round(ams / max(an, 1), 4),
                         # the top-block launches alone (compare with attention_kernel<half,40> in profiles/*_kernel_stats.txt)
                         "top_block": {"launches": top_n, "avg_ms": round(top_ms, 4),
                                       "tflops": round(top_flops / (top_ms * 1e-3) / 1e12, 1) if top_ms > 0 else 0.0},
                         "attention_ms_per_step": round(ams / args.steps, 3)},
            # the fused similarity + top-1 step (second largest): algorithmic fp32 FLOPs of the reference's
            # `a @ b.T` + max over HIP-event time; the filtered matcher produces the fp32-exact result with
            # fp16-MFMA filtering, so its algorithmic rate may exceed the fp32 peak it is quoted against
            "matching": {"kernels": "filter_kernel + refine_kernel (vtm_match_filtered)" if _merge.MATCH_MODE != "exact"
                                    else "match_kernel (vtm_match)",
                         "algorithmic_tflops": round(mat_tf, 1), "fp32_peak": FP32_PEAK_TFLOPS,
                         "frac_of_fp32_peak": round(mat_tf / FP32_PEAK_TFLOPS, 3), "calls": mn,
                         "matching_ms_per_step": round(mms / args.steps, 3)},
        }
        if not args.no_cpu_baseline and world == 1:           # reported on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
