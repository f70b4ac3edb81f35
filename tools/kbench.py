#!/usr/bin/env python3
"""Kernel micro-benchmarks on the cfg-2 shapes (used while optimising; run under rocprofv3 for counters).

    python tools/kbench.py match  [--shape top_l1|top_l2|top_g|mid_l1] [--iters 5] [--data n01|corr01|corr002|corr05|smooth|flat25|dup|zero|all]
                                  (data regimes: vidtome_amd/sites.DATA_REGIMES; `zero` = corr05 with ONE zero token among the
                                   dst rows, which sends the whole call to the exact escape; `all` prints the table of regimes)
    python tools/kbench.py attn   [--M 52224 --d 40 --heads 8 --B 2] [--iters 3]
    python tools/kbench.py sort   [--n 49152]
    python tools/kbench.py sites  --shape cfg2|cfg3|cfg4|cfg5            (a hot-path pass at the other BASELINE configurations)
    python tools/kbench.py gather|unmerge|layernorm [--B 4 --n 147456 --C 320]   (HBM-bound kernels beyond the MALL)
    python tools/kbench.py ff     [--n 131072 --C 320]                       (panel GEMMs of the feed-forward vs library GEMMs)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vidtome_amd import _lib  # noqa: E402

SHAPES = {"top_l1": (2, 49152, 16384, 320), "top_l2": (2, 12288, 28672, 320), "top_g": (2, 34816, 34816, 320),
          "mid_l1": (2, 12288, 4096, 640), "mid_g": (2, 8704, 8704, 640), "sd21_l1": (2, 110592, 36864, 320)}


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("--shape", default="top_l1")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--M", type=int, default=52224)
    ap.add_argument("--Mq", type=int, default=0, help="attn: number of query rows (0 = M, square self-attention)")
    ap.add_argument("--d", type=int, default=40)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--n", type=int, default=49152)
    ap.add_argument("--align", action="store_true")
    ap.add_argument("--no-seed", action="store_true", help="match: the filtered matcher WITHOUT its same-position seeds (default: "
                    "seeded like the product path -- the rows of a kbench call are frame-ordered: dst index = position in the "
                    "first dst frame)")
    ap.add_argument("--plan", default="one", choices=["one", "range"], help="match: launch plan of the filtered matcher "
                    "(one launch / scout + range; same bits)")
    ap.add_argument("--shuffle", action="store_true", help="match: src rows and the dst rows behind the first dst frame in "
                    "random order (what levels 2 / global see: similarity-rank order) instead of (frame, position) order")
    ap.add_argument("--scout-steps", type=int, default=0, help="match --plan range: the scout's test depth in 64-channel steps "
                    "(0 = the filter's own)")
    ap.add_argument("--ordered", action="store_true", help="match: sort both row lists by token position first "
                    "(vtm_position_order + vtm_match_filtered_ordered, what levels 2 / global do); the sort is inside the timing")
    ap.add_argument("--C", type=int, default=320)
    ap.add_argument("--share", type=int, default=1, help="attn: share_groups (B must be a multiple)")
    ap.add_argument("--bounded", type=float, default=0.0, help="attn: device-side query count = this fraction of Mq")
    ap.add_argument("--check", action="store_true", help="attn: compare a few rows with an fp32 torch reference")
    ap.add_argument("--power", type=float, default=0.0, help="attn: seconds of back-to-back launches under the sysfs power sampler "
                    "(prints W, MHz and joules per launch)")
    ap.add_argument("--data", default="random", help="attn: random | zeros | const (operand values); match: n01 | corr01 | "
                    "corr05 | flat25 | dup | zero | all (token regime; random = n01 in fp16 straight from the device generator)")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    if a.what == "match":
        from vidtome_amd import sites
        B, Ns, Nd, C = SHAPES[a.shape]
        ra = torch.arange(Ns, dtype=torch.int32, device=dev).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=dev).expand(B, Nd).contiguous()

        N = 4096
        while Ns % N or Nd % N:
            N //= 2
        if a.shuffle:
            gp = torch.Generator().manual_seed(5)
            ra = ra[:, torch.randperm(Ns, generator=gp).to(dev)].contiguous()
            rb = torch.cat([rb[:, :N], rb[:, N:][:, torch.randperm(Nd - N, generator=gp).to(dev)]], 1).contiguous()
        seed = None if a.no_seed else (N, Ns + Nd, None, None)

        def tokens(regime):
            if regime == "random":
                return torch.randn(B, Ns + Nd, C, generator=g, device=dev, dtype=torch.float16)
            gc = torch.Generator().manual_seed(0)
            x = sites.regime_tokens("corr05" if regime == "zero" else regime, B, (Ns + Nd) // N, N, C, gc)
            x = torch.nn.functional.layer_norm(x, (C,)).reshape(B, Ns + Nd, C).half()   # the matcher sees norm1's output
            if regime == "zero":
                x[0, Ns + 5] = 0
            return x.to(dev)

        regimes = ["n01", "corr01", "corr002", "corr05", "smooth", "flat25", "dup", "zero"] if a.data == "all" else [a.data]
        fl = 2.0 * B * Ns * Nd * C
        print(f"match {a.shape} B={B} Ns={Ns} Nd={Nd} C={C} align={a.align} seeded={bool(seed)} plan={a.plan} shuffled={a.shuffle} ordered={a.ordered}")
        print(f"{'data':8s} {'filtered ms':>12s} {'nom TFLOP/s':>12s} {'exact ms':>9s} {'pairs/row':>10s} {'escape rows':>12s} {'whole-call':>10s} {'blocks alive':>12s} equal")
        for regime in regimes:
            x = tokens(regime)
            aop, _ = _lib.normalize_gather(x, None, ra)
            bop, _ = _lib.normalize_gather(x, None, rb)
            med, best = timeit(lambda: _lib.match(aop, bop, Ns, Nd, a.align), a.iters)
            mode = _lib.MATCH_SCOUT_RANGE if a.plan == "range" else _lib.MATCH_ONE_LAUNCH
            def run(want_flag=False):
                if not a.ordered:
                    return _lib.match_filtered(x, None, ra, rb, a.align, want_flag=want_flag, seed=seed, mode=mode,
                                               scout_steps=a.scout_steps)
                a_s, a_o, b_s, b_o, tb = _lib.position_order(ra, rb, Ns + Nd, N, None, Ns + Nd)
                return _lib.match_filtered(x, None, a_s, b_s, a.align, want_flag=want_flag, seed=(N, Ns + Nd, None, tb), mode=mode,
                                           order=(a_o, b_o), scout_steps=a.scout_steps)
            medf, bestf = timeit(run, a.iters)
            if a.ordered:
                meds, _ = timeit(lambda: _lib.position_order(ra, rb, Ns + Nd, N, None, Ns + Nd), a.iters)
                print(f"  (vtm_position_order alone: {meds * 1e3:.1f} us)")
            out, fl_ = run(True)
            same = bool(torch.equal(out, _lib.match(aop, bop, Ns, Nd, a.align)))
            f = fl_.tolist()                # [whole-call exact, non-finite, escape rows, refined pairs, blocks tested, alive, 0, 0]
            rows = Ns if a.align else B * Ns
            print(f"{regime:8s} {medf:12.3f} {fl / medf / 1e9:12.1f} {med:9.3f} {f[3] / rows:10.2f} {f[2]:12d} {f[0]:10d} {(f[5] / f[4] if f[4] else 1.0):12.3f} {same}"
                  + (f"  spans {f[7] / f[4]:.3f}" if f[4] and f[7] else ""))
            del aop, bop
    elif a.what == "attn":
        B, M, h, d = a.B, a.M, a.heads, a.d
        C = h * d
        Mp = (M + 7) // 8 * 8
        qk = torch.randn(B, Mp, 2 * C, generator=g, device=dev, dtype=torch.float16)
        vt = torch.randn(B, C, Mp, generator=g, device=dev, dtype=torch.float16)
        if a.data != "random":       # how much of the time is the operands' switching activity (power-limited clock)?
            qk = torch.zeros_like(qk) if a.data == "zeros" else torch.full_like(qk, 0.25)
            vt = torch.zeros_like(vt) if a.data == "zeros" else torch.full_like(vt, 0.25)
        Mq = a.Mq or M
        q = torch.randn(B, (Mq + 7) // 8 * 8, C, generator=g, device=dev, dtype=torch.float16) if a.Mq else qk[:, :, :C]
        if a.Mq and a.data != "random":
            q = torch.zeros_like(q) if a.data == "zeros" else torch.full_like(q, 0.25)
        kk = qk[:, :, C:]
        G = a.share                      # shared probabilities (pnp_utils.py:57-67): q / k of the first B / G samples
        qc = None
        if a.bounded:                    # device-side live-query counts (vtm_attention_kv_bounded)
            qc = torch.full((B,), int(Mq * a.bounded), dtype=torch.int32, device=dev)
        if a.Mq or G > 1 or qc is not None:
            run = lambda: _lib.attention_kv(q, kk, vt, h, Mq, M, d ** -0.5, share_groups=G, q_count=qc)
        else:
            run = lambda: _lib.attention(q, kk, vt, h, M, d ** -0.5, 1)
        med, best = timeit(run, a.iters)
        live = Mq * (a.bounded or 1.0)
        fl = 4.0 * B * live * M * C
        print(f"attention B={B} Mq={Mq} Mk={M} h={h} d={d} share={G} bounded={a.bounded}: median {med:.3f} ms ({fl / med / 1e9:.1f} "
              f"TFLOP/s), best {best:.3f} ms ({fl / best / 1e9:.1f} TFLOP/s)   [VTM_ATT16={os.environ.get('VTM_ATT16', '')} "
              f"SKEW={os.environ.get('VTM_ATT16_SKEW', '')} DEVPLAN={os.environ.get('VTM_ATT_DEVPLAN', '')}]")
        if a.power:                      # ms AND joules: back-to-back launches for ~a.power seconds under the sysfs sampler of bench.py
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import BoxSampler
            n = max(10, int(a.power * 1e3 / med))
            run()
            torch.cuda.synchronize()
            smp = BoxSampler(0, 0.02)
            smp.start()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            box = smp.stop()
            ms = e0.elapsed_time(e1) / n
            if box and box.get("power_w"):
                w, f = box["power_w"]["mean"], box["sclk_mhz"]["mean"]
                print(f"   power: {n} back-to-back launches, {ms:.3f} ms each at {w:.0f} W package power, {f:.0f} MHz mean -> "
                      f"{w * ms * 1e-3:.3f} J per launch ({w * ms * 1e-3 / (fl * 1e-12):.3f} J per TFLOP)")
            else:
                print(f"   power: {n} back-to-back launches, {ms:.3f} ms each (no sysfs power sample on this box)")
        if a.check:                      # a few rows of every sample against fp32 torch (sa_forward's arithmetic)
            out = run()
            nrow = int(live)
            rows = torch.cat([torch.arange(0, min(96, nrow)), torch.arange(max(0, nrow - 96), nrow)]).to(dev)
            worst = 0.0
            for b in range(B):
                bq = b % (B // G)
                for hh in range(h):
                    sl = slice(hh * d, (hh + 1) * d)
                    sc = (q[bq, rows][:, sl].float() @ kk[bq, :M, sl].float().T) * d ** -0.5
                    ref = torch.softmax(sc, -1) @ vt[b, sl, :M].float().T
                    worst = max(worst, float((out[b, rows][:, sl].float() - ref).abs().max()))
            print(f"   check: max |out - fp32 reference| over {len(rows)} rows x {B} samples x {h} heads = {worst:.2e}")
    elif a.what == "sort":
        keys = torch.randint(0, 2 ** 62, (a.B, a.n), generator=g, device=dev, dtype=torch.int64)
        med, best = timeit(lambda: _lib.sort_desc(keys), a.iters)
        print(f"sort rows={a.B} n={a.n}: median {med * 1e3:.1f} us, best {best * 1e3:.1f} us")
    elif a.what in ("gather", "unmerge", "layernorm"):
        # the HBM-bound kernels at a working set BEYOND the 256 MB Infinity Cache (run under rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE for counter-derived GB/s): cfg-5 top block (SD-2.1-768: L = 16 x 9216 tokens, C = 320)
        # with --B 4, i.e. 377 MB per (B, L, C) fp16 tensor
        B, L, C = a.B, a.n, a.C
        M = int(L * 0.53) // 8 * 8
        x = torch.randn(B, L, C, generator=g, device=dev, dtype=torch.float16)
        if a.what == "layernorm":
            w = torch.ones(C, device=dev, dtype=torch.float16)
            bb = torch.zeros(C, device=dev, dtype=torch.float16)
            med, best = timeit(lambda: _lib.layernorm(x, w, bb, 1e-5), a.iters)
            nbytes = 2.0 * B * L * C * 2
        elif a.what == "gather":
            idx = torch.stack([torch.randperm(L, generator=g, device=dev)[:M] for _ in range(B)]).to(torch.int32)
            idx = idx.sort(dim=1).values.contiguous()       # merged rows are mostly ascending runs of pool rows
            med, best = timeit(lambda: _lib.gather_rows(x, None, idx), a.iters)
            nbytes = 2.0 * B * M * C * 2
        else:
            y = torch.randn(B, M, C, generator=g, device=dev, dtype=torch.float16)
            inv = torch.randint(0, M, (B, L), generator=g, device=dev, dtype=torch.int32)
            med, best = timeit(lambda: _lib.unmerge_add(y, inv, x), a.iters)
            nbytes = (B * M + 2.0 * B * L) * C * 2
        print(f"{a.what} B={B} L={L} M={M} C={C}: {nbytes / 1e6:.1f} MB algorithmic, median {med * 1e3:.1f} us "
              f"({nbytes / med / 1e6:.0f} GB/s), best {best * 1e3:.1f} us ({nbytes / best / 1e6:.0f} GB/s)")
    elif a.what == "linear":
        # the block's projection GEMMs at a merged site: K/V^T over all M rows through the merge map, Q over the live
        # rows, out over the attention output -- vtm_linear_rows vs gather + library GEMM (torch -> hipBLASLt)
        import torch.nn.functional as F
        B, L, C, M, Mq = a.B, a.n, a.C, a.M, (a.Mq or a.M)
        x = torch.randn(B, L, C, generator=g, device=dev, dtype=torch.float16)
        idx = torch.stack([torch.randperm(L, generator=g, device=dev)[:M] for _ in range(B)]).to(torch.int32)
        idx = idx.sort(dim=1).values.contiguous()
        q_rows = torch.stack([torch.randperm(M, generator=g, device=dev)[:Mq] for _ in range(B)]).to(torch.int32).contiguous()
        w = (torch.randn(C, C, generator=g, device=dev) * C ** -0.5).half()
        for name, fn, fl in (
                ("k   rows", lambda: _lib.linear_rows(x, None, idx, None, M, w, None), 2.0 * B * M * C * C),
                ("v^T rows", lambda: _lib.linear_rows(x, None, idx, None, M, w, None, transposed=True), 2.0 * B * M * C * C),
                ("q   rows", lambda: _lib.linear_rows(x, None, idx, q_rows, Mq, w, None), 2.0 * B * Mq * C * C),
                ("k   blas", lambda: F.linear(_lib.gather_rows(x, None, idx), w), 2.0 * B * M * C * C),
                ("v^T blas", lambda: torch.matmul(w, _lib.gather_rows(x, None, idx).transpose(1, 2)), 2.0 * B * M * C * C)):
            REP = 8                                   # several launches per timed interval: these kernels are short
            med, best = timeit(lambda: [fn() for _ in range(REP)], a.iters)
            med, best = med / REP, best / REP
            print(f"linear {name} B={B} M={M} Mq={Mq} C={C}: median {med * 1e3:.1f} us ({fl / med / 1e9:.1f} TFLOP/s), best {best * 1e3:.1f} us")
    elif a.what == "ff":
        # the feed-forward of one site: LayerNorm -> GEGLU projection -> gated activation -> output Linear + residual,
        # as panel GEMMs (csrc/ff.hip) and as round 2's library GEMMs around vtm_geglu
        import torch.nn.functional as F
        from vidtome_amd import patch as vpatch
        from vidtome_amd import sites
        n, C = a.n, a.C
        ff = sites.FeedForward(C).to(device=dev, dtype=torch.float16).eval()
        norm = torch.nn.LayerNorm(C).to(device=dev, dtype=torch.float16)
        h = torch.randn(n // 4096 if n >= 4096 else 1, min(n, 4096), C, generator=g, device=dev, dtype=torch.float16)
        n = h.shape[0] * h.shape[1]
        D = 4 * C
        with torch.no_grad():
            vpatch.norm_feed_forward_residual(norm, ff, h)          # packs the weights
            proj, out = ff.net[0].proj, ff.net[2]
            w1, b1 = proj.__dict__["_vtm_packed"]["geglu"][1]
            w2, b2 = out.__dict__["_vtm_packed"]["rows"][1]
            xp = _lib.layernorm_panels(h, norm.weight, norm.bias, norm.eps)
            hp = _lib.ff_geglu(xp, n, w1, D, b1)
            xn = _lib.layernorm(h, norm.weight, norm.bias, norm.eps)
            p8 = proj(xn)
            gg = _lib.geglu(p8)
            rows = [
                ("layernorm_panels      ", lambda: _lib.layernorm_panels(h, norm.weight, norm.bias, norm.eps), 0.0, 2.0 * n * C * 2),
                ("ff_geglu (GEMM1+act)  ", lambda: _lib.ff_geglu(xp, n, w1, D, b1), 2.0 * n * C * 2 * D, n * (C + D) * 2.0),
                ("linear_panels (GEMM2) ", lambda: _lib.linear_panels(hp, n, w2, C, b2, resid=h.view(n, C)), 2.0 * n * D * C, n * (D + 2 * C) * 2.0),
                ("panels: whole ff      ", lambda: vpatch.norm_feed_forward_residual(norm, ff, h), 2.0 * n * C * 3 * D, 0.0),
                ("blas: layernorm       ", lambda: _lib.layernorm(h, norm.weight, norm.bias, norm.eps), 0.0, 2.0 * n * C * 2),
                ("blas: GEMM1 (C -> 8C) ", lambda: proj(xn), 2.0 * n * C * 2 * D, n * (C + 2 * D) * 2.0),
                ("blas: vtm_geglu       ", lambda: _lib.geglu(p8), 0.0, n * 3 * D * 2.0),
                ("blas: GEMM2 + residual", lambda: out(gg) + h, 2.0 * n * D * C, 0.0),
                ("blas: whole ff        ", lambda: vpatch.feed_forward(ff, _lib.layernorm(h, norm.weight, norm.bias, norm.eps)) + h,
                 2.0 * n * C * 3 * D, 0.0)]
            for name, fn, fl, by in rows:
                REP = 4
                med, best = timeit(lambda: [fn() for _ in range(REP)], a.iters)
                med, best = med / REP, best / REP
                print(f"ff n={n} C={C} {name}: median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us"
                      + (f"  {fl / med / 1e9:7.1f} TFLOP/s" if fl else "") + (f"  {by / med / 1e6:7.0f} GB/s" if by else ""))
    elif a.what == "sites":
        # one hot-path pass over all 16 block sites for the BASELINE.json configurations other than the bench's cfg-2
        # (steady state: anchors populated by a preceding chunk): ms per chunk-step
        import vidtome_amd
        from vidtome_amd import sites
        cfgs = {"cfg2": (sites.sd15_sites(), 2, 16, (64, 64), 0.5, 0.5, False),
                "cfg3": (sites.sd15_sites(), 3, 16, (64, 64), 0.5, 0.5, True),       # PnP batch 3, align_batch
                "cfg4": (sites.sd15_sites(), 2, 8, (64, 64), 0.5, 0.5, False),        # one of the 8 chunks of 8 frames
                "cfg5": (sites.sd21_sites(), 2, 16, (96, 96), 0.6, 0.6, False)}       # SD-2.1-768, ratio 0.6
        sl, Bc, Fc, latent, lr, gr, align = cfgs[a.shape]
        unet = sites.SiteUNet(sl, seed=0).to(device=dev, dtype=torch.float16)
        vidtome_amd.apply_patch(unet, local_merge_ratio=lr, merge_global=True, global_merge_ratio=gr, batch_size=Bc,
                                align_batch=align)
        unet.set_size(latent)
        torch.manual_seed(123)
        # bench.py's regime: 3 chunks of one clip rotate, anchor chain re-seeded like a step of 8 chunks (sites.ClipStream)
        stream = sites.ClipStream(unet, sl, Bc, Fc, latent, torch.float16, dev)
        stream.populate()
        chunk = [0]

        def one_pass():
            stream.step(chunk[0])
            chunk[0] += 1
        for _ in range(2):
            one_pass()
        med, best = timeit(one_pass, max(a.iters, 7))
        print(f"sites {a.shape}: B={Bc} F={Fc} latent={latent} local={lr} global={gr} align={align}: median {med:.2f} ms per "
              f"chunk-step ({1e3 / med:.2f} steps/s), best {best:.2f} ms  [3 rotating chunks, 7-update anchor chain]")
    else:
        raise SystemExit("unknown benchmark")


if __name__ == "__main__":
    main()
