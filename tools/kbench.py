#!/usr/bin/env python3
"""Kernel micro-benchmarks on the cfg-2 shapes (used while optimising; run under rocprofv3 for counters).

    python tools/kbench.py match  [--shape top_l1|top_l2|top_g|mid_l1] [--iters 5]
    python tools/kbench.py attn   [--M 52224 --d 40 --heads 8 --B 2] [--iters 3]
    python tools/kbench.py sort   [--n 49152]
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
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    if a.what == "match":
        B, Ns, Nd, C = SHAPES[a.shape]
        x = torch.randn(B, Ns + Nd, C, generator=g, device=dev, dtype=torch.float16)
        ra = torch.arange(Ns, dtype=torch.int32, device=dev).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=dev).expand(B, Nd).contiguous()
        aop, _ = _lib.normalize_gather(x, None, ra)
        bop, _ = _lib.normalize_gather(x, None, rb)
        med, best = timeit(lambda: _lib.match(aop, bop, Ns, Nd, a.align), a.iters)
        fl = 2.0 * B * Ns * Nd * C
        medf, bestf = timeit(lambda: _lib.match_filtered(x, None, ra, rb, a.align), a.iters)
        _, fl_ = _lib.match_filtered(x, None, ra, rb, a.align, want_flag=True)
        print("flags [whole-call exact, non-finite, overflow rows, -]:", fl_.tolist())
        print(f"match_filtered {a.shape}: median {medf:.3f} ms ({2.0 * B * Ns * Nd * C / medf / 1e9:.1f} algorithmic "
              f"TFLOP/s), best {bestf:.3f} ms")
        print(f"match {a.shape} B={B} Ns={Ns} Nd={Nd} C={C}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), "
              f"best {best:.3f} ms ({fl / best / 1e9:.1f} TFLOP/s)")
    elif a.what == "attn":
        B, M, h, d = a.B, a.M, a.heads, a.d
        C = h * d
        Mp = (M + 7) // 8 * 8
        qk = torch.randn(B, Mp, 2 * C, generator=g, device=dev, dtype=torch.float16)
        vt = torch.randn(B, C, Mp, generator=g, device=dev, dtype=torch.float16)
        if a.Mq:
            Mq = a.Mq
            q = torch.randn(B, (Mq + 7) // 8 * 8, C, generator=g, device=dev, dtype=torch.float16)
            med, best = timeit(lambda: _lib.attention_kv(q, qk[:, :, C:], vt, h, Mq, M, d ** -0.5), a.iters)
        else:
            Mq = M
            med, best = timeit(lambda: _lib.attention(qk[:, :, :C], qk[:, :, C:], vt, h, M, d ** -0.5, 1), a.iters)
        fl = 4.0 * B * Mq * M * C
        print(f"attention B={B} Mq={Mq} Mk={M} h={h} d={d}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), "
              f"best {best:.3f} ms ({fl / best / 1e9:.1f} TFLOP/s)")
    elif a.what == "sort":
        keys = torch.randint(0, 2 ** 62, (a.B, a.n), generator=g, device=dev, dtype=torch.int64)
        med, best = timeit(lambda: _lib.sort_desc(keys), a.iters)
        print(f"sort rows={a.B} n={a.n}: median {med * 1e3:.1f} us, best {best * 1e3:.1f} us")
    else:
        raise SystemExit("unknown benchmark")


if __name__ == "__main__":
    main()
