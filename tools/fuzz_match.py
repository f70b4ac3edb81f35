#!/usr/bin/env python3
"""Randomised check that vtm_match_filtered equals the exact fp32 matcher bit for bit, over random shapes, dtypes,
data regimes (iid, frame-correlated, duplicated rows, scaled), aligned / non-aligned batches -- unseeded, and with a random
SEED description (vtm_match_filtered_seeded: frame length, pool split, position tables with valid, random, missing and
out-of-range entries; the guesses may be anything, the result may not change).

    python tools/fuzz_match.py [--cases 300] [--seed 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vidtome_amd import _lib as L  # noqa: E402


def run(cases: int, seed: int, verbose: bool = True) -> int:
    """Returns the number of mismatching cases."""
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    fails = 0
    for case in range(cases):
        B = ri(1, 3)
        C = [8, 24, 64, 160, 320, 640, 1280][ri(0, 6)]
        big = ri(0, 9) == 0
        Ns = ri(1, 9000 if big else 1500)
        Nd = ri(1, 9000 if big else 1500)
        dtype = [torch.float16, torch.float32, torch.bfloat16][ri(0, 2)]
        align = bool(ri(0, 1))
        regime = ri(0, 3)
        x = torch.randn(B, Ns + Nd, C, generator=g)
        if regime == 1:      # strongly correlated tokens: many near-maximal scores
            x = torch.randn(B, 1, C, generator=g) + 0.03 * x
        elif regime == 2:    # duplicated dst rows and src copies of them
            for _ in range(ri(1, 5)):
                j = Ns + ri(0, Nd - 1)
                k = ri(1, min(40, Nd))
                s0 = Ns + ri(0, Nd - k)
                x[:, s0:s0 + k] = x[:, j:j + 1]
                x[:, ri(0, Ns - 1)] = x[:, j]
        elif regime == 3:    # wild scales per row
            x = x * torch.exp(3.0 * torch.randn(B, Ns + Nd, 1, generator=g))
        x = x.to(dtype).to(dev)
        ra = torch.arange(Ns, dtype=torch.int32, device=dev).expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=dev).expand(B, Nd).contiguous()
        a_op, _ = L.normalize_gather(x, None, ra)
        b_op, _ = L.normalize_gather(x, None, rb)
        exact = L.match(a_op, b_op, Ns, Nd, align)
        got, flag = L.match_filtered(x, None, ra, rb, align, want_flag=True)
        ok = bool(torch.equal(got, exact))
        # a random seed description: frames of Nf tokens, rows below `seed_L` are (frame, position) rows, the rest carry a
        # position in pos1 (two-part pools only), the table maps a position to any dst index or to nothing
        Nf = ri(1, max(1, min(Ns, Nd)))
        kind = ri(0, 3)
        table = None
        if kind == 1:
            table = torch.randint(0, Nd, (B, Nf), generator=g, dtype=torch.int32).to(dev)
        elif kind >= 2:
            table = torch.randint(-5, Nd + 20, (B, Nf), generator=g, dtype=torch.int32).to(dev)
        seed_L = ri(0, Ns + Nd) if kind == 3 else Ns + Nd
        split = ri(0, 1) == 1 and Ns + Nd > 1
        if split:      # pool = x0 | x1 with positions for the x1 rows
            P0 = ri(1, Ns + Nd - 1)
            x0, x1 = x[:, :P0].contiguous(), x[:, P0:].contiguous()
            pos1 = torch.randint(-2, Nf + 3, (B, Ns + Nd - P0), generator=g, dtype=torch.int32).to(dev)
            sgot = L.match_filtered(x0, x1, ra, rb, align, seed=(Nf, min(seed_L, P0), pos1, table))
        else:
            sgot = L.match_filtered(x, None, ra, rb, align, seed=(Nf, seed_L, None, table))
        if not torch.equal(sgot, exact):
            ok = False
            print("SEEDED MISMATCH", dict(Nf=Nf, kind=kind, seed_L=seed_L, split=split, bad=int((sgot != exact).sum())))
        if not ok:
            fails += 1
            print("MISMATCH", dict(case=case, B=B, C=C, Ns=Ns, Nd=Nd, dtype=str(dtype), align=align, regime=regime,
                                   flags=flag.tolist(), bad=int((got != exact).sum())))
    if verbose:
        print(f"{cases} cases, {fails} mismatches")
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    return 1 if run(a.cases, a.seed) else 0


if __name__ == "__main__":
    raise SystemExit(main())
