#!/usr/bin/env python3
"""Sweep the dst-split count of the filtered matcher (VTM_DEBUG_NSPLIT hook) over the cfg-2 shapes.
    python tools/sweep_nsplit.py [--shuffle] [--kp] [--regime NAME]           (needs an MI355X)
--shuffle: the src and the dst rows are handed over in a random order (what level 2 and the global level look like in a real
pass: the merged sequence is sorted by similarity rank, a row's matches are scattered over the dst range) instead of position order;
--kp: sweep the pruning depth (VTM_DEBUG_KP) at the default split count instead; --seed: with the same-position seeds of the
product path (dst index = position in the first dst frame: meaningful for the position-ordered rows only, i.e. not with --shuffle).
The library reads its VTM_DEBUG_* hooks ONCE per process (round 6), so every setting runs in a child process of its own
(--child SHAPE, the hook in its environment); the parent compares the children's result hashes with the default's."""
import hashlib
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidtome_amd import _lib  # noqa: E402

SHAPES = {"top_l1": (2, 49152, 16384, 320), "top_l2": (2, 12288, 28672, 320), "top_g": (2, 34816, 34816, 320),
          "mid_l1": (2, 12288, 4096, 640), "mid_l2": (2, 3072, 7168, 640), "mid_g": (2, 8704, 8704, 640)}


def timeit(fn, iters=7):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def child(shape):
    """One (shape, hook setting) measurement: prints `<sha256 of the result> <median us>`."""
    shuffle = "--shuffle" in sys.argv
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (B, Ns, Nd, C) in SHAPES.items():
        # (every shape's draws are consumed in order so that a child sees the tokens the one-process sweep saw)
        base = torch.randn(B, Nd, C, generator=g, device="cuda")
        idx = torch.arange(Ns + Nd, device="cuda") % Nd
        x = (base[:, idx] + 0.1 * torch.randn(B, Ns + Nd, C, generator=g, device="cuda")).half()
        perm = None
        if shuffle:
            perm = ([torch.randperm(Ns, generator=g, device="cuda") for _ in range(B)],
                    [torch.randperm(Nd, generator=g, device="cuda") for _ in range(B)])
        if name != shape:
            continue
        if "--regime" in sys.argv:       # the bench's token regimes (sites.DATA_REGIMES) after a LayerNorm, frames of N tokens
            from vidtome_amd import sites
            regime = sys.argv[sys.argv.index("--regime") + 1]
            N = 4096
            while Ns % N or Nd % N:
                N //= 2
            xx = sites.regime_tokens(regime, B, (Ns + Nd) // N, N, C, torch.Generator().manual_seed(0))
            x = torch.nn.functional.layer_norm(xx, (C,)).reshape(B, Ns + Nd, C).half().cuda()
        ra = torch.arange(Ns, dtype=torch.int32, device="cuda").expand(B, Ns).contiguous()
        rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device="cuda").expand(B, Nd).contiguous()
        if shuffle:
            ra = torch.stack(perm[0]).to(torch.int32).contiguous()
            rb = (Ns + torch.stack(perm[1])).to(torch.int32).contiguous()
        seed = None
        if "--seed" in sys.argv:
            Nf = 4096
            while Ns % Nf or Nd % Nf:
                Nf //= 2
            seed = (Nf, Ns + Nd, None, None)
        res = _lib.match_filtered(x, None, ra, rb, False, seed=seed)
        us = timeit(lambda: _lib.match_filtered(x, None, ra, rb, False, seed=seed)) * 1e3
        print(hashlib.sha256(res.cpu().numpy().tobytes()).hexdigest(), f"{us:.0f}")


def run_child(shape, hook=None, value=None):
    env = {k: v for k, v in os.environ.items() if k not in ("VTM_DEBUG_NSPLIT", "VTM_DEBUG_KP")}
    if hook:
        env[hook] = str(value)
    args = [a for a in sys.argv[1:] if a != "--kp"]
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", shape] + args, env=env, check=True,
                         capture_output=True, text=True).stdout.split()
    return out[-2], out[-1]


def main():
    if "--child" in sys.argv:
        return child(sys.argv[sys.argv.index("--child") + 1])
    kp = "--kp" in sys.argv
    for name, (B, Ns, Nd, C) in SHAPES.items():
        ref, us = run_child(name)
        out = [f"default {us}"]
        for v in (range(0, C // 64) if kp else range(2, 15)):
            h, us = run_child(name, "VTM_DEBUG_KP" if kp else "VTM_DEBUG_NSPLIT", v)
            assert h == ref, (name, v)
            out.append(f"{'kp' if kp else ''}{v}: {us}")
        print(f"{name:7s} us  " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
