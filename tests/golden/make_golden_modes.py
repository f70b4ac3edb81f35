#!/usr/bin/env python3
"""The reference's non-"replace" merge modes (vidtome/merge.py:127-131, 431-435) -> tests/golden/modes.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_modes.py        (build container only: imports /root/reference)

`compute_merge` never passes another mode than "replace", but the closures `bipartite_soft_matching_randframe` /
`bipartite_soft_matching_2s` return accept `mode=` ("sum", "prod", "mean", "amax", "amin": torch.scatter_reduce's reductions,
include_self=True), so the closure protocol includes them.  Each case stores the tokens, the reference's
index arrays (read out of the closure cells) and `merge(x, mode=...)` for every mode, for fp32 tokens and for fp16 / bf16
tokens (whose results are stored as the 16-bit patterns).  Inputs are screened like the other fixtures (fp32 == fp64
indices); duplicated destinations are the point, so ratios are high and the dst side is small."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)

ref_merge = mg.ref_merge
MODES = ("sum", "prod", "mean", "amax", "amin")


def bits16(t):
    return t.view(torch.int16).numpy().copy()


def main():
    out, n = {}, 0
    cases = [
        # kind, B, N (or src_len + dst_len), F / src_len, C, ratio, unm_pre, align
        ("randframe", 2, 4 * 24, 4, 16, 0.9, 0, False),
        ("randframe", 3, 8 + 4 * 10, 4, 24, 0.6, 8, True),
        ("randframe", 2, 8 * 12, 8, 40, 0.75, 0, False),
        ("2s", 2, 60 + 20, 60, 16, 0.9, 0, False),
        ("2s", 3, 30 + 50, 30, 8, 0.5, 0, True),
    ]
    for ci, (kind, B, N, F, C, ratio, unm_pre, align) in enumerate(cases):
        for attempt in range(200):
            g = torch.Generator().manual_seed(1000 * ci + attempt)
            x = torch.randn(B, N, C, generator=g)
            x[:, 3] = float("nan") if ci == 2 else x[:, 3]          # one NaN token: amax / amin propagate it
            res = {}
            for dt in (torch.float32, torch.float64):
                if kind == "randframe":
                    m, u, ret = ref_merge.bipartite_soft_matching_randframe(x.to(dt), F, ratio, unm_pre, mg.fork_generator(123), 4, align)
                else:
                    m, u, ret = ref_merge.bipartite_soft_matching_2s(x.to(dt), F, ratio, align)
                c = mg.cells(m)
                res[dt] = (m, {k: mg.np64(c[k])[..., 0] for k in ("unm_idx", "src_idx", "dst_idx")})
            if all(np.array_equal(res[torch.float32][1][k], res[torch.float64][1][k]) for k in res[torch.float32][1]):
                break
        else:
            raise RuntimeError(f"case {ci} could not be screened")
        m32, idx = res[torch.float32]
        randf = int(torch.randint(0, min(4, F), torch.Size([1]), generator=mg.fork_generator(123))) if kind == "randframe" else -1
        out.update({f"{n}/kind": kind, f"{n}/B": B, f"{n}/N": N, f"{n}/F": F, f"{n}/C": C, f"{n}/ratio": ratio,
                    f"{n}/unm_pre": unm_pre, f"{n}/align": align, f"{n}/randf": randf, f"{n}/x": x.numpy()})
        for k, v in idx.items():
            out[f"{n}/{k}"] = v.astype(np.int32)
        # how often a destination is hit (the interesting part of these modes)
        d = idx["dst_idx"][0]
        out[f"{n}/max_sources_per_dst"] = int(np.bincount(d).max()) if d.size else 0
        for mode in MODES:
            out[f"{n}/f32/{mode}"] = m32(x, mode=mode).numpy()
        # 16-bit tokens: the SAME matching (the closure's indices come from the fp32 metric), values in the 16-bit dtype
        for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
            xl = x.to(dt)
            for mode in MODES:
                out[f"{n}/{name}/{mode}"] = bits16(m32(xl, mode=mode))
        print(f"case {n}: {kind} B={B} N={N} C={C} r={idx['src_idx'].shape[1]} max sources per dst {out[f'{n}/max_sources_per_dst']}"
              f" (attempt {attempt})", flush=True)
        n += 1
    out["n_cases"] = np.array(n)
    path = os.path.join(HERE, "modes.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
