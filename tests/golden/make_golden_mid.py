#!/usr/bin/env python3
"""Mid- and full-size matcher fixtures from the REFERENCE (vidtome/merge.py), stored as sha256 of the index arrays ->
tests/golden/planted_mid.npz.  Run in the build container only (imports /root/reference):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mid.py [--set cfg14]

`--set cfg14` (round 4) writes tests/golden/planted_cfg14.npz instead: the EXACT level shapes of BASELINE.json's cfg-1
(4 frames 256 x 256, local merging only: 3 072 x 1 024 x 320 at the top blocks, 768 x 256 x 640 at the mid blocks) and of one
of cfg-4's eight 8-frame chunks (level 1: 24 576 x 8 192 x 320, level 2 with the carried-over unmerged tokens: 4 096 x
16 384, global level 18 432^2; mid blocks 6 144 x 2 048 x 640 and 4 608^2), same format.

Round 1's fixtures stop at 64 tokens per frame / C <= 40 (and ONE full-size case, the planted cfg-2 local level 1).  These
cover the sizes where the 128 x 256 tiles of the HIP matcher are fully populated, for the LOCAL matcher with and without
carried-over unmerged tokens (level 2 shape) and for the GLOBAL matcher (`bipartite_soft_matching_2s`, both
`unmerge_chunk` values, square and rectangular, `align_batch` both), up to the full cfg-2 global-level sizes
8 704^2 x 640 (mid blocks) and 34 816^2 x 320 (top blocks).

At these sizes random inputs cannot pin indices bit for bit (the reference's own result depends on
its BLAS's summation order once adjacent similarity values get closer than a few ulp), so the inputs are PLANTED
(inputs.planted_batch: every src row has one well-separated best dst row and the row maxima are strictly spaced, also
across the samples of an aligned batch); each case is kept only if the reference gives the same indices in fp32 and fp64
wherever fp64 is affordable, and its fp64 margins (adjacent sorted maxima, top-1 / top-2) are recorded.
Inputs are regenerated from seeds by tests/golden/inputs.py (numpy only), nothing but hashes + 16-entry heads is stored.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from vidtome import merge as ref_merge  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from inputs import idx_sha, planted_batch, planted_local_chunk  # noqa: E402

torch.set_grad_enabled(False)


def cells(fn):
    out = {n: c.cell_contents for n, c in zip(fn.__code__.co_freevars, fn.__closure__)}
    if "split" in out:
        out.update(cells(out["split"]))
    return out


def idx_of(m):
    c = cells(m)
    return {n: c[n].detach().numpy()[..., 0].astype(np.int32) for n in ("unm_idx", "src_idx", "dst_idx")}


def fork_generator(seed):
    torch.manual_seed(seed)
    return torch.Generator(device="cpu").set_state(torch.get_rng_state())


LOCAL = [
    # name, B, F, tnum, unm_pre, C, ratio, align, seed  (+ optional: check in fp64, default True)
    ("local_f8_n256_c320", 2, 8, 256, 0, 320, 0.5, False, 11),
    ("local_f16_n1024_c640_aligned", 2, 16, 1024, 0, 640, 0.5, True, 12),
    ("local_l2_f4_n1024_u6144_c640", 2, 4, 1024, 6144, 640, 0.5, False, 13),       # cfg-2 mid block, level 2
    ("local_l2_f4_n256_u1536_c320_aligned", 3, 4, 256, 1536, 320, 0.6, True, 14),
    ("local_f8_n1024_c320_r09", 2, 8, 1024, 0, 320, 0.9, False, 15),
    # cfg-5 (SD-2.1-768, 16 x 9 216 tokens, ratio 0.6): the largest level of any BASELINE configuration, one sample
    # (110 592 x 36 864 scores = 16.3 GB in fp32: no fp64 run)
    ("local_cfg5_l1_110592x36864_c320", 1, 16, 9216, 0, 320, 0.6, False, 16, False),
    # cfg-3 (PnP: batch 3, aligned matching) at the full top-block level-1 size: ONE matching over the three samples'
    # concatenated scores (49 152 x 3 * 16 384)
    ("local_cfg3_l1_aligned_b3_49152x16384_c320", 3, 16, 4096, 0, 320, 0.5, True, 17, False),
]
GLOBAL = [
    # name, B, src_len, dst_len, C, ratio, align, unmerge_chunk, seed, check in fp64
    ("global_2176_c320", 2, 2176, 2176, 320, 0.5, False, 0, 21, True),
    ("global_8704_c640_chunk0", 2, 8704, 8704, 640, 0.5, False, 0, 22, True),         # cfg-2 mid global level
    ("global_8704_c640_chunk1", 2, 8704, 8704, 640, 0.5, False, 1, 23, True),
    ("global_8704_c640_aligned", 2, 8704, 8704, 640, 0.5, True, 0, 24, True),
    ("global_rect_8704x4352_c640", 2, 8704, 4352, 640, 0.8, False, 1, 25, True),
    ("global_rect_4352x8704_c320_aligned", 3, 4352, 8704, 320, 0.6, True, 0, 26, True),
    ("global_34816_c320", 1, 34816, 34816, 320, 0.5, False, 0, 27, False),            # cfg-2 top global level (B = 1)
    ("global_cfg5_64513_c320", 1, 64513, 64513, 320, 0.6, False, 1, 28, False),       # cfg-5 top global level: ragged, 16.6 GB of scores
]


LOCAL_CFG14 = [
    ("local_cfg1_top_f4_n1024_c320", 2, 4, 1024, 0, 320, 0.5, False, 31),            # cfg-1 top block: its ONLY level
    ("local_cfg1_mid_f4_n256_c640", 2, 4, 256, 0, 640, 0.5, False, 32),
    ("local_cfg4_l1_f8_n4096_c320", 2, 8, 4096, 0, 320, 0.5, False, 33),             # 24 576 x 8 192
    ("local_cfg4_l2_f2_n4096_u12288_c320", 2, 2, 4096, 12288, 320, 0.5, False, 34),  # 4 096 x 16 384
    ("local_cfg4_mid_l1_f8_n1024_c640", 2, 8, 1024, 0, 640, 0.5, False, 35),         # 6 144 x 2 048
    ("local_cfg4_mid_l2_f2_n1024_u3072_c640", 2, 2, 1024, 3072, 640, 0.5, False, 36),
]
GLOBAL_CFG14 = [
    ("global_cfg4_18432_c320_chunk0", 2, 18432, 18432, 320, 0.5, False, 0, 37, True),
    ("global_cfg4_18432_c320_chunk1", 2, 18432, 18432, 320, 0.5, False, 1, 38, False),
    ("global_cfg4_mid_4608_c640", 2, 4608, 4608, 640, 0.5, False, 1, 39, True),
]


def margins64(a, b, align):
    """fp64: (min gap between adjacent sorted row maxima, min top-1 / top-2 gap)."""
    a = torch.from_numpy(a).double()
    b = torch.from_numpy(b).double()
    a = a / a.norm(dim=-1, keepdim=True)
    b = b / b.norm(dim=-1, keepdim=True)
    sc = a @ b.transpose(-1, -2)
    if align:
        sc = torch.cat([*sc], dim=-1)[None]
    top2 = sc.topk(2, dim=-1).values
    nm = top2[..., 0].sort(dim=-1).values
    return float((nm[..., 1:] - nm[..., :-1]).min()), float((top2[..., 0] - top2[..., 1]).min())


def main():
    cfg14 = "--set" in sys.argv and sys.argv[sys.argv.index("--set") + 1] == "cfg14"
    local_cases, global_cases = (LOCAL_CFG14, GLOBAL_CFG14) if cfg14 else (LOCAL, GLOBAL)
    out = {}
    n = 0
    for name, B, F, tnum, unm_pre, C, ratio, align, seed, *rest in local_cases:
        check64 = rest[0] if rest else True
        t0 = time.time()
        gen = fork_generator(123)
        randf = int(torch.randint(0, min(4, F), torch.Size([1]), generator=fork_generator(123)))
        x = planted_local_chunk(B, F, tnum, unm_pre, C, randf, seed)
        res = {}
        for dt in (torch.float32, torch.float64) if check64 else (torch.float32,):
            g = fork_generator(123)
            m, u, ret = ref_merge.bipartite_soft_matching_randframe(torch.from_numpy(x).to(dt), F, ratio, unm_pre, g, 4, align)
            res[dt] = idx_of(m)
        assert not check64 or all(np.array_equal(res[torch.float32][k], res[torch.float64][k]) for k in res[torch.float32]), name
        idx = res[torch.float32]
        out.update({f"{n}/kind": "local", f"{n}/name": name, f"{n}/B": B, f"{n}/F": F, f"{n}/tnum": tnum,
                    f"{n}/unm_pre": unm_pre, f"{n}/C": C, f"{n}/ratio": ratio, f"{n}/align": align, f"{n}/seed": seed,
                    f"{n}/randf": randf, f"{n}/unm_num": ret["unm_num"]})
        for k, v in idx.items():
            out[f"{n}/{k}_sha256"] = idx_sha(v)
            out[f"{n}/{k}_head"] = v[..., :16]
            out[f"{n}/{k}_shape"] = np.array(v.shape)
        print(f"{name}: randf {randf}, r {idx['src_idx'].shape[-1]}, {time.time() - t0:.1f} s", flush=True)
        n += 1
    for name, B, sl, dl, C, ratio, align, chunk, seed, check64 in global_cases:
        t0 = time.time()
        a, b = planted_batch(sl, dl, C, seed, B)
        x = np.concatenate([a, b], axis=1)
        m, u, ret = ref_merge.bipartite_soft_matching_2s(torch.from_numpy(x), sl, ratio, align, unmerge_chunk=chunk)
        idx = idx_of(m)
        g1 = g2 = -1.0
        if check64:
            m64, _, _ = ref_merge.bipartite_soft_matching_2s(torch.from_numpy(x).double(), sl, ratio, align, unmerge_chunk=chunk)
            i64 = idx_of(m64)
            assert all(np.array_equal(idx[k], i64[k]) for k in idx), name
            g1, g2 = margins64(a, b, align)
        # the unmerge slice the reference returns (merge.py:459): its length pins unmerge_chunk
        y = torch.zeros(B, (sl - idx["src_idx"].shape[-1]) + dl, 1)
        out.update({f"{n}/kind": "global", f"{n}/name": name, f"{n}/B": B, f"{n}/src_len": sl, f"{n}/dst_len": dl, f"{n}/C": C,
                    f"{n}/ratio": ratio, f"{n}/align": align, f"{n}/unmerge_chunk": chunk, f"{n}/seed": seed,
                    f"{n}/unm_num": ret["unm_num"], f"{n}/unmerged_len": u(y).shape[1], f"{n}/gap_sorted_max": g1,
                    f"{n}/gap_top2": g2})
        for k, v in idx.items():
            out[f"{n}/{k}_sha256"] = idx_sha(v)
            out[f"{n}/{k}_head"] = v[..., :16]
            out[f"{n}/{k}_shape"] = np.array(v.shape)
        print(f"{name}: r {idx['src_idx'].shape[-1]}, fp64 gaps {g1:.2e} / {g2:.2e}, {time.time() - t0:.1f} s", flush=True)
        n += 1
    out["n_cases"] = np.array(n)
    path = os.path.join(HERE, "planted_cfg14.npz" if cfg14 else "planted_mid.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, n, "cases", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
