#!/usr/bin/env python3
"""Block outputs of the REFERENCE for the DEFAULT (fp16) path of vidtome_amd -> tests/golden/chain16_*.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chain16.py        (build container only: imports /root/reference)

Rounds 1-3 held the reference-recorded block outputs (chain_*.npz) only against an fp32 model, which takes the library-GEMM
projection path; the fp16 model's own path (gather-fed projection GEMMs, live / compacted queries, panel GEMMs) was checked
against the oracle and against attention.npz, never against a recorded BLOCK output of the reference.

These fixtures close that: the reference's `apply_patch` + `ToMeBlock.forward` + `sa_forward` run -- in fp32, the CPU path
north_star names as the oracle -- on a stand-in UNet with head dims / channel counts the HIP attention and projection
kernels are instantiated for (d = 40, 64, 80; C = 160, 128, 320), whose weights, hidden states AND norm1 outputs lie on the
fp16 grid: the model an fp16 run holds (norm1 = LayerNorm computed in fp32, rounded once -- what vtm_layernorm computes,
tested against fp32 PyTorch on its own).  Everything downstream of norm1 is the unmodified reference in fp32.

Merge decisions on nearly tied similarities are not a property of the algorithm but of the last bit of its inputs, so a case
is SCREENED: kept only if no merge decision changes -- block outputs and anchors stay within 2e-3 of the output scale; a
changed decision shows as >= 3e-3, the roundings themselves as 2-7e-4 -- (a) when the reference runs in fp64 (summation
order; also moves a few norm1 roundings) and (b) when a random 2e-4 of norm1's fp16 outputs are moved by one ulp (three
different draws) -- the freedom another correct fp16 LayerNorm has.
tests/test_gpu_parity.py::test_default_fp16_path_vs_reference_chain holds the fp16 model's block outputs and anchors to 1e-3
of the output scale against the recorded run.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)

CFGS = [
    # d = 40, C = 160: gather-fed projections (vtm_linear_rows), live + compacted queries, both coin outcomes, an F = 1 chunk
    dict(name="chain16_cfg_f4_d40", B=2, C=160, heads=4, H=6, W=6, chunk_frames=[4, 4, 4, 1, 4, 4], reset_before=[3],
         local_ratio=0.5, merge_global=True, global_ratio=0.5, align=False, injection=None, t=500,
         rng_seed=2, data_seed=7001, frame_noise=0.6),
    # d = 64 (SD-2.1's head dim), PnP batch 3: aligned matching, shared-probability attention
    dict(name="chain16_pnp_f4_d64", B=3, C=128, heads=2, H=6, W=6, chunk_frames=[4, 4, 4],
         local_ratio=0.6, merge_global=True, global_ratio=0.6, align=True, injection=[500], t=500,
         rng_seed=6, data_seed=7002, frame_noise=0.6),
    # d = 80 (SD-1.5's mid blocks), two local levels (F = 8); C = 320 so that the test can also force the panel-GEMM
    # projection path (C % 64 == 0), which SD-1.5 takes at C >= 640
    dict(name="chain16_cfg_f8_d80", B=2, C=320, heads=4, H=4, W=4, chunk_frames=[8, 8, 3],
         local_ratio=0.5, merge_global=True, global_ratio=0.5, align=False, injection=None, t=500,
         rng_seed=3, data_seed=7003, frame_noise=0.6),
]


def disagreement(plain, other):
    worst = 0.0
    for a, b in zip(plain, other):
        for ra, rb in zip(a["records"], b["records"]):
            scale = max(1.0, float(ra["out"].abs().max()))
            worst = max(worst, float((ra["out"] - rb["out"].to(ra["out"].dtype)).abs().max()) / scale)
        for k, v in a["global_tokens"].items():
            if v is not None and k != "":
                w_ = b["global_tokens"][k].to(v.dtype)
                if w_.shape != v.shape:
                    return 1.0
                worst = max(worst, float((v - w_).abs().max()) / max(1.0, float(v.abs().max())))
    return worst


def main():
    only = sys.argv[1:]
    for base in CFGS:
        if only and base["name"] not in only:
            continue
        for attempt in range(400):
            cfg = dict(base, data_seed=base["data_seed"] + 1000 * attempt, fp16_grid=True, round_norm1=1)
            plain, w, names, rng_state = mg.run_chain(torch.float32, cfg, weights_seed=2024)
            worst = 0.0
            for kind in ("fp64", 2, 3, 4):
                other, _, _, _ = mg.run_chain(torch.float64 if kind == "fp64" else torch.float32,
                                              cfg if kind == "fp64" else dict(cfg, round_norm1=kind), weights_seed=2024)
                worst = max(worst, disagreement(plain, other))
                if worst >= 2e-3:
                    break
            if worst < 2e-3:
                print(cfg["name"], "kept after", attempt + 1, "attempts; worst disagreement of the screening runs:",
                      f"{worst:.2e}", flush=True)
                break
        else:
            raise RuntimeError("could not screen " + base["name"])
        data = {"cfg_json": json.dumps(cfg), "rng_state": rng_state.numpy(), "block_names": np.array(names),
                "screen_disagreement": np.array(worst)}
        for k, v in w.items():          # on the fp16 grid: stored as fp16
            assert torch.equal(v, v.half().float()), k
            data["w/" + k] = v.half().numpy()
        for ck, ch in enumerate(plain):
            data[f"c{ck}/latent_shape"] = np.array(ch["latent"].shape)
            for r in ch["records"]:
                bi = r["block"]
                hid = r["hidden"].numpy()
                assert np.array_equal(hid, hid.astype(np.float16).astype(np.float32))
                data[f"c{ck}/b{bi}/hidden"] = hid.astype(np.float16)
                data[f"c{ck}/b{bi}/out"] = r["out"].numpy()
            for k, v in ch["global_tokens"].items():
                if v is not None and k != "":
                    data[f"c{ck}/gt/{k}"] = v.numpy()
        path = os.path.join(HERE, cfg["name"] + ".npz")
        np.savez_compressed(path, **data)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()
