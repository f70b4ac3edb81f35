#!/usr/bin/env python3
"""How many keys does duplicate-key folding remove?  Runs bench.py's chunk stream (cfg-2, same seeds, same re-seeding) for a few
passes and prints, per top-level block and pass: which side the local chunk was on, the distinct queries, and -- when the block's
anchors carried content ids -- surviving keys / keys.      python tools/diag/fold_stats.py [--data REGIME] [--passes N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vidtome_amd  # noqa: E402
from vidtome_amd import patch as vpatch, sites  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--passes", type=int, default=14)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    unet = sites.SiteUNet(sites.sd15_sites(), seed=0).to(device=dev, dtype=torch.float16)
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=2, target_stride=4,
                            global_rand=0.5)
    unet.set_size((64, 64))
    torch.manual_seed(123)
    stream = sites.ClipStream(unet, sites.sd15_sites(), 2, 16, (64, 64), torch.float16, dev, n_sets=3, chunks_per_step=8,
                              regime=a.data)
    plans = []
    orig = vpatch.compute_merge

    def rec(module, x, info, **kw):
        res = orig(module, x, info, **kw)
        plans.append(getattr(res[0], "plan", None))
        return res
    vpatch.compute_merge = rec
    stream.populate()
    tot_keys = tot_kept = with_ids = launches = 0
    for p in range(a.passes):
        plans.clear()
        stream.step(p)
        torch.cuda.synchronize()
        line = []
        for plan in plans:
            if plan is None or plan.global_level is None or plan.x_joined.shape[2] != 320:
                continue
            launches += 1
            side = "src" if plan.local_chunk == 0 else "dst"
            q = "-" if plan.q_count is None else str(int(plan.q_count.sum()))
            if plan._key_fold is not None:
                kc = int(plan._key_fold[2].sum())
                keys = plan.M * plan.x_joined.shape[0]
                tot_keys += keys
                tot_kept += kc
                with_ids += 1
                line.append(f"{side} q={q} keys {kc}/{keys} (-{100 * (1 - kc / keys):.1f}%)")
            else:
                line.append(f"{side} q={q} no-ids")
        print(f"pass {p:2d}: " + " | ".join(line), flush=True)
    print(f"top-level launches {launches}, with ids {with_ids}, keys folded away {100 * (1 - tot_kept / max(1, tot_keys)):.2f}% of those launches' keys")


if __name__ == "__main__":
    main()
