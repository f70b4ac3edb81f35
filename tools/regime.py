#!/usr/bin/env python3
"""What the global level sees in the two bench regimes, and how many DISTINCT queries the top block would need.

    python tools/regime.py [--passes 7] [--same-chunk]

Per pass and merging site kind (top N=4096/C=320, mid N=1024/C=640) of the cfg-2 step:
  * coin outcome (local chunk = src or dst side of the global level, patch.py:62-71);
  * pairs the refine pass evaluated (flags_out[3] of vtm_match_filtered), overflow rows (flags_out[2]);
  * exact node_max ties at the global level (groups of equal similarity);
  * local-is-src passes: r, the number of DISTINCT matched global dst rows and the histogram of how many local src rows
    merged into the same dst row -- every dst row that several local tokens merged into is computed as a query several
    times by the live-query attention (MergePlan.q_rows has one entry per local token).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vidtome_amd  # noqa: E402
from vidtome_amd import _lib, merge, sites  # noqa: E402
from vidtome_amd import patch as vpatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=7)
    ap.add_argument("--same-chunk", action="store_true")
    ap.add_argument("--chunks", type=int, default=3)
    ap.add_argument("--chunks-per-step", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, F, LAT = 2, 16, (64, 64)
    sl = [s for s in sites.sd15_sites() if s.name in ("down0.0", "down1.0")]
    unet = sites.SiteUNet(sl, seed=0).to(device=dev, dtype=torch.float16)
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
    unet.set_size(LAT)
    torch.manual_seed(123)
    stream = sites.ClipStream(unet, sl, B, F, LAT, torch.float16, dev, n_sets=a.chunks, chunks_per_step=a.chunks_per_step,
                              same_chunk=a.same_chunk)
    stream.populate()

    flags = []
    orig_mf = _lib.match_filtered

    def mf(x0, x1, ar, br, align, want_flag=False, seed=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        best, flag = orig_mf(x0, x1, ar, br, align, want_flag=True, seed=seed)
        e1.record()
        flags.append((ar.shape[1], br.shape[1], flag, e0, e1))
        return best
    _lib.match_filtered = mf
    plans = []
    orig_cm = vpatch.compute_merge

    def cm(module, x, info, **kw):
        kw["want_indices"] = True
        res = orig_cm(module, x, info, **kw)
        plans.append(res[0].plan)
        return res
    vpatch.compute_merge = cm

    out = []
    for p in range(a.passes):
        flags.clear(), plans.clear()
        stream.step(p)
        torch.cuda.synchronize()
        fl = [(ns, nd, f.cpu().tolist() + [round(e0.elapsed_time(e1), 3)]) for ns, nd, f, e0, e1 in flags]
        fi = 0
        for site, plan in zip(sl, plans):
            nlev = len(plan.levels) + (plan.global_level is not None)
            lev = fl[fi:fi + nlev]
            fi += nlev
            rec = {"pass": p, "site": site.name, "levels": [{"Ns": ns, "Nd": nd, "pairs": f[3], "overflow_rows": f[2],
                                                           "all_exact": f[0], "ms": f[4]} for ns, nd, f in lev]}
            gl = plan.global_level
            if gl is not None:
                nm, _ = _lib.decode_best(gl.best)
                ties = []
                for b in range(nm.shape[0]):
                    v = nm[b].sort().values
                    ties.append(int((v[1:] == v[:-1]).sum()))
                rec["global"] = {"local_is_src": plan.local_chunk == 0, "Ns": gl.Ns, "Nd": gl.Nd, "r": gl.r,
                                 "tied_node_max_pairs": ties}
                if plan.local_chunk == 0:
                    d = []
                    for b in range(gl.dst_idx.shape[0]):
                        cnt = torch.bincount(gl.dst_idx[b].long(), minlength=gl.Nd)
                        hist = torch.bincount(cnt[cnt > 0])
                        d.append({"distinct_dst": int((cnt > 0).sum()), "max_mult": int(cnt.max()),
                                  "hist_mult_1_to_8": hist[1:9].tolist()})
                    rec["global"]["dst_multiplicity"] = d
                    Mq = plan.q_rows.shape[1]
                    rec["global"]["queries_now"] = Mq
                    rec["global"]["queries_distinct"] = [Mq - gl.r + x["distinct_dst"] for x in d]
            out.append(rec)
            print(json.dumps(rec), flush=True)
    _lib.match_filtered = orig_mf
    vpatch.compute_merge = orig_cm


if __name__ == "__main__":
    main()
