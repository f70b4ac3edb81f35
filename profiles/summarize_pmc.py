#!/usr/bin/env python3
"""Turn the PMC passes of one GPU session (tools/gpu_session.sh TAG -> gpurun_out/TAG/pmc/*.csv) into
profiles/rNN_pmc_counters.json and profiles/rNN_pmc_traffic.json:   python profiles/summarize_pmc.py gpurun_out/r02m r02_m

Counter units: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so hbm_read_bytes = 2 * FETCH_SIZE * 1024; SQ_* wave counters are quad-cycles."""
import collections
import csv
import json
import os
import sys

base, tag = sys.argv[1].rstrip("/") + "/pmc/", sys.argv[2]


def counters(fname, kern):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(base + fname)):
        if kern in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}, (len(next(iter(d.values()))) if d else 0)


def duration_us(fname, kern):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(base + fname))
               if kern in r["Kernel_Name"])
    return d[len(d) // 2]


sq = {"_how": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES "
              "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -- python tools/kbench.py {attn --Mq 34816 --M 52224 "
              "--d 40 | match --shape top_l1} (tools/gpu_session.sh); per-launch averages; cycles = GRBM_GUI_ACTIVE / 8 XCDs; "
              "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); valu_active = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / "
              "(1024 x cycles); profiled launches run ~6 % slower than un-profiled ones"}
traffic = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python "
                   "tools/kbench.py ... (tools/gpu_session.sh); KiB; hbm_read_bytes = 2 * FETCH_SIZE * 1024 (gfx950), WRITE_SIZE matched the "
                   "written bytes exactly in the three gather-path kernels (calibration)"}
# (round 6: the d = 40 launch of the bench is attention16s_kernel -- csrc/attention16.hip -- since VTM_ATT16 defaults to on)
for nm, kern, label, alg in (("attn", "attention16s_kernel", "attention16s_kernel<half,40> B=2 h=8 Mq=34816 Mk=52224", 2 * (34816 + 52224) * 320 * 2 * 2),
                             ("match", "filter_kernel", "filter_kernel top_l1 (B=2 Ns=49152 Nd=16384 C=320)", 83886080),
                             # round 3: the GEGLU projection of the cfg-2 top site (131 072 tokens, C = 320 -> 2 x 1280, gated
                             # activation in the epilogue): panels in (84 MB + 1.6 MB of weights), panels out (336 MB)
                             ("ff", "panel_gemm_kernel<__half, 0>", "panel_gemm_kernel<half,GEGLU> n=131072 K=320 D=1280", 131072 * (320 + 1280) * 2 + 2560 * 320 * 2)):
    if not os.path.exists(base + f"sq_{nm}_counter_collection.csv"):
        continue
    c, n = counters(f"sq_{nm}_counter_collection.csv", kern)
    us = duration_us(f"sq_{nm}_kernel_trace.csv", kern)
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    f, _ = counters(f"FETCH_SIZE_{nm}_counter_collection.csv", kern)
    w, _ = counters(f"WRITE_SIZE_{nm}_counter_collection.csv", kern)
    hbm = int(2 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024)
    sq[kern] = {"launches": n, "median_us": round(us, 1), "clock_GHz": round(cyc / us / 1e3, 3),
                "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 3),
                "valu_active_frac_of_simd_cycles": round(4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3),
                "wait_any_frac_of_wave": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                "wait_inst_any_frac_of_wave": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                "active_inst_any_frac_of_wave": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                "raw": {k: int(v) for k, v in c.items()}}
    traffic[f"{label} ({tag})"] = {"FETCH_SIZE_KiB": round(f["FETCH_SIZE"]), "WRITE_SIZE_KiB": round(w["WRITE_SIZE"]),
                                   "hbm_bytes_per_launch": hbm, "algorithmic_bytes": alg}
gp = {"_working_set": "cfg-5 top block geometry (SD-2.1-768: L = 16 x 9216 tokens, C = 320) at batch 4: 377.5 MB per (B, L, C) fp16 tensor -- "
                      "beyond the 256 MB Infinity Cache, unlike the cfg-2 tensors (84 MB)"}
for what, kern, alg_mb in (("layernorm", "layernorm", 755.0), ("gather", "gather_rows_kernel", 400.1), ("unmerge", "unmerge_add_kernel", 955.1)):
    if not os.path.exists(base + f"FETCH_SIZE_{what}_counter_collection.csv"):
        continue
    f, _ = counters(f"FETCH_SIZE_{what}_counter_collection.csv", kern)
    w, _ = counters(f"WRITE_SIZE_{what}_counter_collection.csv", kern)
    us = duration_us(f"FETCH_SIZE_{what}_kernel_trace.csv", kern)
    rd, wr = 2 * f["FETCH_SIZE"] * 1024 / 1e6, w["WRITE_SIZE"] * 1024 / 1e6
    gp[kern] = {"FETCH_SIZE_KiB": round(f["FETCH_SIZE"]), "WRITE_SIZE_KiB": round(w["WRITE_SIZE"]), "hbm_read_MB": round(rd, 1),
                "hbm_write_MB": round(wr, 1), "algorithmic_MB": alg_mb, "median_us": round(us, 1),
                "hbm_GBps": round((rd + wr) / us * 1e3), "frac_of_8TBps": round((rd + wr) / us * 1e3 / 8000, 3)}
traffic["gather_path"] = gp
rnd = tag.split("_")[0]
here = os.path.dirname(os.path.abspath(__file__))       # (the session script runs this from /tmp)
json.dump(sq, open(os.path.join(here, f"{rnd}_pmc_counters.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(here, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in sq.items() if k != "_how"}, indent=1)[:1500])
print(json.dumps(gp, indent=1))
