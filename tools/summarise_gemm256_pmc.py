#!/usr/bin/env python3
"""gpurun_out/prof_gemm256_pmc (tools/collect_gemm256_pmc.sh) -> profiles/r04_gemm256_pmc.json: per big step shape the raw counters
of gemm256_kernel (mean over its three launches) and the derived numbers north_star asks for - MFMA-busy share, wave wait share,
LDS bank-conflict share, fabric-side bytes against the algorithmic bytes (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md
prescribes).  --stage box reduces the counter CSVs to a small JSON on the GPU box; --stage repo copies it into profiles/."""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_gemm256_pmc")


def stage_box():
    meta = json.load(open(os.path.join(SRC, "meta.json")))
    # launches arrive in program order: 3 per shape; match gemm256 dispatches by order
    per_counter = defaultdict(list)                 # counter -> [values in dispatch order]
    for f in sorted(glob.glob(os.path.join(SRC, "p*", "*", "*_counter_collection.csv"))):
        rows = [r for r in csv.DictReader(open(f)) if "gemm256_kernel" in r["Kernel_Name"]]
        by = defaultdict(dict)
        for r in rows:
            by[int(r["Dispatch_Id"])][r["Counter_Name"]] = by[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for did in sorted(by):
            for c, v in by[did].items():
                per_counter[c].append(v)
    dur = []
    tf = glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_trace.csv"))
    if tf:
        rows = [r for r in csv.DictReader(open(tf[0])) if "gemm256_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    out = []
    for i, m in enumerate(meta):
        sl = slice(3 * i, 3 * i + 3)
        c = {k: (sum(v[sl]) / max(1, len(v[sl]))) for k, v in per_counter.items() if len(v) >= 3 * (i + 1)}
        d = dict(m)
        d["counters_mean_of_3_launches"] = {k: round(v, 1) for k, v in sorted(c.items())}
        if dur[sl]:
            d["avg_us"] = round(sum(dur[sl]) / len(dur[sl]), 2)
            d["tflops"] = round(m["flops"] / d["avg_us"] / 1e6, 1)
        g = lambda k: c.get(k)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
            # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (= 16 cycles x MFMA count for 16x16x32 bf16: checked
            # against mfma_per_launch); GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles -> share of SIMD-cycles the MFMA pipe is busy
            d["mfma_busy_share"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)
            d["mfma_busy_cycles_per_mfma"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / m["mfma_per_launch"], 2)
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
            d["wave_wait_any_share"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
            d["wave_issue_stall_share"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share_of_lds_cycles"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            d["fabric_bytes"] = round((2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)          # KB counters; FETCH doubled (gfx950)
            d["traffic_over_algorithmic"] = round(d["fabric_bytes"] / m["algorithmic_bytes"], 3)
        if d.get("avg_us") and g("GRBM_GUI_ACTIVE"):
            d["clock_ghz_from_gui_active"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / d["avg_us"] / 1e3, 3)    # (duration from the un-profiled trace)
        if d.get("avg_us"):
            # independent of the counters: MFMA pipe occupancy = MFMAs x 16 cycles (16x16x32 bf16: 8 passes) / (1024 SIMDs x cycles at 2.4 GHz)
            d["mfma_pipe_share_at_2p4ghz"] = round(m["mfma_per_launch"] * 16 / (1024 * d["avg_us"] * 2400.0), 4)
        out.append(d)
    import sys
    sys.path.insert(0, ROOT)
    from vla_adapter_amd import flops
    json.dump(dict(kernel="gemm256_kernel", source_digest=flops.source_digest(), note="rocprofv3 --pmc, one pass per counter group, kernel-trace only; means over three launches per shape; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md; SQ cycle counters in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES",
                   shapes=out), open(os.path.join(SRC, "gemm256_pmc.json"), "w"), indent=1)
    for d in out:
        print({k: v for k, v in d.items() if k != "counters_mean_of_3_launches"})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="repo", choices=["box", "repo"])
    if ap.parse_args().stage == "box":
        stage_box()
    else:
        os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
        d = json.load(open(os.path.join(SRC, "gemm256_pmc.json")))
        for sh in d["shapes"]:                        # derived shares recomputed here from the raw counters (the formulas may have moved on)
            c = sh["counters_mean_of_3_launches"]
            sh.pop("mfma_busy_over_sq_busy", None)
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
                sh["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
                sh["mfma_busy_cycles_per_mfma"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / sh["mfma_per_launch"], 2)
                if sh.get("avg_us"):
                    sh["clock_ghz_from_gui_active"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / sh["avg_us"] / 1e3, 3)
        json.dump(d, open(os.path.join(ROOT, "profiles", "r04_gemm256_pmc.json"), "w"), indent=1)
        for sh in d["shapes"]:
            print({k: v for k, v in sh.items() if k != "counters_mean_of_3_launches"})
