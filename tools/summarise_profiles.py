#!/usr/bin/env python3
"""gpurun_out/prof_final (tools/collect_profiles.sh) -> profiles/r01_final_*: kernel stats of the bench command, the bench
line, and HBM bytes per GEMM launch from the two PMC passes (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md says)."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_final"), os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(DST, "r01_final_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, "r01_final_bench.json"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, "r01_final_bench_under_rocprof.json"))


def pmc_sum(sub, counter):
    f = glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv"))[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "gemm_nt_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch, n1 = pmc_sum("pmc_fetch", "FETCH_SIZE")      # KB
write, n2 = pmc_sum("pmc_write", "WRITE_SIZE")
assert n1 == n2 and n1 > 0, (n1, n2)
out = {"kernel": "gemm_nt_kernel", "launches": n1, "fetch_size_kb_sum": fetch, "write_size_kb_sum": write, "fetch_correction": 2.0,
       "bytes_per_launch": (2.0 * fetch + write) * 1024 / n1,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh) over bench.py --steps 2 "
               "--warmup 1 --eager; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"}
json.dump(out, open(os.path.join(DST, "r01_gemm_traffic_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
rows = list(csv.DictReader(open(os.path.join(DST, "r01_final_kernel_stats.csv"))))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%")
