#!/usr/bin/env python3
"""gpurun_out/prof_r02 (tools/profile_step.sh) -> profiles/r02_*.

--stage box  (on the GPU box, where the multi-100-MB kernel trace lives): reduce the trace to
             gpurun_out/prof_r02/kernel_summary.csv = one row per (kernel, grid, workgroup) with calls, calls per step,
             average / total microseconds - small enough to travel back.
--stage repo (default, in the repository): copy the summaries into profiles/, derive
             profiles/r02_gemm_in_situ.json = per-GEMM-instantiation calls/step, FLOPs/launch, average microseconds,
             sum of in-situ GEMM time per step (the denominator of bench.py's roofline.frac_in_situ) and
             profiles/r02_gemm_traffic_pmc.json (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md says).
"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_r02"), os.path.join(ROOT, "profiles")
GEMM_KERNELS = ("gemm_nt_kernel", "gemm256_kernel")


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


def stage_box():
    f = glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_trace.csv"))[0]
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    steps = json.load(open(os.path.join(SRC, "bench_under_rocprof.json")))["executed_steps"]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(os.path.join(SRC, "kernel_summary.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "grid_x", "wg_x", "calls", "calls_per_step", "avg_us", "total_us", "us_per_step", "steps_in_trace"])
        for (k, g, wg), (n, us) in rows:
            w.writerow([k, g, wg, n, round(n / steps, 2), round(us / n, 2), round(us, 1), round(us / steps, 1), steps])
    tot = sum(v[1] for v in agg.values())
    print(f"{len(rows)} (kernel, grid) groups, {tot / steps / 1e3:.2f} ms of kernel time per step over {steps} steps")
    for (k, g, wg), (n, us) in rows[:25]:
        print(f"{k[:64]:64s} grid {g:>8s} calls/step {n / steps:7.1f} avg {us / n:8.1f} us  {us / steps / 1e3:6.2f} ms/step")


def pmc_sum(sub, counter):
    f = glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv"))
    if not f:
        return None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if any(g in r["Kernel_Name"] for g in GEMM_KERNELS) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def stage_repo():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "kernel_summary.csv"), os.path.join(DST, "r02_kernel_summary.csv"))
    st = glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(DST, "r02_kernel_stats.csv"))
    for n in ("bench.json", "bench_under_rocprof.json"):
        if os.path.exists(os.path.join(SRC, n)):
            shutil.copy(os.path.join(SRC, n), os.path.join(DST, "r02_" + n))
    rows = list(csv.DictReader(open(os.path.join(DST, "r02_kernel_summary.csv"))))
    gem = [r for r in rows if any(g in r["kernel"] for g in GEMM_KERNELS)]
    steps = int(rows[0]["steps_in_trace"])
    bench = json.load(open(os.path.join(DST, "r02_bench_under_rocprof.json")))
    out = {"source": "rocprofv3 --kernel-trace of `bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe` "
                     "(tools/profile_step.sh): every kernel in the trace belongs to a default training step",
           "steps_in_trace": steps,
           "gemm_us_per_step_in_situ": round(sum(float(r["total_us"]) for r in gem) / steps, 1),
           "gemm_launches_per_step": round(sum(int(r["calls"]) for r in gem) / steps, 1),
           "all_kernels_us_per_step": round(sum(float(r["total_us"]) for r in rows) / steps, 1),
           "gemm_flops_per_step": bench.get("gemm_flops_per_step"),
           "instantiations": [dict(kernel=r["kernel"], grid_x=r["grid_x"], calls_per_step=float(r["calls_per_step"]), avg_us=float(r["avg_us"]),
                                   us_per_step=float(r["us_per_step"])) for r in gem]}
    if out["gemm_flops_per_step"]:
        out["gemm_tflops_in_situ"] = round(out["gemm_flops_per_step"] / out["gemm_us_per_step_in_situ"] / 1e6, 1)
        out["frac_in_situ"] = round(out["gemm_tflops_in_situ"] / 2500.0, 4)
    json.dump(out, open(os.path.join(DST, "r02_gemm_in_situ.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "instantiations"}, indent=1))
    fetch, n1 = pmc_sum("pmc_fetch", "FETCH_SIZE")      # KB
    write, n2 = pmc_sum("pmc_write", "WRITE_SIZE")
    if fetch is not None and n1 == n2 and n1 > 0:
        pm = {"kernels": list(GEMM_KERNELS), "launches": n1, "fetch_size_kb_sum": fetch, "write_size_kb_sum": write, "fetch_correction": 2.0,
              "bytes_per_launch": (2.0 * fetch + write) * 1024 / n1,
              "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_step.sh) over bench.py --steps 2 "
                      "--warmup 1 --eager; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"}
        json.dump(pm, open(os.path.join(DST, "r02_gemm_traffic_pmc.json"), "w"), indent=1)
        print(json.dumps(pm, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="repo", choices=["box", "repo"])
    a = ap.parse_args()
    stage_box() if a.stage == "box" else stage_repo()
