#!/usr/bin/env python3
"""gpurun_out/prof_<round> (tools/profile_step.sh) -> profiles/<round>_*  (<round> = $VLA_ROUND, default r03; "{RND}" below).

--stage box  (on the GPU box, where the multi-100-MB kernel trace lives): reduce the trace to
             gpurun_out/prof_r02/kernel_summary.csv = one row per (kernel, grid, workgroup) with calls, calls per step,
             average / total microseconds - small enough to travel back.
--stage repo (default, in the repository): copy the summaries into profiles/, derive
             profiles/{RND}_gemm_in_situ.json = per-GEMM-instantiation calls/step, FLOPs/launch, average microseconds,
             sum of in-situ GEMM time per step (the denominator of bench.py's roofline.frac_in_situ) and
             profiles/{RND}_gemm_traffic_pmc.json (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md says).
"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("VLA_ROUND", "r04")                       # profiles are named per round
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_" + RND), os.path.join(ROOT, "profiles")
GEMM_KERNELS = ("gemm_nt_kernel", "gemm256_kernel", "gemm_tn_kernel", "gemm_tn256_kernel", "gemm_tn_grouped_kernel", "gemm_tn256_grouped_kernel")


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


def _write_summary(path, agg, steps):
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(path, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "grid_x", "wg_x", "calls", "calls_per_step", "avg_us", "total_us", "us_per_step", "steps_in_trace"])
        for (k, g, wg), (n, us) in rows:
            w.writerow([k, g, wg, n, round(n / steps, 2), round(us / n, 2), round(us, 1), round(us / steps, 1), steps])
    return rows


def newest(pattern):
    """The most recent match (gpurun MERGES a call's files into the local gpurun_out/: earlier collections' raw traces stay beside the new ones)."""
    f = glob.glob(pattern)
    return max(f, key=os.path.getmtime) if f else None


def stage_box():
    """Two summaries: the whole trace (weight init, capture warm-ups, the recording step included: per-step columns divide by
    every executed step) and kernel_summary_steady.csv = ONLY the launches between bench.py's two marker launches, i.e. the K
    timed graph replays - the step's own kernels and nothing else."""
    f = newest(os.path.join(SRC, "stats", "*", "*_kernel_trace.csv"))
    bench = json.load(open(os.path.join(SRC, "bench_under_rocprof.json")))
    steps, timed, mgrid = bench["executed_steps"], bench["steps"], str(bench.get("marker_grid_x", -1))
    rows = list(csv.DictReader(open(f)))
    grid_of = lambda r: r.get("Grid_Size_X", r.get("Grid_Size", ""))
    marks = sorted(int(r["Start_Timestamp"]) for r in rows if "fill_zero_kernel" in r["Kernel_Name"] and grid_of(r) == mgrid)
    agg, steady = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    for r in rows:
        key = (short(r["Kernel_Name"]), grid_of(r), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[key]
        a[0] += 1
        a[1] += us
        if len(marks) == 2 and marks[0] < int(r["Start_Timestamp"]) < marks[1]:
            a = steady[key]
            a[0] += 1
            a[1] += us
    allrows = _write_summary(os.path.join(SRC, "kernel_summary.csv"), agg, steps)
    tot = sum(v[1] for v in agg.values())
    print(f"{len(allrows)} (kernel, grid) groups, {tot / steps / 1e3:.2f} ms of kernel time per step over {steps} steps (whole trace)")
    if len(marks) == 2:
        srows = _write_summary(os.path.join(SRC, "kernel_summary_steady.csv"), steady, timed)
        tot = sum(v[1] for v in steady.values())
        foreign = sorted({k for (k, _, _) in steady if "at::" in k or "rocclr" in k or "Cijk" in k})
        print(f"timed region: {len(srows)} groups, {tot / timed / 1e3:.2f} ms of kernel time per step over {timed} steps; "
              f"kernels that are not this library's: {foreign if foreign else 'none'}")
        for (k, g, wg), (n, us) in srows[:25]:
            print(f"{k[:64]:64s} grid {g:>8s} calls/step {n / timed:7.1f} avg {us / n:8.1f} us  {us / timed / 1e3:6.2f} ms/step")
    else:
        print(f"no marker pair found (grid {mgrid}): steady summary not written")


def family_of(kernel_name: str) -> str:
    return "gemm256_kernel" if "gemm256_kernel" in kernel_name else "gemm_tn_kernel" if "gemm_tn" in kernel_name else "gemm_nt_kernel"


def pmc_sum(sub, counter, by_family=None):
    """Sum of `counter` over the GEMM dispatches of the PMC pass `sub` (+ per kernel family into by_family[family] = [sum, launches]).
    One dispatch may appear on several rows (one per XCD / counter instance): launches are counted by dispatch id."""
    f = newest(os.path.join(SRC, sub, "*", "*_counter_collection.csv"))
    if not f:
        return None, 0
    tot, seen = 0.0, set()
    for r in csv.DictReader(open(f)):
        if any(g in r["Kernel_Name"] for g in GEMM_KERNELS) and r["Counter_Name"] == counter:
            v = float(r["Counter_Value"])
            tot += v
            did = r.get("Dispatch_Id", len(seen))
            new = did not in seen
            seen.add(did)
            if by_family is not None:
                e = by_family.setdefault(family_of(r["Kernel_Name"]), [0.0, 0])
                e[0] += v
                e[1] += int(new)
    return tot, len(seen)


def stage_repo():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "kernel_summary.csv"), os.path.join(DST, RND + "_kernel_summary.csv"))
    steady = os.path.join(SRC, "kernel_summary_steady.csv")
    if os.path.exists(steady):
        shutil.copy(steady, os.path.join(DST, RND + "_kernel_summary_steady.csv"))
    st = newest(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st, os.path.join(DST, RND + "_kernel_stats.csv"))
    for n in ("bench.json", "bench_under_rocprof.json"):
        if os.path.exists(os.path.join(SRC, n)):
            shutil.copy(os.path.join(SRC, n), os.path.join(DST, RND + "_" + n))
    have_steady = os.path.exists(os.path.join(DST, RND + "_kernel_summary_steady.csv"))
    rows = list(csv.DictReader(open(os.path.join(DST, RND + "_kernel_summary_steady.csv" if have_steady else RND + "_kernel_summary.csv"))))
    gem = [r for r in rows if any(g in r["kernel"] for g in GEMM_KERNELS)]
    steps = int(rows[0]["steps_in_trace"])
    bench = json.load(open(os.path.join(DST, RND + "_bench_under_rocprof.json")))
    out = {"source": "rocprofv3 --kernel-trace of `bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe` "
                     "(tools/profile_step.sh)" + (": the launches between the two marker launches that bracket the timed graph replays "
                                                  f"(profiles/{RND}_kernel_summary_steady.csv)" if have_steady else ": whole trace"),
           "kernels_not_from_this_library_in_timed_region": (sorted({r["kernel"] for r in rows if "at::" in r["kernel"] or "rocclr" in r["kernel"]})
                                                             if have_steady else None),
           "steps_in_trace": steps,
           "source_digest": bench.get("source_digest"),       # of the code that was profiled (bench.py stamps it; flops.source_digest)
           "gemm_us_per_step_in_situ": round(sum(float(r["total_us"]) for r in gem) / steps, 1),
           "gemm_launches_per_step": round(sum(int(r["calls"]) for r in gem) / steps, 1),
           "all_kernels_us_per_step": round(sum(float(r["total_us"]) for r in rows) / steps, 1),
           "gemm_flops_per_step": bench.get("gemm_flops_per_step"),
           "instantiations": [dict(kernel=r["kernel"], grid_x=r["grid_x"], calls_per_step=float(r["calls_per_step"]), avg_us=float(r["avg_us"]),
                                   us_per_step=float(r["us_per_step"])) for r in gem]}
    if out["gemm_flops_per_step"]:
        out["gemm_tflops_in_situ"] = round(out["gemm_flops_per_step"] / out["gemm_us_per_step_in_situ"] / 1e6, 1)
        out["frac_in_situ"] = round(out["gemm_tflops_in_situ"] / 2500.0, 4)
    json.dump(out, open(os.path.join(DST, RND + "_gemm_in_situ.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "instantiations"}, indent=1))
    ff, fw = {}, {}
    fetch, n1 = pmc_sum("pmc_fetch", "FETCH_SIZE", ff)      # KB
    write, n2 = pmc_sum("pmc_write", "WRITE_SIZE", fw)
    if fetch is not None and n1 == n2 and n1 > 0:
        # per kernel family: fabric-side bytes per launch against the algorithmic bytes per launch of the SAME family (bench.py's
        # recording step tags every GEMM call with the kernel it routes to: bench_under_rocprof.json "gemm_families"); the PMC passes
        # run `--steps 2 --warmup 1 --eager` + the capture-free recording step = 4 executed steps
        fams, alg = {}, bench.get("gemm_families") or {}
        for fam in sorted(set(ff) | set(fw)):
            fb, nf = ff.get(fam, [0.0, 0])
            wb, _ = fw.get(fam, [0.0, 0])
            e = {"launches": nf, "fabric_bytes_per_launch": round((2.0 * fb + wb) * 1024 / max(1, nf))}
            if fam in alg and alg[fam]["launches"]:
                e["algorithmic_bytes_per_launch"] = round(alg[fam]["algorithmic_bytes"] / alg[fam]["launches"])
                e["launches_per_step"] = alg[fam]["launches"]
                e["traffic_over_algorithmic"] = round(e["fabric_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 3)
            fams[fam] = e
        pm = {"kernels": list(GEMM_KERNELS), "launches": n1, "fetch_size_kb_sum": fetch, "write_size_kb_sum": write, "fetch_correction": 2.0,
              "bytes_per_launch": (2.0 * fetch + write) * 1024 / n1, "by_kernel_family": fams, "source_digest": bench.get("source_digest"),
              "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_step.sh) over bench.py --steps 2 "
                      "--warmup 1 --eager; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"}
        json.dump(pm, open(os.path.join(DST, RND + "_gemm_traffic_pmc.json"), "w"), indent=1)
        print(json.dumps(pm, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="repo", choices=["box", "repo"])
    a = ap.parse_args()
    stage_box() if a.stage == "box" else stage_repo()
