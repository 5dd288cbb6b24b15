"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: find the graph-replayed steps (adamw kernels delimit them) and
print, for the last timed step, the busy time per queue and a coarse phase timeline (which kernels ran when)."""
import csv
import sys
from collections import defaultdict

f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
print("adamw launches:", len(adam))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
# a step = kernels between two consecutive adamw launches (graph replays: vision .. adamw(prev) .. rest)
a, b = adam[which - 1], adam[which]
step = rows[a:b]
t0, t1 = step[0]["s"], step[-1]["e"]
print(f"window between adamw launches: {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, span {(rs[-1]['e'] - rs[0]['s']) / 1e6:.3f} ms")
mainq = max(byq, key=lambda q: sum(r["e"] - r["s"] for r in byq[q]))
# idle gaps on the main queue
rs = byq[mainq]
gaps = [(rs[i + 1]["s"] - rs[i]["e"], i) for i in range(len(rs) - 1)]
tot_gap = sum(g for g, _ in gaps if g > 0)
print(f"main queue {mainq}: idle inside span {tot_gap / 1e6:.3f} ms; largest gaps:")
for g, i in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:8.1f} us after {rs[i]['Kernel_Name'][:60]} @ {(rs[i]['e'] - t0) / 1e6:.3f} ms -> {rs[i + 1]['Kernel_Name'][:50]}")
# phase timeline: 1 ms buckets, per queue top kernel
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:28]
nb = int((t1 - t0) / 1e6) + 1
for k in range(nb):
    lo, hi = t0 + k * 1_000_000, t0 + (k + 1) * 1_000_000
    line = f"{k:3d} ms |"
    for q, rs in byq.items():
        acc = defaultdict(int)
        for r in rs:
            ov = min(r["e"], hi) - max(r["s"], lo)
            if ov > 0:
                acc[short(r["Kernel_Name"])] += ov
        busy = sum(acc.values())
        top = sorted(acc.items(), key=lambda x: -x[1])[:2]
        line += f" q{q}: {busy / 1e4:5.1f}% " + ",".join(f"{n}:{v // 1000}" for n, v in top).ljust(60) + "|"
    print(line)
