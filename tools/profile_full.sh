# Run ON THE GPU BOX: kernel trace of the full fine-tune step (bench.py --mode full), top kernels by total time.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_full
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --mode ${MODE:-full} --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    k = (n, r.get("Grid_Size_X"))
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
with open("$OUT/summary.txt", "w") as o:
    o.write(f"total kernel time {tot/1e3:.1f} ms\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:45]:
        o.write(f"{v[1]/1e3:8.2f} ms  n={v[0]:5d} avg {v[1]/v[0]:8.1f} us  grid {k[1]:>9s}  {k[0]}\n")
print(open("$OUT/summary.txt").read())
PY
tail -2 $OUT/bench.json | cut -c1-600
