# Run ON THE GPU BOX: kernel trace of one bench.py mode (MODE = full | lora, ARGS = extra bench.py arguments), top kernels by total time
# of the whole process (capture warm-ups included) -> gpurun_out/prof_$TAG/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG:-mode}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 700 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --mode ${MODE:-lora} --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-probe $ARGS > $OUT/bench.json 2> $OUT/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    k = (n, r.get("Grid_Size_X"))
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
with open("$OUT/summary.txt", "w") as o:
    o.write(f"total kernel time {tot/1e3:.1f} ms (whole process: ${STEPS:-6} timed + 2 warm-up + 2 capture warm-up + 1 capture steps)\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:60]:
        o.write(f"{v[1]/1e3:8.2f} ms  n={v[0]:5d} avg {v[1]/v[0]:8.1f} us  grid {k[1]:>9s}  {k[0]}\n")
print(open("$OUT/summary.txt").read())
PY
rm -rf $OUT/*/ 2>/dev/null; find $OUT -name "*.csv" -size +5M -delete
tail -2 $OUT/bench.json | cut -c1-400
