# Run ON THE GPU BOX: kernel trace of the batch-1 predict_action replays (tools/bench_inference.py, configuration index CFG = 0 | 1)
# -> gpurun_out/prof_predict$CFG/summary.txt: kernels by total time, with calls per replay (the process runs ~66 replays + 3 eager passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${CFG:-0}
OUT=gpurun_out/prof_predict$CFG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_inference.py $CFG > $OUT/bench.json 2> $OUT/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    k = (n, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
REPLAYS = 63 + 3      # 3 warm-up + 30 + 30 replays, + 2 capture warm-ups + 1 eager pass before them
tot = sum(v[1] for v in agg.values())
with open("$OUT/summary.txt", "w") as o:
    o.write(f"total kernel time {tot/1e3:.1f} ms over ~{REPLAYS} passes = {tot/1e3/REPLAYS:.2f} ms of kernel time per predict\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:70]:
        o.write(f"{v[1]/REPLAYS:8.1f} us/predict  n/predict={v[0]/REPLAYS:6.1f} avg {v[1]/v[0]:7.1f} us  grid {k[1]:>8s} x{k[2]:>4s} x{k[3]:>3s}  {k[0]}\n")
print(open("$OUT/summary.txt").read())
PY
rm -rf $OUT/*/ 2>/dev/null; find $OUT -name "*.csv" -size +5M -delete
cat $OUT/bench.json
