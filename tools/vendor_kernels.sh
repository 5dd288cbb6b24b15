# Run ON THE GPU BOX: kernel trace of tools/bench_gemm256.py (vendor calibration column included) - which kernel shapes /
# resources the vendor library picks for the step's hot GEMM shapes.  A study aid only: nothing of it is on the product path.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_vendor
rm -rf $OUT && mkdir -p $OUT
SHAPES="${SHAPES:-fc1,d->dh,down half,gate_up live,vit qkv}" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_gemm256.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections, statistics
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"], r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("Scratch_Size"))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/summary.txt", "w") as o:
    for k, v in sorted(agg.items(), key=lambda x: -sum(x[1]))[:40]:
        o.write(f"{statistics.median(v):8.1f}us n={len(v):4d} grid={k[1]} wg={k[2]} lds={k[3]} vgpr={k[4]} agpr={k[5]} scr={k[6]} {k[0][:220]}\n")
print(open("$OUT/summary.txt").read())
PY
