# Run ON THE GPU BOX: fabric-side traffic of gemm256_kernel with the XCD-blocked tile order against the plain group-M order
# (VLA_GEMM256_XCD=0), FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kernel-trace only), tools/pmc_gemm256.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_gemm256_ab
rm -rf $OUT && mkdir -p $OUT
for order in blocked plain; do
  if [ $order = blocked ]; then export VLA_GEMM256_XCD=1; else unset VLA_GEMM256_XCD; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${order}_$c -- python3 tools/pmc_gemm256.py > $OUT/${order}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, json, os
out = {}
for order in ("blocked", "plain"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/prof_gemm256_ab/{order}_{c}/*/*_counter_collection.csv")[0]
        by = {}
        for r in csv.DictReader(open(f)):
            if "gemm256_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        vals = [by[k] for k in sorted(by)]
        out[f"{order}_{c}_kb_per_launch"] = [round(sum(vals[3 * i:3 * i + 3]) / 3, 1) for i in range(len(vals) // 3)]
json.dump(out, open("gpurun_out/prof_gemm256_ab/traffic_ab.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/*_SIZE
