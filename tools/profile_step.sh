# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace + stats of the DEFAULT step alone (no CPU baseline, no full-backward
# variant, no GEMM probe replays: every kernel in the trace belongs to a training step), then the two PMC passes (separate runs,
# as MI355X_MICROARCH.md prescribes) and an un-profiled bench line.  tools/summarise_profiles.py turns gpurun_out/prof_<round> into
# profiles/<round>_* (round = $VLA_ROUND, default r04).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${VLA_ROUND:-r04}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
if [ -z "$SKIP_PMC" ]; then
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-full-backward --no-probe > /dev/null 2> $OUT/pmc_fetch.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-full-backward --no-probe > /dev/null 2> $OUT/pmc_write.log
fi
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log
python3 tools/summarise_profiles.py --stage box
ls $OUT
