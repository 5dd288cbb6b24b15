# Run ON THE GPU BOX (gpurun): kernel-trace stats of the bench command + the two PMC passes (separate runs, as the
# microarch guide prescribes), everything under gpurun_out/prof_final; tools/summarise_profiles.py turns it into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-backward > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-full-backward > /dev/null 2> $OUT/pmc_fetch.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-full-backward > /dev/null 2> $OUT/pmc_write.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.log
ls $OUT/*
