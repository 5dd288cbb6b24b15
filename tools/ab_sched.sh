# same-box A/B of schedule / routing knobs with the round-2 kernels (only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --mode full --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full $*', d['ms_per_step'])"; }
runa() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('adapter $*', d['ms_per_step'])"; }
for i in 1 2; do
run A=default
run VLA_GEMM256_MIN_TILES=160
run VLA_GEMM256_MIN_TILES=260
runa A=default
runa VLA_GEMM256_MIN_TILES=160
runa VLA_GEMM256_MIN_TILES=260
done
