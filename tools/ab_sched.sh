# same-box A/B of schedule knobs with the round-2 kernels (only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for i in 1 2; do
run A=default
run VLA_FWD_CHUNKS=6,6,6,3,1,1,1
run VLA_FWD_CHUNKS=8,8,4,2,1,1
run VLA_FWD_CHUNKS=3,3,3,3,3,3,3,1,1,1
run VLA_FWD_CHUNKS=4,4,4,4,4,1,1,1,1
run VLA_FWD_CHUNKS=4,4,4,4,4,4
done
