# same-box A/B of schedule / routing knobs with the round-2 kernels (only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for i in 1 2 3; do
run A=default
run VLA_PRIO=h
done
