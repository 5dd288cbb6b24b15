# same-box A/B of whole-step variants (boxes differ by +-5-10 %: only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-backward 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run A=1
run VLA_VIS_AFTER=0
run VLA_VIS_AFTER=2
run VLA_VIS_AFTER=4
run VLA_VIS_AFTER=6
run A=1
