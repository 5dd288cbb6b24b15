# same-box A/B of whole-step variants (only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run A=new
run VLA_NARROW_SMALL=32
run VLA_NARROW_SMALL=128
run VLA_NARROW_SMALL=320
run A=new
run VLA_NO_SPLITK=1
