# same-box A/B of the gemm256 walk on the whole step (only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-backward --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run A=persistent-tickets
run VLA_GEMM256_GRID=0
run VLA_GEMM256_STATIC=1
run A=persistent-tickets
run VLA_GEMM256_GRID=0
run VLA_GEMM256_STATIC=1
