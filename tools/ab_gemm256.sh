# same-box A/B of the GEMM routing on the whole step (only compare numbers from ONE gpurun call)
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-backward 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run A=auto
run VLA_NO_GEMM256=1
run VLA_NO_LLM_SPLIT=1
run VLA_NO_LLM_SPLIT=1 VLA_NO_GEMM256=1
run A=auto
run VLA_NO_GEMM256=1
