# same-box A/B of one environment knob on the whole step: tools/ab_env.sh VAR=VALUE  (alternates off/on three times)
run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-backward 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for i in 1 2 3; do run A=1; run "$@"; done
