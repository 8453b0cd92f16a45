# Single-query configurations at a glance (run on the GPU box): bash tools/scan_ab.sh [bench.py options]
set -e
run() { timeout -k 10 120 python bench.py --cpu-seconds 0 --no-secondary --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], '|', round(d['value'],1),'QPS', round(d['ms_per_step'],4),'ms/step  scan', round(d['roofline']['avg_launch_ms'],4),'ms frac', round(d['roofline']['frac'],4), 'fallbacks', d['config']['exact_fallbacks'])"; }
run --rows 1000000 "$@"
run --rows 12500000 --dim 128 --dtype bf16 --metric ip --k 100 "$@"
run --rows 1250000 "$@"
run --rows 10000000 "$@"
