set -e
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], '|', round(d['value'],1),'QPS', round(d['ms_per_step'],4),'ms/step  kernel', round(d['roofline']['avg_launch_ms'],4),'ms frac(fp32 bytes)', round(d['roofline']['frac'],4), 'fallbacks', d['config']['exact_fallbacks'], d['config']['queries_counted'], d['host_latency_ms'])"; }
run --steps 400 --warmup 20 --opt scan_mirror=1
run --steps 400 --warmup 20 --opt scan_mirror=1 --metric l2
run --steps 400 --warmup 20 --opt scan_mirror=1 --metric ip
run --steps 1000 --warmup 50 --rows 1000000 --opt scan_mirror=1
run --steps 200 --warmup 20 --opt scan_mirror=1 --k 100
