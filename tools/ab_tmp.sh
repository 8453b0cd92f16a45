set -e
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], '|', round(d['value'],1),'QPS', round(d['ms_per_step'],4),'ms/step  kernel', round(d['roofline']['avg_launch_ms'],4),'ms frac', round(d['roofline']['frac'],4), 'fallbacks', d['config']['exact_fallbacks'])"; }
run --steps 300 --warmup 20 --opt scan_mirror=1
run --steps 300 --warmup 20 --rows 20000000 --dtype f16 --metric l2
run --steps 300 --warmup 20 --rows 20000000 --dtype bf16 --metric ip
run --steps 1000 --warmup 100 --rows 1000000 --opt scan_mirror=1
run --steps 300 --warmup 20 --rows 20000000 --dim 384
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "16bit or fp32_dims or mirror or sizes" 2>&1 | tail -3
