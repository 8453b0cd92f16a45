out=gpurun_out/k_sweep.jsonl
: > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary "$@" 2>/dev/null | grep '^{' >> "$out" || exit 1; }
for k in 10 40 50 100; do
run --k $k --steps 60 --warmup 10
done
run --k 100 --steps 60 --warmup 10 --opt scan_waves=4
run --k 100 --steps 60 --warmup 10 --opt scan_blocks_per_cu=2
run --k 100 --steps 60 --warmup 10 --opt scan_sample=4096
run --k 100 --steps 60 --warmup 10 --streams 1
run --k 10 --steps 60 --warmup 10 --streams 1
python - <<'PY'
import json
for l in open('gpurun_out/k_sweep.jsonl'):
    j=json.loads(l); r=j['roofline']
    print(j['config']['workload'][:40], j['config']['options'], j['config']['streams'], 'qps %.1f'%j['value'], 'ms/step %.4f'%j['ms_per_step'], 'launch %.4f'%r['avg_launch_ms'], 'frac %.4f'%r['frac'])
PY
