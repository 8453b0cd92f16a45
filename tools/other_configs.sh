#!/bin/bash
# The other workload shapes of the measurement table, one bench line each (GPU box, repository root):
#   bash tools/other_configs.sh <out.jsonl>
out=${1:-gpurun_out/other_configs.jsonl}
: > "$out"
run() { echo "== $*" >&2; timeout -k 10 400 python bench.py --cpu-seconds 0 --no-secondary "$@" 2>/dev/null | grep '^{' >> "$out" || exit 1; }
run --streams 1
run --opt scan_mirror=1 --steps 300 --warmup 20
run --rows 1250000 --steps 1000 --warmup 100
run --rows 1250000 --steps 1000 --warmup 100 --opt scan_mirror=1
run --rows 1000000 --steps 1000 --warmup 100 --opt scan_mirror=1
run --nq 1024 --steps 8 --warmup 3
run --nq 1024 --steps 8 --warmup 3 --metric l2
run --nq 1024 --steps 4 --warmup 2 --opt gemm32_split=1
run --nq 1024 --steps 3 --warmup 1 --opt gemm32_split=0
run --nq 1024 --steps 8 --warmup 3 --opt gemm_pp=0
run --nq 256 --steps 10 --warmup 3
run --nq 128 --steps 20 --warmup 3
run --nq 8 --steps 20 --warmup 3
run --nq 2 --steps 20 --warmup 3
run --rows 12500000 --dim 128 --dtype bf16 --metric ip --k 100 --steps 1000 --warmup 100
run --rows 20000000 --dtype f16 --metric l2 --nq 256 --steps 8 --warmup 3
run --rows 125000000 --dtype f16 --metric l2 --nq 256 --steps 4 --warmup 2
echo done >&2
