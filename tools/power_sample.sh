#!/bin/bash
# Socket power and clocks while a workload runs (GPU box): the batched pass against the single-query scan.
#   bash tools/power_sample.sh > gpurun_out/power.log
sample() {   # $1 = label, rest = bench arguments
    label=$1; shift
    python3 bench.py --cpu-seconds 0 --no-secondary "$@" > /dev/null 2>&1 &
    pid=$!
    sleep 9      # table fill and warm-up
    for i in 1 2 3 4 5 6 7 8; do
        if ! kill -0 $pid 2>/dev/null; then break; fi
        echo "== $label sample $i"
        /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E -i "power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" | head -8
        sleep 0.7
    done
    wait $pid
}
echo "== idle"
/opt/rocm/bin/rocm-smi --showpower --showclocks 2>&1 | grep -E -i "power|sclk|mclk" | head -4
sample "batch 1024 fp32 (gemm_pp_kernel)" --nq 1024 --steps 1200 --warmup 3
sample "batch 256 fp16 L2 20M (gemm_pp384_kernel)" --rows 20000000 --dtype f16 --metric l2 --nq 256 --steps 1800 --warmup 3
sample "single query scan" --steps 4000 --warmup 10
