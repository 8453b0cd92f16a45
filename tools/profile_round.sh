#!/bin/bash
# Profiler passes behind the numbers in profiles/: kernel summaries of the default and the
# batch-1024 bench commands, then counter passes of their own (one rocprofv3 run per counter set).
# Usage (GPU box, repository root): bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*
set -u
tag=${1:-prof}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
step() { echo "== $*"; }

step "kernel summary, default command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_default" -o p -- \
    python3 bench.py > "$out/${tag}_bench_default.json" 2> "$out/${tag}_bench_default.err" || exit 1
step "kernel summary, headline only"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_headline" -o p -- \
    python3 bench.py --no-secondary --cpu-seconds 0 > "$out/${tag}_bench_headline.json" 2> "$out/${tag}_bench_headline.err" || exit 1
step "kernel summary, batch 1024"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_b1024" -o p -- \
    python3 bench.py --nq 1024 --steps 10 --warmup 2 --cpu-seconds 0 > "$out/${tag}_bench_b1024.json" 2> "$out/${tag}_bench_b1024.err" || exit 1
step "FETCH_SIZE, scan"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_scan" -o p -- \
    python3 bench.py --steps 5 --warmup 2 --no-secondary --cpu-seconds 0 > "$out/${tag}_pmc_scan.json" 2> "$out/${tag}_pmc_scan.err" || exit 1
step "FETCH_SIZE, batch 1024"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_b1024" -o p -- \
    python3 bench.py --nq 1024 --steps 3 --warmup 1 --cpu-seconds 0 > "$out/${tag}_pmc_b1024.json" 2> "$out/${tag}_pmc_b1024.err" || exit 1
step "SQ counters, batch 1024 (a)"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$out/${tag}_sq_a_b1024" -o p -- \
    python3 bench.py --nq 1024 --steps 3 --warmup 1 --cpu-seconds 0 > "$out/${tag}_sq_a_b1024.json" 2> "$out/${tag}_sq_a_b1024.err" || exit 1
step "SQ counters, batch 1024 (b)"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU \
    --output-format csv -d "$out/${tag}_sq_b_b1024" -o p -- \
    python3 bench.py --nq 1024 --steps 3 --warmup 1 --cpu-seconds 0 > "$out/${tag}_sq_b_b1024.json" 2> "$out/${tag}_sq_b_b1024.err" || exit 1
step "in-kernel clock, batch 1024 (GRBM_GUI_ACTIVE / 8 / kernel time)"
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/${tag}_grbm_b1024" -o p -- \
    python3 bench.py --nq 1024 --steps 3 --warmup 1 --cpu-seconds 0 > "$out/${tag}_grbm_b1024.json" 2> "$out/${tag}_grbm_b1024.err" || exit 1
F16="--rows 20000000 --dtype f16 --metric l2 --nq 256 --cpu-seconds 0"
step "kernel summary, 20M x 768 fp16 L2 batch 256"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_f16b256" -o p -- \
    python3 bench.py $F16 --steps 10 --warmup 6 > "$out/${tag}_bench_f16b256.json" 2> "$out/${tag}_bench_f16b256.err" || exit 1
step "FETCH_SIZE, fp16 batch 256"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_f16b256" -o p -- \
    python3 bench.py $F16 --steps 3 --warmup 1 > "$out/${tag}_pmc_f16b256.json" 2> "$out/${tag}_pmc_f16b256.err" || exit 1
step "SQ counters, fp16 batch 256"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$out/${tag}_sq_a_f16b256" -o p -- \
    python3 bench.py $F16 --steps 3 --warmup 1 > "$out/${tag}_sq_a_f16b256.json" 2> "$out/${tag}_sq_a_f16b256.err" || exit 1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/${tag}_grbm_f16b256" -o p -- \
    python3 bench.py $F16 --steps 3 --warmup 1 > "$out/${tag}_grbm_f16b256.json" 2> "$out/${tag}_grbm_f16b256.err" || exit 1
step "kernel summary and FETCH_SIZE, scan over the bf16 mirror"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_mirror" -o p -- \
    python3 bench.py --no-secondary --cpu-seconds 0 --opt scan_mirror=1 > "$out/${tag}_bench_mirror.json" 2> "$out/${tag}_bench_mirror.err" || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_mirror" -o p -- \
    python3 bench.py --steps 5 --warmup 2 --no-secondary --cpu-seconds 0 --opt scan_mirror=1 > "$out/${tag}_pmc_mirror.json" 2> "$out/${tag}_pmc_mirror.err" || exit 1
step done
find "$out" -name '*.csv' -path "*${tag}_*" | sort
