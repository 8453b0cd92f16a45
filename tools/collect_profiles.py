"""Copy what tools/profile_round.sh left under gpurun_out/<tag>_* into profiles/<tag>_*: the bench lines,
the kernel summaries, the counter files cut down to the kernel they are about, the per-dispatch
durations of the batched kernel; then regenerate profiles/traffic.json from the FETCH_SIZE passes.

usage: python tools/collect_profiles.py <tag>
"""
import csv
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def copy(src, dst):
    shutil.copyfile(os.path.join(OUT, src), os.path.join(PROF, dst))
    print("profiles/" + dst)


def first_json_line(src, dst):
    with open(os.path.join(OUT, src)) as f:
        lines = [l for l in f if l.startswith("{")]
    with open(os.path.join(PROF, dst), "w") as f:
        f.write(lines[-1])
    print("profiles/" + dst)


def filtered(src, dst, kernel):
    with open(os.path.join(OUT, src)) as f:
        rows = list(csv.reader(f))
    k = rows[0].index("Kernel_Name")
    with open(os.path.join(PROF, dst), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(rows[0])
        for r in rows[1:]:
            if kernel in r[k]:
                w.writerow([int(x) if x.isdigit() else x for x in r])
    print("profiles/" + dst)


def dispatches(src, dst, kernel):
    with open(os.path.join(OUT, src)) as f:
        rows = list(csv.DictReader(f))
    with open(os.path.join(PROF, dst), "w") as f:
        f.write("dispatch,kernel,duration_ms\n")
        for r in rows:
            if kernel in r["Kernel_Name"]:
                f.write("%s,%s,%.4f\n" % (r["Dispatch_Id"], kernel, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    print("profiles/" + dst)


def traffic(csv_name, kernel, key, source, floor=0.0, group=1):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"),
                           os.path.join(PROF, csv_name), kernel, key, source, str(floor), str(group)])


def main():
    t = sys.argv[1]
    first_json_line(t + "_bench_default.json", t + "_bench_default.json")
    copy(t + "_default/p_kernel_stats.csv", t + "_bench_default_kernel_stats.csv")
    first_json_line(t + "_bench_headline.json", t + "_bench_headline.json")
    copy(t + "_headline/p_kernel_stats.csv", t + "_bench_headline_kernel_stats.csv")
    first_json_line(t + "_bench_b1024.json", t + "_bench_batch1024.json")
    copy(t + "_b1024/p_kernel_stats.csv", t + "_bench_batch1024_kernel_stats.csv")
    dispatches(t + "_b1024/p_kernel_trace.csv", t + "_bench_batch1024_gemm_pp_dispatches.csv", "gemm_pp")
    first_json_line(t + "_bench_f16b256.json", t + "_bench_f16_batch256.json")
    copy(t + "_f16b256/p_kernel_stats.csv", t + "_bench_f16_batch256_kernel_stats.csv")
    dispatches(t + "_f16b256/p_kernel_trace.csv", t + "_bench_f16_batch256_gemm_pp_dispatches.csv", "gemm_pp")
    filtered(t + "_pmc_f16b256/p_counter_collection.csv", t + "_gemm_pp_f16b256_pmc_fetch_size.csv", "gemm_pp")
    filtered(t + "_sq_a_f16b256/p_counter_collection.csv", t + "_gemm_pp_f16b256_pmc_sq_a.csv", "gemm_pp")
    filtered(t + "_grbm_f16b256/p_counter_collection.csv", t + "_gemm_pp_f16b256_pmc_grbm.csv", "gemm_pp")
    filtered(t + "_grbm_b1024/p_counter_collection.csv", t + "_gemm_pp_pmc_grbm.csv", "gemm_pp")
    first_json_line(t + "_bench_mirror.json", t + "_bench_mirror.json")
    copy(t + "_mirror/p_kernel_stats.csv", t + "_bench_mirror_kernel_stats.csv")
    filtered(t + "_pmc_scan/p_counter_collection.csv", t + "_scan_pmc_fetch_size.csv", "scan_kernel<")
    filtered(t + "_pmc_mirror/p_counter_collection.csv", t + "_mirror_scan_pmc_fetch_size.csv", "scan_kernel<")
    filtered(t + "_pmc_b1024/p_counter_collection.csv", t + "_gemm_pp_pmc_fetch_size.csv", "gemm_pp")
    filtered(t + "_sq_a_b1024/p_counter_collection.csv", t + "_gemm_pp_pmc_sq_a.csv", "gemm_pp")
    filtered(t + "_sq_b_b1024/p_counter_collection.csv", t + "_gemm_pp_pmc_sq_b.csv", "gemm_pp")
    cmd = "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py "
    traffic(t + "_scan_pmc_fetch_size.csv", "scan_kernel<0, 64, 3", "scan_kernel|rows_per_gpu=10000000|dim=768|dtype=f32|nq=1",
            "profiles/%s_scan_pmc_fetch_size.csv (%s--steps 5 --warmup 2 --no-secondary --cpu-seconds 0)" % (t, cmd), 1e6)
    traffic(t + "_mirror_scan_pmc_fetch_size.csv", "scan_kernel<1, 32, 3", "scan_kernel (bf16 mirror)|rows_per_gpu=10000000|dim=768|dtype=f32|nq=1",
            "profiles/%s_mirror_scan_pmc_fetch_size.csv (%s--steps 5 --warmup 2 --no-secondary --cpu-seconds 0 --opt scan_mirror=1)" % (t, cmd), 1e6)
    traffic(t + "_gemm_pp_pmc_fetch_size.csv", "gemm_pp", "gemm_pp_kernel|rows_per_gpu=10000000|dim=768|dtype=f32|nq=1024",
            "profiles/%s_gemm_pp_pmc_fetch_size.csv (%s--nq 1024 --steps 3 --warmup 1 --cpu-seconds 0; the two launches of a full pass added)" % (t, cmd),
            5e5, 2)
    traffic(t + "_gemm_pp_f16b256_pmc_fetch_size.csv", "gemm_pp", "gemm_pp384_kernel|rows_per_gpu=20000000|dim=768|dtype=f16|nq=256",
            "profiles/%s_gemm_pp_f16b256_pmc_fetch_size.csv (%s--rows 20000000 --dtype f16 --metric l2 --nq 256 --steps 3 --warmup 1 --cpu-seconds 0; the two launches of a full pass added)" % (t, cmd),
            1e6, 2)


main()
